"""CPU: libnncf_hip.so loads and exports every symbol include/nncf_hip.h declares."""
import ctypes
import os
import re

from nncf_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    names = set()
    for fn in os.listdir(os.path.join(ROOT, "include")):
        if not fn.endswith(".h"):
            continue
        text = open(os.path.join(ROOT, "include", fn)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names |= set(re.findall(r"\b(nncf_[a-z0-9_]+)\s*\(", text))
    return sorted(names)


def test_every_declared_symbol_is_exported():
    lib = ctypes.CDLL(_lib.LIB_PATH)
    names = _declared()
    assert len(names) >= 10
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing


def test_version_and_device_probe():
    L = _lib.lib()
    assert b"gfx950" in L.nncf_version()
    assert L.nncf_device_count() >= 0


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: nothing under nncf_amd/ may reference it."""
    bad = []
    for dp, _, fns in os.walk(os.path.join(ROOT, "nncf_amd")):
        for fn in fns:
            if fn.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dp, fn), errors="replace").read()
                if re.search(r"\bfrom\s+oracle\b|\bimport\s+oracle\b|oracle/|liboracle", txt):
                    bad.append(os.path.join(dp, fn))
    assert not bad, bad
