#!/usr/bin/env python3
"""Register / scratch audit of the SHIPPED code objects.

Reads nncf_amd/libnncf_hip.so, takes every gfx950 code object out of its offload bundles, and lists
per kernel what the code-object metadata says: VGPRs, AGPRs, SGPRs, spilled VGPRs / SGPRs, private
(scratch) segment, static LDS, and -- from the disassembly -- the number of scratch_load / scratch_store
instructions.  Nothing is recompiled: the numbers are those of the binary the GPU box runs.

  python tools/isa_resources.py                      table on stdout
  python tools/isa_resources.py --out profiles/r05_isa_resources.txt
  python tools/isa_resources.py --spills             only the kernels that spill

`audit()` is what tests/test_isa_resources.py calls (a default-path kernel that spills fails the CPU suite).
"""
import argparse
import os
import re
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"

FIELDS = (".vgpr_count", ".agpr_count", ".sgpr_count", ".vgpr_spill_count", ".sgpr_spill_count",
          ".private_segment_fixed_size", ".group_segment_fixed_size", ".max_flat_workgroup_size")


def code_objects(path):
    """Yield the gfx950 ELF images held by the offload bundles of a host object / shared library."""
    d = open(path, "rb").read()
    pos = 0
    while True:
        i = d.find(MAGIC, pos)
        if i < 0:
            return
        n, = struct.unpack_from("<Q", d, i + len(MAGIC))
        p = i + len(MAGIC) + 8
        end = i + len(MAGIC)
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", d, p)
            triple = d[p + 24:p + 24 + tl].decode()
            p += 24 + tl
            if "gfx950" in triple and size:
                yield d[i + off:i + off + size]
            end = max(end, i + off + size)
        pos = end


def _demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True)
    return out.stdout.split("\n")[:len(names)]


def _scratch_counts(co_path):
    """symbol -> (scratch loads, scratch stores) from the disassembly."""
    dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", co_path], capture_output=True, text=True, check=True).stdout
    counts, cur = {}, None
    for line in dis.split("\n"):
        m = re.match(r"^[0-9a-f]+ <([^>]+)>:", line)
        if m:
            cur = m.group(1)
            counts[cur] = [0, 0]
        elif cur is not None:
            s = line.strip()
            if s.startswith("scratch_load"):
                counts[cur][0] += 1
            elif s.startswith("scratch_store"):
                counts[cur][1] += 1
    return counts


def audit(lib=None, with_scratch=True):
    """List of dicts, one per kernel of the library: name (demangled), symbol, the metadata fields
    (without the leading dot), scratch_loads, scratch_stores."""
    lib = lib or os.path.join(ROOT, "nncf_amd", "libnncf_hip.so")
    rows = []
    with tempfile.TemporaryDirectory() as tmp:
        for k, image in enumerate(code_objects(lib)):
            co = os.path.join(tmp, "co%d.elf" % k)
            with open(co, "wb") as fh:
                fh.write(image)
            notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], capture_output=True, text=True, check=True).stdout
            scratch = _scratch_counts(co) if with_scratch else {}
            cur = None
            for line in notes.split("\n"):
                # a kernel's record starts at "  - .key:" and holds its own fields at a four-space indent
                if line.startswith("  - ."):
                    cur = {}
                    rows.append(cur)
                    line = "    " + line[4:]
                if cur is None or not line.startswith("    .") or line.startswith("     "):
                    continue
                m = re.match(r"^    (\.[a-z_]+):\s*(.*)$", line)
                if not m:
                    continue
                key, val = m.group(1), m.group(2).strip()
                if key in FIELDS:
                    cur[key[1:]] = int(val)
                elif key == ".symbol":
                    cur["symbol"] = val[:-3] if val.endswith(".kd") else val
            for r in rows:
                if "symbol" in r and "scratch_loads" not in r:
                    ld, st = scratch.get(r["symbol"], (0, 0))
                    r["scratch_loads"], r["scratch_stores"] = ld, st
    rows = [r for r in rows if "symbol" in r]
    for r, nm in zip(rows, _demangle([r["symbol"] for r in rows])):
        r["name"] = re.sub(r"^void ", "", nm)
        r["name"] = re.sub(r"\(.*\)$", "", r["name"]) if r["name"].endswith(")") else r["name"]
    rows.sort(key=lambda r: r["name"])
    return rows


def fmt(rows):
    head = "%6s %5s %5s %6s %6s %8s %7s %4s %4s  %s" % ("vgpr", "agpr", "sgpr", "vspill", "sspill", "scratchB", "ldsB", "s_ld", "s_st", "kernel")
    out = [head]
    for r in rows:
        out.append("%6d %5d %5d %6d %6d %8d %7d %4d %4d  %s" % (
            r.get("vgpr_count", -1), r.get("agpr_count", 0), r.get("sgpr_count", -1), r.get("vgpr_spill_count", 0),
            r.get("sgpr_spill_count", 0), r.get("private_segment_fixed_size", 0), r.get("group_segment_fixed_size", 0),
            r.get("scratch_loads", 0), r.get("scratch_stores", 0), r["name"]))
    return "\n".join(out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default=None)
    ap.add_argument("--out", default=None)
    ap.add_argument("--spills", action="store_true")
    a = ap.parse_args()
    rows = audit(a.lib)
    if a.spills:
        rows = [r for r in rows if r.get("vgpr_spill_count", 0) or r.get("private_segment_fixed_size", 0)]
    text = fmt(rows)
    nsp = sum(1 for r in rows if r.get("vgpr_spill_count", 0))
    text += "\n# %d kernels, %d with spilled VGPRs\n" % (len(rows), nsp)
    if a.out:
        with open(a.out, "w") as fh:
            fh.write(text)
    print(text)


if __name__ == "__main__":
    sys.exit(main())
