"""TEST INFRASTRUCTURE ONLY -- ctypes front-ends for the sampler checkers.

* ``OracleSampler``   : plain-C restatement (oracle/nodesampler_oracle.c) of
  /root/reference/sampler/nodesampler.cpp:14-76.
* ``RefSampler``      : the unmodified reference C++ compiled by oracle/Makefile into
  oracle/_ref/ (present only where that build ran; ``ref_available()`` tells).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module.  The product package nncf_amd never does.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ORC_SO = os.path.join(_HERE, "liboracle_sampler.so")
_REF_SO = os.path.join(_HERE, "_ref", "libnodesampler_ref.so")

_c_double_p = ctypes.POINTER(ctypes.c_double)
_c_int_p = ctypes.POINTER(ctypes.c_int32)


def build(force=False):
    """Compile the C restatement (and the reference shim when /root/reference exists)."""
    if force or not os.path.exists(_ORC_SO) or (
            os.path.getmtime(_ORC_SO) < os.path.getmtime(os.path.join(_HERE, "nodesampler_oracle.c"))):
        subprocess.check_call(["make", "-C", _HERE, "liboracle_sampler.so"], stdout=subprocess.DEVNULL)
    if os.path.isdir("/root/reference/sampler") and (force or not os.path.exists(_REF_SO)):
        subprocess.check_call(["make", "-C", _HERE, "ref"], stdout=subprocess.DEVNULL)


def ref_available():
    return os.path.exists(_REF_SO)


_orc = None
_ref = None


def _load_orc():
    global _orc
    if _orc is None:
        build()
        lib = ctypes.CDLL(_ORC_SO)
        lib.orc_sampler_create.restype = ctypes.c_void_p
        lib.orc_sampler_create.argtypes = [_c_double_p, ctypes.c_int, ctypes.c_double, ctypes.c_uint64]
        lib.orc_sampler_sample.restype = ctypes.c_int
        lib.orc_sampler_sample.argtypes = [ctypes.c_void_p]
        lib.orc_sampler_sample_batch.restype = None
        lib.orc_sampler_sample_batch.argtypes = [ctypes.c_void_p, ctypes.c_longlong, _c_int_p]
        lib.orc_sampler_get_state.restype = ctypes.c_uint64
        lib.orc_sampler_get_state.argtypes = [ctypes.c_void_p]
        lib.orc_sampler_set_state.restype = None
        lib.orc_sampler_set_state.argtypes = [ctypes.c_void_p, ctypes.c_uint64]
        lib.orc_sampler_filled.restype = ctypes.c_longlong
        lib.orc_sampler_filled.argtypes = [ctypes.c_void_p]
        lib.orc_sampler_bounds.restype = None
        lib.orc_sampler_bounds.argtypes = [ctypes.c_void_p, _c_int_p]
        lib.orc_sampler_destroy.restype = None
        lib.orc_sampler_destroy.argtypes = [ctypes.c_void_p]
        lib.orc_lcg_draws.restype = None
        lib.orc_lcg_draws.argtypes = [ctypes.c_uint64, ctypes.c_int, ctypes.POINTER(ctypes.c_int64)]
        _orc = lib
    return _orc


def _load_ref():
    global _ref
    if _ref is None:
        if not ref_available():
            raise RuntimeError("oracle/_ref/libnodesampler_ref.so not built (run `make -C oracle ref` "
                               "in a container that has /root/reference)")
        lib = ctypes.CDLL(_REF_SO)
        lib.ref_sampler_create.restype = ctypes.c_void_p
        lib.ref_sampler_create.argtypes = [_c_double_p, ctypes.c_int, ctypes.c_double, ctypes.c_ulonglong]
        lib.ref_sampler_sample.restype = ctypes.c_int
        lib.ref_sampler_sample.argtypes = [ctypes.c_void_p]
        lib.ref_sampler_sample_batch.restype = None
        lib.ref_sampler_sample_batch.argtypes = [ctypes.c_void_p, ctypes.c_int, _c_int_p]
        lib.ref_sampler_get_state.restype = ctypes.c_ulonglong
        lib.ref_sampler_get_state.argtypes = [ctypes.c_void_p]
        lib.ref_sampler_set_state.restype = None
        lib.ref_sampler_set_state.argtypes = [ctypes.c_void_p, ctypes.c_ulonglong]
        lib.ref_sampler_table.restype = _c_int_p
        lib.ref_sampler_table.argtypes = [ctypes.c_void_p]
        lib.ref_sampler_table_size.restype = ctypes.c_longlong
        lib.ref_sampler_destroy.restype = None
        lib.ref_sampler_destroy.argtypes = [ctypes.c_void_p]
        _ref = lib
    return _ref


def lcg_draws(state0, n):
    """First n values of Rand() (nodesampler.cpp:23-26) from LCG state `state0`."""
    out = np.zeros(n, dtype=np.int64)
    _load_orc().orc_lcg_draws(ctypes.c_uint64(state0 & (2**64 - 1)), n,
                              out.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)))
    return out


def lcg_draws_py(state0, n):
    """Same, in pure Python big-int arithmetic (independent of any C compiler)."""
    s = state0 & (2**64 - 1)
    out = []
    for _ in range(n):
        s = (s * 25214903917 + 11) & (2**64 - 1)
        out.append((s >> 16) % 100000000)
    return out


class _Base(object):
    def sample(self):
        raise NotImplementedError

    def bounds(self):
        """bound[i] = #table entries with id <= i  (int32[n])."""
        raise NotImplementedError


class OracleSampler(_Base):
    """Plain-C restatement; `state0` is the LCG state before the first draw."""

    def __init__(self, dist, power=0.75, state0=1):
        dist = np.ascontiguousarray(dist, dtype=np.float64)
        self._lib = _load_orc()
        self.n = int(dist.size)
        self._h = self._lib.orc_sampler_create(dist.ctypes.data_as(_c_double_p), self.n,
                                               float(power), ctypes.c_uint64(state0 & (2**64 - 1)))
        if not self._h:
            raise MemoryError("oracle sampler table (400 MB) allocation failed")

    def sample(self):
        return int(self._lib.orc_sampler_sample(self._h))

    def sample_batch(self, n):
        out = np.zeros(int(n), dtype=np.int32)
        self._lib.orc_sampler_sample_batch(self._h, int(n), out.ctypes.data_as(_c_int_p))
        return out

    @property
    def state(self):
        return int(self._lib.orc_sampler_get_state(self._h))

    @state.setter
    def state(self, s):
        self._lib.orc_sampler_set_state(self._h, ctypes.c_uint64(s & (2**64 - 1)))

    @property
    def filled(self):
        return int(self._lib.orc_sampler_filled(self._h))

    def bounds(self):
        out = np.zeros(self.n, dtype=np.int32)
        self._lib.orc_sampler_bounds(self._h, out.ctypes.data_as(_c_int_p))
        return out

    def close(self):
        if getattr(self, "_h", None):
            self._lib.orc_sampler_destroy(self._h)
            self._h = None

    __del__ = close


class RefSampler(_Base):
    """The unmodified reference NodeSampler with its LCG state pinned to `state0`."""

    def __init__(self, dist, power=0.75, state0=1):
        dist = np.ascontiguousarray(dist, dtype=np.float64)
        self._lib = _load_ref()
        self.n = int(dist.size)
        self._h = self._lib.ref_sampler_create(dist.ctypes.data_as(_c_double_p), self.n,
                                               float(power), ctypes.c_ulonglong(state0 & (2**64 - 1)))
        if not self._h:
            raise MemoryError("reference sampler allocation failed")

    def sample(self):
        return int(self._lib.ref_sampler_sample(self._h))

    def sample_batch(self, n):
        out = np.zeros(int(n), dtype=np.int32)
        self._lib.ref_sampler_sample_batch(self._h, int(n), out.ctypes.data_as(_c_int_p))
        return out

    @property
    def state(self):
        return int(self._lib.ref_sampler_get_state(self._h))

    @state.setter
    def state(self, s):
        self._lib.ref_sampler_set_state(self._h, ctypes.c_ulonglong(s & (2**64 - 1)))

    def table(self):
        size = int(self._lib.ref_sampler_table_size())
        ptr = self._lib.ref_sampler_table(self._h)
        return np.ctypeslib.as_array(ptr, shape=(size,))

    def bounds(self):
        t = self.table()
        # table is monotone: bound[i] = index of first entry > i
        return np.searchsorted(t, np.arange(self.n), side="right").astype(np.int32)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.ref_sampler_destroy(self._h)
            self._h = None

    __del__ = close
