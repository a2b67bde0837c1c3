"""Drop-in for the reference's `sampler` extension module (sampler/sampler.pyx).

    from nncf_amd.sampler import MultinomialSampler

mirrors `MultinomialSampler(input, dist_size, neg_sampling_power=0.75, rand_seed=0)` with
`.sample()` / `.sample_batch(n)` (sampler.pyx:19-38), drawing on the MI355X through
libnncf_hip.so.  Index streams are bit-identical to the reference C++ for equal LCG state.
"""
import ctypes

import numpy as np

from . import _lib


def build_bounds(dist, power=0.75):
    """Host half of the constructor: int32 boundary array of the reference's 1e8-entry table
    (set_table, nodesampler.cpp:29-49).  Needs no GPU."""
    dist = _check_dist(dist)
    out = np.zeros(dist.size, dtype=np.int32)
    _lib.check(_lib.lib().nncf_sampler_build_bounds(
        dist.ctypes.data_as(_lib.c_f64p), dist.size, float(power), out.ctypes.data_as(_lib.c_i32p)))
    return out


def lcg_jump(state, n):
    """LCG state after n draws (nodesampler.cpp:23-26 applied n times), O(log n)."""
    return int(_lib.lib().nncf_lcg_jump(ctypes.c_uint64(state & (2**64 - 1)), ctypes.c_uint64(n)))


def _check_dist(dist):
    # sampler.pyx:25 `np.ndarray[double, ndim=1, mode="c"] input not None`
    if dist is None:
        raise TypeError("Argument 'input' must not be None")
    if not isinstance(dist, np.ndarray):
        raise TypeError("Argument 'input' has incorrect type (expected numpy.ndarray, got %s)" % type(dist).__name__)
    if dist.dtype != np.float64:
        raise ValueError("Buffer dtype mismatch, expected 'double' but got '%s'" % dist.dtype)
    if dist.ndim != 1:
        raise ValueError("Buffer has wrong number of dimensions (expected 1, got %d)" % dist.ndim)
    if not dist.flags["C_CONTIGUOUS"]:
        raise ValueError("ndarray is not C-contiguous")
    return dist


class MultinomialSampler(object):
    """sample from given categorical distribution (sampler.pyx:19-38).

    input: float64 C-contiguous 1-D weights (degree histogram); dist_size: len(input);
    neg_sampling_power: weights are raised to this power; rand_seed: LCG state before the
    first draw (0 => time-seeded, nodesampler.cpp:59-60).

    Extra (not in the reference): `device` ordinal, `.state` property, and
    `.sample_batch_device(n)` which returns a torch int32 CUDA tensor without the host copy.
    """

    _SINGLE_CACHE = 1024

    def __init__(self, input, dist_size, neg_sampling_power=0.75, rand_seed=0, device=0):
        dist = _check_dist(input)
        dist_size = int(dist_size)
        if dist_size > dist.size:
            raise ValueError("dist_size %d exceeds len(input) %d" % (dist_size, dist.size))
        self._L = _lib.lib()
        self._h = ctypes.c_void_p()
        self.device = int(device)
        _lib.check(self._L.nncf_sampler_create(
            dist.ctypes.data_as(_lib.c_f64p), dist_size, float(neg_sampling_power),
            ctypes.c_uint64(int(rand_seed) & (2**64 - 1)), self.device, ctypes.byref(self._h)))
        self.dist_size = dist_size
        self._cache = None
        self._cache_pos = 0

    # -- reference API ---------------------------------------------------------------
    def sample(self):
        """return a node index (int) -- sampler.pyx:30-32.  Served from a block of
        device-generated draws so that interleaving with sample_batch keeps one stream."""
        if self._cache is None or self._cache_pos >= self._cache.size:
            import torch
            buf = torch.empty(self._SINGLE_CACHE, dtype=torch.int32, device="cuda:%d" % self.device)
            with torch.cuda.device(self.device):
                _lib.check(self._L.nncf_sampler_sample_slice_dev(
                    self._h, 0, self._SINGLE_CACHE, ctypes.c_void_p(buf.data_ptr()), _lib.stream_ptr()))
            self._cache = buf.cpu().numpy()
            self._cache_pos = 0
        v = int(self._cache[self._cache_pos])
        self._cache_pos += 1
        _lib.check(self._L.nncf_sampler_skip(self._h, 1))
        return v

    def sample_batch(self, n):
        """return an int32 array of n node indices -- sampler.pyx:33-38."""
        n = int(n)
        if n < 0:
            raise ValueError("negative dimensions are not allowed")
        out = np.zeros(n, dtype=np.int32)
        if n:
            import torch
            self._drop_cache()
            with torch.cuda.device(self.device):
                _lib.check(self._L.nncf_sampler_sample_batch(
                    self._h, n, out.ctypes.data_as(_lib.c_i32p), _lib.stream_ptr()))
        return out

    # -- device-resident extensions --------------------------------------------------
    def sample_batch_device(self, n, out=None):
        """n draws into a torch int32 CUDA tensor (no host round trip, asynchronous)."""
        import torch
        n = int(n)
        if out is None:
            out = torch.empty(n, dtype=torch.int32, device="cuda:%d" % self.device)
        assert out.dtype == torch.int32 and out.is_contiguous() and out.numel() >= n
        self._drop_cache()
        with torch.cuda.device(self.device):
            _lib.check(self._L.nncf_sampler_sample_dev(self._h, n, ctypes.c_void_p(out.data_ptr()), _lib.stream_ptr()))
        return out[:n]

    def sample_slice_device(self, offset, count, out=None):
        """Draws [offset, offset+count) of the next batch, without advancing the state."""
        import torch
        if out is None:
            out = torch.empty(int(count), dtype=torch.int32, device="cuda:%d" % self.device)
        self._drop_cache()
        with torch.cuda.device(self.device):
            _lib.check(self._L.nncf_sampler_sample_slice_dev(
                self._h, int(offset), int(count), ctypes.c_void_p(out.data_ptr()), _lib.stream_ptr()))
        return out

    def skip(self, n):
        self._drop_cache()
        _lib.check(self._L.nncf_sampler_skip(self._h, int(n)))

    def bounds(self):
        out = np.zeros(self.dist_size, dtype=np.int32)
        _lib.check(self._L.nncf_sampler_bounds(self._h, out.ctypes.data_as(_lib.c_i32p)))
        return out

    @property
    def state(self):
        s = ctypes.c_uint64()
        _lib.check(self._L.nncf_sampler_get_state(self._h, ctypes.byref(s)))
        return int(s.value)

    @state.setter
    def state(self, value):
        self._drop_cache()
        _lib.check(self._L.nncf_sampler_set_state(self._h, ctypes.c_uint64(int(value) & (2**64 - 1))))

    def _drop_cache(self):
        self._cache = None
        self._cache_pos = 0

    def close(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._L.nncf_sampler_destroy(h)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
