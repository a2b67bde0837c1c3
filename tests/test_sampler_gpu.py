"""GPU: the HIP sampler against the reference's golden streams and the oracle (bit-exact)."""
import hashlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import sampler_oracle as so  # noqa: E402
from nncf_amd.sampler import MultinomialSampler  # noqa: E402


def _power(tag):
    return 1.0 if tag.endswith("_p1") else 0.75


def test_golden_streams_all_cases(sampler_golden):
    g = sampler_golden
    for tag in g["tags"].tolist():
        dist = np.ascontiguousarray(g[tag + "_dist"])
        s = MultinomialSampler(dist, dist.size, _power(tag), 1)
        assert np.array_equal(s.bounds(), g[tag + "_bounds"]), tag
        for s0 in g["states"].tolist():
            s.state = s0
            got = s.sample_batch(4096)
            assert got.dtype == np.int32 and got.shape == (4096,)
            assert np.array_equal(got, g["%s_s%d_first" % (tag, s0)]), (tag, s0)
            assert s.state == int(g["%s_s%d_state_after" % (tag, s0)][0])
            singles = np.array([s.sample() for _ in range(16)], dtype=np.int32)
            assert np.array_equal(singles, g["%s_s%d_single" % (tag, s0)]), (tag, s0)
        s.close()


def test_long_stream_checksum(sampler_golden):
    g = sampler_golden
    for tag in ("zipf17000_p1", "mod13_p1"):
        dist = np.ascontiguousarray(g[tag + "_dist"])
        s = MultinomialSampler(dist, dist.size, 1.0, 12345)
        h = hashlib.sha256()
        for _ in range(10):
            h.update(s.sample_batch(1000000).tobytes())
        assert np.array_equal(np.frombuffer(h.digest(), dtype=np.uint8), g[tag + "_s12345_sha256_1e7"])
        # one 1e7 batch == ten 1e6 batches (grid-stride path, jump-ahead)
        s.state = 12345
        h2 = hashlib.sha256(s.sample_batch(10000000).tobytes())
        assert h2.digest() == h.digest()
        s.close()


def test_ragged_batches_and_interleaving_vs_oracle():
    rs = np.random.RandomState(11)
    dist = rs.randint(0, 30, size=2500).astype(np.float64)
    dist[0] = 0.
    dist[-1] = 3.
    o = so.OracleSampler(dist, 0.75, 777)
    s = MultinomialSampler(dist, dist.size, 0.75, 777)
    for n in [0, 1, 2, 63, 64, 65, 255, 256, 257, 1000, 5120, 1, 70001, 3]:
        assert np.array_equal(s.sample_batch(n), o.sample_batch(n)), n
        if n % 2:
            assert s.sample() == o.sample()
    assert s.state == o.state
    s.close()
    o.close()


def test_slices_concatenate_to_one_stream():
    """Multi-GPU sharding contract: rank g draws slice g of the same logical batch."""
    import torch
    dist = np.arange(1, 1001, dtype=np.float64)
    s = MultinomialSampler(dist, dist.size, 1.0, 4242)
    n = 100003
    whole = s.sample_slice_device(0, n).cpu().numpy()
    parts = []
    for gk in range(8):
        lo, hi = gk * n // 8, (gk + 1) * n // 8
        parts.append(s.sample_slice_device(lo, hi - lo).cpu().numpy())
    assert np.array_equal(np.concatenate(parts), whole)
    st = s.state
    dev = s.sample_batch_device(n)
    torch.cuda.synchronize()
    assert np.array_equal(dev.cpu().numpy(), whole)
    assert s.state != st
    s.close()


def test_large_vocab_guide_path():
    """V = 1e6 (config-4 item count): boundaries via the product builder, stream vs oracle."""
    rs = np.random.RandomState(5)
    dist = np.floor(1.0 + rs.pareto(1.1, size=1000000) * 2.0)
    o = so.OracleSampler(dist, 1.0, 31337)
    s = MultinomialSampler(dist, dist.size, 1.0, 31337)
    assert np.array_equal(s.bounds(), o.bounds())
    assert np.array_equal(s.sample_batch(300000), o.sample_batch(300000))
    s.close()
    o.close()


def test_seed_zero_is_time_seeded():
    dist = np.ones(10)
    s = MultinomialSampler(dist, 10, 1.0, 0)
    assert s.state != 0  # nodesampler.cpp:59-60
    x = s.sample_batch(100)
    assert x.min() >= 0 and x.max() < 10
    s.close()
