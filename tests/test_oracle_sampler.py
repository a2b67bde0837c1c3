"""CPU: pins the oracle (oracle/nodesampler_oracle.c) to golden vectors produced by the
compiled, unmodified reference (tools/gen_golden_sampler.py), and checks the product's host
logic (boundary builder, LCG jump-ahead) against both.  SURVEY.md section 8(c)."""
import hashlib

import numpy as np
import pytest

from oracle import sampler_oracle as so
from nncf_amd import sampler as ns


def _power(tag):
    return 1.0 if tag.endswith("_p1") else 0.75


def test_lcg_known_answers():
    # state0 = 42 (compiled reference + independent big-int arithmetic)
    kat = [16159453, 28765935, 26410514, 53133836, 51638694, 96321661]
    assert so.lcg_draws_py(42, 6) == kat
    assert so.lcg_draws(42, 6).tolist() == kat
    # high-bit state: 64-bit wrap-around, not a 48-bit java.util.Random
    s0 = 2**63 + 5
    assert so.lcg_draws(s0, 64).tolist() == so.lcg_draws_py(s0, 64)


def test_table_known_answers(sampler_golden):
    g = sampler_golden
    # dist = [5,3,0,2]: per-id table counts
    assert np.diff(np.concatenate([[0], g["toy_p1_bounds"]])).tolist() == [50000000, 30000000, 0, 20000000]
    assert np.diff(np.concatenate([[0], g["toy_p0_75_bounds"]])).tolist() == [45772771, 31204745, 0, 23022484]
    assert g["toy_p1_s42_first"][:10].tolist() == [0, 0, 0, 1, 1, 3, 3, 1, 0, 3]
    assert g["mod13_p1_s12345_first"][:8].tolist() == [8455, 9552, 8549, 15325, 1973, 14580, 9007, 2885]


@pytest.mark.parametrize("tag", ["toy_p1", "toy_p0_75", "zipf17000_p0_75", "zeros_edges5000_p1", "single_p1"])
def test_oracle_matches_reference_golden(sampler_golden, tag):
    g = sampler_golden
    o = so.OracleSampler(g[tag + "_dist"], _power(tag), 1)
    assert o.filled == 100000000
    assert np.array_equal(o.bounds(), g[tag + "_bounds"])
    for s0 in g["states"].tolist():
        o.state = s0
        assert np.array_equal(o.sample_batch(4096), g["%s_s%d_first" % (tag, s0)])
        assert o.state == int(g["%s_s%d_state_after" % (tag, s0)][0])
        singles = np.array([o.sample() for _ in range(16)], dtype=np.int32)
        assert np.array_equal(singles, g["%s_s%d_single" % (tag, s0)])
    o.close()


def test_oracle_long_stream_checksum(sampler_golden):
    g = sampler_golden
    o = so.OracleSampler(g["zipf17000_p1_dist"], 1.0, 12345)
    h = hashlib.sha256()
    for _ in range(10):
        h.update(o.sample_batch(1000000).tobytes())
    assert np.array_equal(np.frombuffer(h.digest(), dtype=np.uint8), g["zipf17000_p1_s12345_sha256_1e7"])
    o.close()


@pytest.mark.skipif(not so.ref_available(), reason="oracle/_ref not built here")
def test_oracle_matches_compiled_reference_live():
    rs = np.random.RandomState(7)
    dist = np.floor(rs.pareto(1.2, size=3000) * 3).astype(np.float64)  # many zeros, heavy tail
    dist[17] = 0.
    for power in (1.0, 0.75, 0.5):
        r = so.RefSampler(dist, power, 99)
        o = so.OracleSampler(dist, power, 99)
        assert np.array_equal(r.bounds(), o.bounds())
        assert np.array_equal(r.sample_batch(20000), o.sample_batch(20000))
        assert r.state == o.state
        assert np.array_equal(ns.build_bounds(dist, power), r.bounds())
        r.close()
        o.close()


def test_product_bounds_bit_exact(sampler_golden):
    g = sampler_golden
    for tag in g["tags"].tolist():
        b = ns.build_bounds(np.ascontiguousarray(g[tag + "_dist"]), _power(tag))
        assert np.array_equal(b, g[tag + "_bounds"]), tag


def test_product_bounds_random_vs_oracle():
    rs = np.random.RandomState(3)
    for trial in range(3):
        v = int(rs.randint(2, 4000))
        dist = rs.randint(0, 50, size=v).astype(np.float64)
        dist[rs.randint(v)] = 1.0
        power = [1.0, 0.75, 0.3][trial]
        o = so.OracleSampler(dist, power, 1)
        assert np.array_equal(ns.build_bounds(dist, power), o.bounds())
        o.close()


def test_lcg_jump_matches_sequential():
    for s0 in (1, 42, 12345, 2**63 + 5, 2**64 - 1):
        s = s0
        for n in range(0, 300):
            assert ns.lcg_jump(s0, n) == s
            s = (s * 25214903917 + 11) & (2**64 - 1)
    assert ns.lcg_jump(ns.lcg_jump(7, 1000003), 999) == ns.lcg_jump(7, 1000003 + 999)


def test_sampler_argument_errors():
    # mirrors the Cython buffer validation of sampler.pyx:25
    with pytest.raises(TypeError):
        ns.build_bounds(None)
    with pytest.raises(ValueError):
        ns.build_bounds(np.ones(4, dtype=np.float32))
    with pytest.raises(ValueError):
        ns.build_bounds(np.ones((2, 2)))
    with pytest.raises(ValueError):
        ns.build_bounds(np.ones(8)[::2])
    with pytest.raises(ValueError):
        ns.build_bounds(np.zeros(5))  # nothing to sample (reference: undefined behaviour)
    with pytest.raises(ValueError):
        ns.build_bounds(np.array([1., -2.]))


def test_multiply_high_modulo_is_the_percent_operator():
    """csrc/sampler.hip mod_table: r = x mod 1e8 for x < 2^48 as x - ((x * ceil(2^75 / 1e8)) >> 75) * 1e8 in the low 32
    bits -- what replaces the software 64-bit `%` of nodesampler.cpp:25.  Exactness on the corner cases (every multiple of
    1e8 and its neighbours at both ends of the 48-bit range) and on random states; the kernel itself is pinned by the golden
    streams and the SHA-256 of 10^7 draws (tests/test_sampler_gpu.py)."""
    D, M = 10 ** 8, -(-2 ** 75 // 10 ** 8)
    assert M == 377789318629572 and 2 ** 75 <= M * D < 2 ** 75 + 2 ** 27

    def mod_table(x):
        q = (x * M) >> 75
        return ((x & 0xFFFFFFFF) - ((q & 0xFFFFFFFF) * D & 0xFFFFFFFF)) & 0xFFFFFFFF

    top = 2 ** 48 - 1
    corners = [0, 1, D - 1, D, D + 1, top, top - 1]
    for k in list(range(1, 2000)) + list(range(top // D - 2000, top // D + 1)):
        corners += [k * D - 1, k * D, k * D + 1]
    for x in corners:
        if 0 <= x <= top:
            assert mod_table(x) == x % D, x
    rs = np.random.RandomState(0)
    state = 12345
    for _ in range(200000):   # states of the reference's own LCG (nodesampler.cpp:24)
        state = (state * 25214903917 + 11) & (2 ** 64 - 1)
        x = state >> 16
        assert mod_table(x) == x % D
    xs = (rs.randint(0, 2 ** 24, size=300000).astype(object) << 24) | rs.randint(0, 2 ** 24, size=300000).astype(object)
    for x in xs:
        assert mod_table(int(x)) == int(x) % D


def test_two_multiply_modulo_is_the_percent_operator():
    """csrc/sampler.hip mod_table, DEVICE form (round 5: the kernel is bound by quarter-rate integer multiplies): with y = x >> 16
    and C = floor(2^58 / 1e8) < 2^32, q = mulhi32(y, C) >> 10 is floor(x / 1e8) or one less, so r = (x - q 1e8) mod 2^32 lies
    in [0, 2e8) and one conditional subtraction finishes it.  Corner cases (every multiple of 1e8 +- 1 in a stride through the
    48-bit range, the values around every 2^16 boundary next to a multiple), the reference LCG's own states, 10^7 random
    states (vectorised in uint64: y C < 2^64); the kernel itself is pinned by the golden streams and the SHA-256 of 10^7 draws."""
    D = 10 ** 8
    C = 2 ** 58 // D
    assert C == 2882303761 and C < 2 ** 32

    def mod_dev(x):
        q = (((x >> 16) * C) >> 32) >> 10
        assert q in (x // D, x // D - 1), x
        r = ((x & 0xFFFFFFFF) - (q * D & 0xFFFFFFFF)) & 0xFFFFFFFF
        return r - D if r >= D else r

    top = 2 ** 48 - 1
    corners = [0, 1, D - 1, D, D + 1, top, top - 1, 2 ** 32 - 1, 2 ** 32, 2 ** 32 + 1]
    for k in list(range(1, 3000)) + list(range(1, top // D, 997)) + list(range(top // D - 3000, top // D + 1)):
        v = k * D
        corners += [v - 1, v, v + 1, v | 0xFFFF, v & ~0xFFFF, (v & ~0xFFFF) - 1]
    for x in corners:
        if 0 <= x <= top:
            assert mod_dev(x) == x % D, x
    state = 12345
    for _ in range(200000):   # states of the reference's own LCG (nodesampler.cpp:24)
        state = (state * 25214903917 + 11) & (2 ** 64 - 1)
        assert mod_dev(state >> 16) == (state >> 16) % D
    rs = np.random.RandomState(1)
    x = (rs.randint(0, 2 ** 24, size=10 ** 7).astype(np.uint64) << np.uint64(24)) | rs.randint(0, 2 ** 24, size=10 ** 7).astype(np.uint64)
    q = (((x >> np.uint64(16)) * np.uint64(C)) >> np.uint64(42))
    r = ((x & np.uint64(0xFFFFFFFF)) - ((q * np.uint64(D)) & np.uint64(0xFFFFFFFF))) & np.uint64(0xFFFFFFFF)
    r = np.where(r >= np.uint64(D), r - np.uint64(D), r)
    assert np.array_equal(r, x % np.uint64(D))
