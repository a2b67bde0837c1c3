"""GPU: the HIP sampler against the reference's golden streams and the oracle (bit-exact)."""
import hashlib
import os

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


def test_lds_resident_tables_and_misaligned_outputs_give_the_same_stream():
    """Round 5: long streams run with bound[] + a coarser guide in LDS (sample_kernel<true>), four draws per thread and
    16-byte stores; an output that does not start on 16 bytes shifts the quads.  Every variant is the oracle's stream."""
    import torch
    rs = np.random.RandomState(3)
    dist = (rs.zipf(1.3, 17000) % 5000 + 1).astype(np.float64)
    n = (1 << 20) + 37
    o = so.OracleSampler(dist, 1.0, 999)
    want = o.sample_batch(n)
    o.close()
    s = MultinomialSampler(dist, dist.size, 1.0, 999)
    buf = torch.full((n + 8,), -7, dtype=torch.int32, device="cuda")
    for off in (0, 1, 2, 3):
        buf.fill_(-7)
        s.sample_slice_device(0, n, out=buf[off:off + n])
        got = buf.cpu().numpy()
        assert np.array_equal(got[off:off + n], want), off
        assert (got[:off] == -7).all() and (got[off + n:] == -7).all(), off   # nothing written outside the slice
    # a slice in the middle of the stream, LDS variant (>= 2^19 draws) and global-table variant (short)
    for lo, cnt in ((12345, 1 << 19), (5, (1 << 17) + 1), (777, 5000), (3, 1), (1 << 19, 3)):
        for off in (0, 3):
            got = s.sample_slice_device(lo, cnt, out=buf[off:off + cnt]).cpu().numpy()
            assert np.array_equal(got, want[lo:lo + cnt]), (lo, cnt, off)
    s.close()


def test_seed_zero_is_time_seeded():
    dist = np.ones(10)
    s = MultinomialSampler(dist, 10, 1.0, 0)
    assert s.state != 0  # nodesampler.cpp:59-60
    x = s.sample_batch(100)
    assert x.min() >= 0 and x.max() < 10
    s.close()


def test_device_group_shuffle_properties():
    """nncf_group_shuffle: a permutation of the links; chop = 0: every group's rows contiguous, group
    order and within-group order change with the seed; chop = c: the aligned runs of the chop = 0
    result (same seed) shuffled as units, ragged tail kept last (data_utils.py:218-241)."""
    import torch
    from nncf_amd.data_utils import group_shuffle_train_device
    rs = np.random.RandomState(0)
    for n, nu, ni, by in [(20003, 500, 300, "item"), (777, 50, 40, "user"), (5, 3, 2, "item"), (1, 1, 1, "item")]:
        u = rs.randint(0, nu, n).astype(np.int32)
        c = (rs.zipf(1.4, n) % ni).astype(np.int32)
        du, dc = torch.as_tensor(u).cuda(), torch.as_tensor(c).cuda()
        uo, co, perm = [t.cpu().numpy() for t in group_shuffle_train_device(du, dc, by=by, chop=0, seed=7, want_perm=True)]
        assert np.array_equal(np.sort(perm), np.arange(n))                       # a permutation ...
        assert np.array_equal(uo, u[perm]) and np.array_equal(co, c[perm])       # ... applied to both columns
        g = uo if by == "user" else co
        starts = np.flatnonzero(np.r_[True, g[1:] != g[:-1]])
        assert len(starts) == len(np.unique(g))                                   # groups contiguous
        u2, c2, perm2 = [t.cpu().numpy() for t in group_shuffle_train_device(du, dc, by=by, chop=0, seed=8, want_perm=True)]
        if n > 100:
            assert not np.array_equal(perm, perm2)
            g2 = u2 if by == "user" else c2
            first = lambda x: x[np.flatnonzero(np.r_[True, x[1:] != x[:-1]])]  # noqa: E731
            assert not np.array_equal(first(g), first(g2))                        # group order differs
        for chop in (1, 4, 5):
            uc, cc, pc = [t.cpu().numpy() for t in group_shuffle_train_device(du, dc, by=by, chop=chop, seed=7, want_perm=True)]
            bulk = (n // chop) * chop
            assert np.array_equal(pc[bulk:], perm[bulk:])                         # tail untouched
            runs0 = perm[:bulk].reshape(-1, chop)
            runs1 = pc[:bulk].reshape(-1, chop)
            key = lambda r: tuple(r.tolist())  # noqa: E731
            assert sorted(map(key, runs0)) == sorted(map(key, runs1))             # same runs, other order
            if bulk // chop > 20:
                assert not np.array_equal(runs0, runs1)
    # uniformity smoke: position of a fixed row inside its group over seeds is spread out
    u = np.zeros(64, dtype=np.int32)
    c = np.zeros(64, dtype=np.int32)
    du, dc = torch.as_tensor(u).cuda(), torch.as_tensor(c).cuda()
    pos = [int(np.flatnonzero(group_shuffle_train_device(du, dc, seed=s, want_perm=True)[2].cpu().numpy() == 0)[0])
           for s in range(40)]
    assert len(set(pos)) > 15


def test_device_group_shuffle_equals_its_oracle_exactly():
    """nncf_group_shuffle's permutation == the numpy restatement of the same counter-based keys and stable sorts
    (oracle/shuffle_oracle.py), bit for bit: items and users, chop 0 / 2 / 5, ragged tails, tiny inputs."""
    import torch
    from nncf_amd.data_utils import group_shuffle_train_device
    from oracle.shuffle_oracle import group_shuffle_perm
    rs = np.random.RandomState(1)
    for n, nu, ni in [(20003, 500, 300), (4096, 60, 4000), (7, 3, 2), (1, 1, 1)]:
        u = rs.randint(0, nu, n).astype(np.int32)
        c = (rs.zipf(1.3, n) % ni).astype(np.int32)
        du, dc = torch.as_tensor(u).cuda(), torch.as_tensor(c).cuda()
        for by in ("item", "user"):
            for chop in (0, 2, 5):
                for seed in (0, 7, 2 ** 63 + 11):
                    uo, co, perm = [t.cpu().numpy() for t in
                                    group_shuffle_train_device(du, dc, by=by, chop=chop, seed=seed, want_perm=True)]
                    want = group_shuffle_perm(u, c, by == "user", chop, seed)
                    assert np.array_equal(perm, want), (n, by, chop, seed)
                    assert np.array_equal(uo, u[want]) and np.array_equal(co, c[want])


def test_device_group_sampler_equals_its_oracle_exactly():
    """GroupSampler.sample_device (nncf_group_sample): the groups are the B1 sampler's stream (bit-exact vs the C
    oracle of nodesampler.cpp), the member picks equal the numpy restatement of the same counter-based hash; every
    sampled link is a training link."""
    import torch
    from nncf_amd.data_utils import GroupSampler, degree_dist
    from oracle.sampler_oracle import OracleSampler
    from oracle.shuffle_oracle import group_sample_rows
    rs = np.random.RandomState(2)
    n, nu, ni = 5000, 200, 90
    train = np.unique(np.stack([rs.randint(0, nu, n), rs.zipf(1.4, n) % ni, np.ones(n, dtype=np.int64)], 1), axis=0)
    links = set(map(tuple, train[:, :2].tolist()))
    for by, chop, b in (("item", 4, 64), ("item", 5, 64), ("user", 3, 32)):
        gs = GroupSampler(train, group_by=by, chop=chop, neg_dist="unigram", neg_sign=-1)
        smp = gs.group_sampler.__self__
        smp.state = 4242
        steps = 7
        uo, co = [t.cpu().numpy() for t in gs.sample_device(b, steps, seed=13)]
        assert uo.shape == (steps * b,)
        gc = 1 if by == "item" else 0
        per = -(-b // chop)
        want_groups = OracleSampler(degree_dist(train, gc, "unigram"), 1.0, 4242).sample_batch(steps * per)
        ptr, mem = [t.cpu().numpy() for t in gs._dev_csr]
        wu, wc = group_sample_rows(want_groups, chop, ptr, mem, 13, by_user=(by == "user"))
        wu = wu.reshape(steps, per * chop)[:, :b].reshape(-1)
        wc = wc.reshape(steps, per * chop)[:, :b].reshape(-1)
        assert np.array_equal(uo, wu) and np.array_equal(co, wc), (by, chop)
        assert all((int(u), int(c)) in links for u, c in zip(uo, co))
        grp = co if by == "item" else uo
        assert np.all(grp.reshape(steps, b)[:, :chop] == grp.reshape(steps, b)[:, :1])     # runs of one group


@pytest.mark.parametrize("neg_dist", ["unigram", "uniform"])
def test_device_sample_with_negs_equals_its_oracle_and_the_reference_semantics(neg_dist):
    """GroupSampler.sample_with_negs_device (nncf_group_sample_with_negs; configs/data_utils.py:335-408): groups and
    negative members are the two B1 samplers' streams (bit-exact vs the C oracle of nodesampler.cpp), everything else
    equals the numpy restatement of the same counter-based hashes; and the reference's semantics hold: positives first
    with label 1, every positive a training link, runs of `chop` positives per group, the negatives of a group follow
    its positives' group order, every batch exactly b (1 + k) rows, E[#negatives per group] = k chop p_n / p_d."""
    import torch
    from nncf_amd.data_utils import GroupSampler, degree_dist
    from oracle.sampler_oracle import OracleSampler
    from oracle.shuffle_oracle import group_sample_with_negs_rows
    rs = np.random.RandomState(2)
    n, nu, ni = 5000, 200, 90
    train = np.unique(np.stack([rs.randint(0, nu, n), rs.zipf(1.4, n) % ni, np.ones(n, dtype=np.int64)], 1), axis=0)
    links = set(map(tuple, train[:, :2].tolist()))
    for by, chop, b, k in (("item", 4, 64, 3), ("item", 5, 64, 2), ("user", 3, 32, 4)):
        gs = GroupSampler(train, group_by=by, chop=chop, neg_dist=neg_dist, neg_sign=-1)
        g_smp, m_smp = gs.group_sampler.__self__, gs.member_neg_sampler.__self__
        g_smp.state, m_smp.state = 4242, 777
        steps = 9
        uo, co, yo = [t.cpu().numpy() for t in gs.sample_with_negs_device(b, k, steps, seed=13)]
        whole = b * (1 + k)
        assert uo.shape == (steps * whole,)
        gc, mc = (1, 0) if by == "item" else (0, 1)
        gcap, ncap = gs.negs_caps(b, k)
        assert (gcap == -(-b // chop)) == (neg_dist == "unigram")
        want_groups = OracleSampler(degree_dist(train, gc, "unigram"), 1.0, 4242).sample_batch(steps * gcap)
        want_negs = OracleSampler(degree_dist(train, mc, neg_dist), 0.75, 777).sample_batch(steps * ncap)
        ptr, mem = [t.cpu().numpy() for t in gs._dev_csr]
        pnd = None
        if neg_dist == "uniform":
            pnd = np.ones(len(ptr) - 1)
            for g, v in gs.p_n_div_p_d.items():
                pnd[g] = v
        wu, wc, wy, used = group_sample_with_negs_rows(want_groups, want_negs, gcap, ncap, b, k, chop, ptr, mem, pnd, 13,
                                                       by_user=(by == "user"), neg_sign=-1)
        assert np.array_equal(uo, wu) and np.array_equal(co, wc) and np.array_equal(yo, wy), (by, chop)
        assert np.array_equal(gs.last_groups_used.cpu().numpy(), used) and used.min() > 0
        for t in range(steps):
            sl = slice(t * whole, (t + 1) * whole)
            y, u, c = yo[sl], uo[sl], co[sl]
            npos = int((y == 1).sum())
            # positives first; under 'uniform' popular groups get few negatives, the rounds add groups and the positives alone
            # can fill the batch (the reference cuts [positives ; negatives] the same way, :404-406)
            assert npos == min(used[t] * chop, whole) and np.all(y[:npos] == 1) and np.all(y[npos:] == -1)
            npos -= npos % chop
            assert all((int(a), int(bb)) in links for a, bb in zip(u[:npos], c[:npos]))
            grp = (c if by == "item" else u)
            assert np.all(grp[:npos].reshape(-1, chop) == grp[:npos].reshape(-1, chop)[:, :1])       # runs of one group
            order = grp[:npos:chop]
            neg_groups = grp[npos:]
            first = [int(np.flatnonzero(order == g)[0]) for g in neg_groups]
            assert first == sorted(first) or len(set(order)) < len(order)     # negatives in the groups' order
        if neg_dist == "unigram":
            assert np.all(used == -(-b // chop))
    # the negatives' count per group is k chop p_n / p_d in expectation (random rounding, :345)
    gs = GroupSampler(train, group_by="item", chop=2, neg_dist="uniform", neg_sign=0)
    x = np.array([3 * 2 * gs.p_n_div_p_d[g] for g in sorted(gs.p_n_div_p_d)])
    assert np.any(np.abs(x - np.round(x)) > 0.05)


@pytest.mark.parametrize("seed", range(int(os.environ.get("NNCF_SOAK_SAMPLER", "12"))))
def test_random_distributions_powers_and_batch_sizes_are_bit_exact(seed):
    """Random vocabularies (2 ... 60,000 entries; counts that are uniform, Zipf-like, mostly zero, one giant entry), powers
    0 / 0.5 / 0.75 / 1, start states and ragged batch sizes: boundaries, every draw and the state after, bit for bit against
    the oracle (sampler.h's table + LCG).  NNCF_SOAK_SAMPLER widens the sweep (tools/soak.sh)."""
    rs = np.random.RandomState(17000 + seed)
    V = int(rs.choice([2, 3, 17, 100, 1000, 8001, 60000]))
    kind = int(rs.randint(4))
    if kind == 0:
        dist = np.ones(V)
    elif kind == 1:
        dist = np.floor(1.0 + rs.pareto(1.1, size=V) * 2.0)
    elif kind == 2:
        dist = rs.randint(0, 3, size=V).astype(np.float64) * (rs.rand(V) < 0.2)
        dist[rs.randint(V)] = 1.0      # at least one entry can be drawn
    else:
        dist = rs.randint(1, 5, size=V).astype(np.float64)
        dist[rs.randint(V)] = 1e7
    power = float(rs.choice([0.0, 0.5, 0.75, 1.0]))
    state0 = int(rs.randint(1, 2 ** 31 - 1))
    o = so.OracleSampler(dist, power, state0)
    s = MultinomialSampler(dist, dist.size, power, state0)
    assert np.array_equal(s.bounds(), o.bounds()), (V, kind, power)
    for n in [int(x) for x in rs.choice([0, 1, 2, 63, 64, 65, 255, 257, 1000, 5121, 30011], size=5)]:
        assert np.array_equal(s.sample_batch(n), o.sample_batch(n)), (V, kind, power, n)
        if n % 2:
            assert s.sample() == o.sample()
    assert s.state == o.state
    s.close()
    o.close()
