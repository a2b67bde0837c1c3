"""TEST INFRASTRUCTURE ONLY -- numpy restatement of the device-side stratified shuffle
(nncf_amd/csrc/shuffle.hip: nncf_group_shuffle), which is group_shuffle_train
(/root/reference/configs/data_utils.py:218-241) with counter-based random numbers instead of NumPy's
Mersenne stream:
  1. rows in a uniformly random order            (np.random.shuffle(train), :232)   -> stable sort by hash32(s0, row)
  2. stable sort by a random key of the row's group (shuffled iidx :231, argsort :234) -> key (hash32(s1, group) << 32 | group)
  3. chop > 0: the first floor(n / chop) * chop rows cut into runs of `chop` rows, runs shuffled as units, the ragged
     tail last (:235-240)                                                            -> key (hash32(s2, run) >> 1, position)
Only tests/ import this; the product never does.  The permutation it returns must equal the kernel's bit for bit
(integer work: tests/test_sampler_gpu.py)."""
import numpy as np

_M = np.uint64(0xFFFFFFFFFFFFFFFF)


def mix64(x):
    x = np.asarray(x, dtype=np.uint64)
    with np.errstate(over="ignore"):
        x = x ^ (x >> np.uint64(30))
        x = x * np.uint64(0xBF58476D1CE4E5B9)
        x = x ^ (x >> np.uint64(27))
        x = x * np.uint64(0x94D049BB133111EB)
        x = x ^ (x >> np.uint64(31))
    return x


def hash32(seed, i):
    with np.errstate(over="ignore"):
        return (mix64(np.uint64(seed) + np.asarray(i, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)) >> np.uint64(32)).astype(np.uint64)


def group_shuffle_perm(uid, cid, by_user, chop, seed):
    """source row of every output row"""
    n = len(uid)
    with np.errstate(over="ignore"):
        s0, s1, s2 = (int(mix64(np.uint64(seed) + np.uint64(k))) for k in range(3))
    rows = np.argsort(hash32(s0, np.arange(n)), kind="stable")
    group = (np.asarray(uid) if by_user else np.asarray(cid)).astype(np.uint64)[rows]
    key = (hash32(s1, group) << np.uint64(32)) | group
    rows = rows[np.argsort(key, kind="stable")]
    if chop > 0:
        bulk = (n // chop) * chop
        i = np.arange(n)
        hi = np.where(i < bulk, hash32(s2, i // chop) >> np.uint64(1), np.uint64(0x80000000))
        key = (hi << np.uint64(32)) | i.astype(np.uint64)
        rows = rows[np.argsort(key, kind="stable")]
    return rows.astype(np.int64)


def group_sample_rows(groups, chop, member_ptr, members, seed, by_user=False):
    """numpy restatement of nncf_group_sample (csrc/shuffle.hip): rows (uid, cid) of GroupSampler.sample
    (/root/reference/configs/data_utils.py:312-330) for the given sampled groups."""
    groups = np.asarray(groups, dtype=np.int64)
    n = len(groups) * chop
    i = np.arange(n)
    g = groups[i // chop]
    p0 = np.asarray(member_ptr)[g].astype(np.uint64)
    ln = (np.asarray(member_ptr)[g + 1] - np.asarray(member_ptr)[g]).astype(np.uint64)
    with np.errstate(over="ignore"):
        s = mix64(np.uint64(seed))
    r = hash32(int(s), i)
    pick = (r * ln) >> np.uint64(32)
    m = np.asarray(members)[(p0 + pick).astype(np.int64)]
    return (g, m) if by_user else (m, g)


def group_sample_with_negs_rows(groups, neg_members, gcap, ncap, b, k, chop, member_ptr, members, pnd, seed, by_user=False, neg_sign=0):
    """numpy restatement of nncf_group_sample_with_negs (csrc/shuffle.hip): the batches of GroupSampler.sample_with_negs
    (/root/reference/configs/data_utils.py:335-408, strict_return_shape=True) for the given sampler draws
    (groups [n_steps * gcap], neg_members [n_steps * ncap]).  Returns (uid, cid, y, groups_used) like the kernel."""
    groups = np.asarray(groups, dtype=np.int64).reshape(-1, gcap)
    negm = np.asarray(neg_members, dtype=np.int64).reshape(-1, ncap)
    ptr, mem = np.asarray(member_ptr, dtype=np.int64), np.asarray(members, dtype=np.int64)
    with np.errstate(over="ignore"):
        s0, s1 = int(mix64(np.uint64(seed))), int(mix64(np.uint64(seed) + np.uint64(1)))
    whole = b * (1 + k)
    U, C, Y, used_out = [], [], [], []
    for t in range(groups.shape[0]):
        g = groups[t]
        x = float(k) * float(chop) * (np.ones(gcap) if pnd is None else np.asarray(pnd, dtype=np.float64)[g])
        fl = x.astype(np.int64)
        u = hash32(s1, np.uint64(t) * np.uint64(gcap) + np.arange(gcap, dtype=np.uint64)).astype(np.float64) / 4294967296.0
        P = np.cumsum(fl + (u < x - fl))
        used, ok = -(-b // chop), True
        for rnd in range(12):
            if used > gcap:
                ok, used = False, gcap
                break
            diff = whole - (used * chop + P[used - 1])
            if diff <= 0:
                break
            if rnd >= 10:
                ok = False
                break
            used += diff if diff <= 100 else diff // (chop * (1 + k)) + 100
        npos = used * chop
        r = np.arange(whole)
        pos = r < npos
        grp = np.empty(whole, dtype=np.int64)
        mm = np.zeros(whole, dtype=np.int64)
        rp = r[pos]
        grp[pos] = g[rp // chop]
        p0 = ptr[grp[pos]].astype(np.uint64)
        ln = (ptr[grp[pos] + 1] - ptr[grp[pos]]).astype(np.uint64)
        h = hash32(s0, np.uint64(t) * np.uint64(gcap) * np.uint64(chop) + rp.astype(np.uint64))
        mm[pos] = mem[(p0 + ((h * ln) >> np.uint64(32))).astype(np.int64)]
        q = r[~pos] - npos
        grp[~pos] = g[np.searchsorted(P[:used], q, side="right").clip(max=used - 1)]
        if q.size and q.max() >= ncap:
            ok = False
        mm[~pos] = negm[t][np.minimum(q, ncap - 1)]
        U.append(grp if by_user else mm)
        C.append(mm if by_user else grp)
        Y.append(np.where(pos, 1, neg_sign))
        used_out.append(used if ok else -1)
    return np.concatenate(U), np.concatenate(C), np.concatenate(Y), np.asarray(used_out)
