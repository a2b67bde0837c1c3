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
