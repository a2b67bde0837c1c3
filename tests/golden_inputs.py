"""Seeded inputs of the loss-golden cases (tests/golden/loss_golden.npz): shared by the generator
(tools/gen_golden_losses.py, which runs the reference's own loss code on them) and by the tests,
so that the large cases need not store their tables."""
import numpy as np


def case_inputs(seed, family, b, k, d, nu, ni, scale):
    """(U, V, uid, cid, ubias, cbias): float64 tables / biases and int64 batch ids laid out as the
    reference's trainers feed them (train_original.py:49-56, train_neg_shared.py:46-49,
    train_group_neg_shared.py:53-59, train_sampled_neg_shared.py:46-52), with duplicate users
    and (Zipf-like) duplicate items."""
    rs = np.random.RandomState(seed)
    U = (rs.rand(nu, d) * 2 - 1) * scale
    V = (rs.rand(ni, d) * 2 - 1) * scale
    uid = rs.randint(0, nu, b)
    cid = np.minimum((ni * rs.power(0.5, b)).astype(np.int64), ni - 1)
    if family == "group_neg_shared":   # stratified batches: runs of one item
        chop = 2 if b <= 8 else 4
        cid = np.repeat(cid[:(b + chop - 1) // chop], chop)[:b]
        if len(np.unique(cid)) < 2:   # n_u = 1 makes the reference divide by zero (objectives.py:203)
            cid[b // 2:] = (cid[0] + 1) % ni
    if family == "original":
        uid = np.concatenate([uid, np.repeat(uid, k)])
        cid = np.concatenate([cid, rs.randint(0, ni, b * k)])
    if family == "sampled_neg_shared":
        uid = np.concatenate([uid, uid[:k]])
        cid = np.concatenate([cid, rs.randint(0, ni, k)])
    ub = (rs.rand(nu) - 0.5) * 0.2
    cb = (rs.rand(ni) - 0.5) * 0.2
    return U, V, uid.astype(np.int64), cid.astype(np.int64), ub, cb


def parse_meta(entry):
    f = str(entry).split("|")
    return dict(family=f[0], loss=f[1], tag=f[2], b=int(f[3]), k=int(f[4]), d=int(f[5]), nu=int(f[6]), ni=int(f[7]),
                scale=float(f[8]), lam=float(f[9]), gamma=float(f[10]), u_reg=float(f[11]), normalize=bool(int(f[12])),
                mf=bool(int(f[13])), bias=f[14] or None, full=bool(int(f[15])), seed=int(f[16]))
