#!/usr/bin/env python3
"""tests/golden/group_sampler_golden.npz: outputs of the REFERENCE's GroupSampler
(configs/data_utils.py:244-408) run in this container.  The module cannot be imported (py2 print
statements, cPickle/numba/Cython imports), so the class's source lines are exec'd from where they
lie with the two py2 `print` statements inside them blanked, into a namespace whose `get_sampler`
is the reference's own function body's equivalent over the COMPILED reference sampler
(oracle/_ref, sampler/nodesampler.cpp): degree histogram -> MultinomialSampler -> sample_batch.
Only inputs, seeds and the produced batches are stored."""
import os
import random
import re
import sys
from collections import defaultdict

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference"
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ref_integrity  # noqa: E402

from oracle import sampler_oracle as so  # noqa: E402


def ref_get_sampler(ratings, neg_dist="unigram", neg_sampling_power=0.75, column=1, rand_seed=0, batch_mode=True):
    neg_dist = neg_dist.split("_")[0]
    dist = np.bincount(ratings[:, column].astype(np.int64)).astype(float)
    if neg_dist == "uniform":
        dist[dist > 0] = 1
    s = so.RefSampler(dist, neg_sampling_power, rand_seed)
    return s.sample_batch


def count_jit(x, dist):
    for v in x:
        dist[v] += 1


def ref_class():
    lines = ref_integrity.lines("configs/data_utils.py")
    start = next(i for i, l in enumerate(lines) if l.startswith("class GroupSampler"))
    src = "\n".join(lines[start:])
    src = re.sub(r"print '\[INFO\] In GroupSampler, setting neg_sampling_power' \\\n\s+'to user defined \{\}'\.format\(neg_sampling_power\)",
                 "pass", src)
    ns = {"np": np, "random": random, "defaultdict": defaultdict, "get_sampler": ref_get_sampler, "count_jit": count_jit}
    exec(compile(src, "ref_GroupSampler", "exec"), ns)
    return ns["GroupSampler"]


def main():
    GS = ref_class()
    out = {}
    rs = np.random.RandomState(7)
    n, nu, ni = 3000, 120, 60
    pop = rs.zipf(1.5, n) % ni
    train = np.stack([rs.randint(0, nu, n), pop, np.ones(n, dtype=np.int64)], 1).astype(np.int64)
    out["train"] = train
    case = 0
    for neg_dist in ("unigram", "uniform", "uniform_no_correction"):
        for chop in (1, 4):
            gs = GS(train, group_by="item", chop=chop, neg_dist=neg_dist, neg_sign=-1)
            np.random.seed(100 + case)
            random.seed(200 + case)
            a = gs.sample(64)
            b = gs.sample(50)
            c = gs.sample_with_negs(32, 3)
            d = gs.sample_with_negs(40, 2)
            out["c%d_cfg" % case] = np.array([chop, {"unigram": 0, "uniform": 1, "uniform_no_correction": 2}[neg_dist]])
            for name, v in (("a", a), ("b", b), ("c", c), ("d", d)):
                out["c%d_%s" % (case, name)] = np.asarray(v, dtype=np.int64)
            case += 1
    gs = GS(train, group_by="user", chop=3, neg_dist="unigram", neg_sign=0)
    np.random.seed(999)
    random.seed(998)
    out["u_a"] = np.asarray(gs.sample(30), dtype=np.int64)
    out["u_c"] = np.asarray(gs.sample_with_negs(30, 2), dtype=np.int64)
    out["n_cases"] = np.array([case])
    path = os.path.join(ROOT, "tests", "golden", "group_sampler_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KB")


if __name__ == "__main__":
    main()
