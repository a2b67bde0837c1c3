#!/usr/bin/env python3
"""Generate tests/golden/host_golden.npz by RUNNING reference Python code in this container
(needs /root/reference): only inputs and the outputs the reference produced are stored.

  * eval_multiple (utils/metrics_ranking.py:6-35): imported with a shim that maps the absent
    `bottleneck.argpartition` to numpy's; tie-free scores so the random tie-break noise is inert.
  * group_shuffle_train (configs/data_utils.py:218-241): the module itself cannot be imported
    (py2 syntax elsewhere, cPickle/numba/sampler imports), so the function's source lines are
    exec'd as-is into a namespace holding numpy.  chop = 0 only: with chop > 0 the output
    depends on the within-group order of the reference's UNSTABLE argsort (:234), which is
    NumPy-build specific.
"""
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ref_integrity  # noqa: E402


def ref_eval_multiple():
    sys.modules["bottleneck"] = types.SimpleNamespace(argpartition=np.argpartition)
    ref_integrity.read_reviewed("utils/metrics_ranking.py")      # hashed before it is imported
    sys.path.insert(0, os.path.join(REF, "utils"))
    import metrics_ranking
    return metrics_ranking.eval_multiple


def ref_group_shuffle():
    lines = ref_integrity.lines("configs/data_utils.py")
    src = "\n".join(lines[217:241])  # def group_shuffle_train ... return train
    ns = {"np": np}
    exec(compile(src, "ref_group_shuffle_train", "exec"), ns)
    return ns["group_shuffle_train"]


def ref_group_shuffle_stable():
    """The same source lines with ONE token changed in memory: `.argsort()` (:234, an unstable quicksort whose
    within-group order is NumPy-build specific) -> `.argsort(kind='stable')`, the documented deviation of the build.
    With it the chop > 0 outputs are reproducible, and the build's function must equal them exactly."""
    lines = ref_integrity.lines("configs/data_utils.py")
    src = "\n".join(lines[217:241])
    assert src.count(".argsort()") == 1
    ns = {"np": np}
    exec(compile(src.replace(".argsort()", ".argsort(kind='stable')"), "ref_group_shuffle_train_stable", "exec"), ns)
    return ns["group_shuffle_train"]


def main():
    out = {}
    ev = ref_eval_multiple()
    rs = np.random.RandomState(0)
    cases = []
    for c in range(12):
        n = int(rs.randint(60, 400))
        topk = int(rs.choice([5, 10, 50]))
        pred = rs.permutation(n).astype(np.float64) / n + rs.rand() * 1e-3  # tie-free
        truth = (rs.rand(n) < rs.choice([0.0, 0.02, 0.1, 0.5])).astype(np.int32)
        np.random.seed(c)
        ap, rec, prec = ev(truth, pred, topk)
        out["ev%d_pred" % c] = pred
        out["ev%d_truth" % c] = truth
        out["ev%d_topk" % c] = np.array([topk])
        out["ev%d_out" % c] = np.array([ap, rec, prec], dtype=np.float64)
        cases.append(c)
    out["ev_cases"] = np.array(cases)

    gs = ref_group_shuffle()
    for c, (n, nu, ni, by) in enumerate([(2000, 300, 120, "item"), (777, 50, 40, "item"), (500, 30, 200, "user")]):
        rs = np.random.RandomState(100 + c)
        train = np.stack([rs.randint(0, nu, n), rs.randint(0, ni, n), np.ones(n, dtype=np.int64)], 1).astype(np.int64)
        col = 0 if by == "user" else 1
        iidx = np.arange(train[:, col].max() + 1)
        np.random.seed(1234 + c)
        res = gs(train.copy(), by=by, chop=0, iidx=iidx.copy())
        out["gs%d_train" % c] = train
        out["gs%d_out" % c] = res
        out["gs%d_seed" % c] = np.array([1234 + c])
        out["gs%d_by" % c] = np.array([col])
    out["gs_cases"] = np.arange(3)
    gss = ref_group_shuffle_stable()
    k = 0
    for (n, nu, ni, by) in [(2003, 300, 120, "item"), (777, 50, 40, "user")]:
        for chop in (0, 2, 5):
            rs = np.random.RandomState(200 + k)
            train = np.stack([rs.randint(0, nu, n), rs.randint(0, ni, n), np.ones(n, dtype=np.int64)], 1).astype(np.int64)
            col = 0 if by == "user" else 1
            iidx = np.arange(train[:, col].max() + 1)
            np.random.seed(4321 + k)
            ii = iidx.copy()
            res = gss(train.copy(), by=by, chop=chop, iidx=ii)
            res2 = gss(res.copy(), by=by, chop=chop, iidx=ii)     # a second epoch: iidx persists shuffled (:231)
            out["gst%d_train" % k] = train
            out["gst%d_out" % k] = res
            out["gst%d_out2" % k] = res2
            out["gst%d_cfg" % k] = np.array([4321 + k, col, chop])
            k += 1
    out["gst_cases"] = np.arange(k)
    path = os.path.join(ROOT, "tests", "golden", "host_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KB")


if __name__ == "__main__":
    main()
