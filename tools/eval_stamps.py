#!/usr/bin/env python3
"""Where a stage of the one-kernel evaluator goes (developer build: VARIANT_FILES=eval tools/build_variant.sh evalstamps
"-DNNCF_EVAL_STAMPS=1", copied over nncf_amd/libnncf_hip.so).  Every wave of workgroup 0 adds up the shader cycles of each phase of
its stages; printed per stage (mean over the sweep) and wave."""
import ctypes
import os
import sys

import numpy as np

os.environ["NNCF_DEBUG_TIMING"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from nncf_amd import _lib, synthetic
    from nncf_amd.conf import get_conf
    from nncf_amd.data_utils import get_data
    from nncf_amd.model_framework import get_model
    from nncf_amd.objectives import Evaluator
    data = synthetic.make_citeulike_like()
    pd = {"reset_after_getconf": True, "max_epoch": 1, "loss": "skip-gram", "num_negatives": 0, "neg_loss_weight": 16, "loss_gamma": 10,
          "learn_rate": 0.1, "batch_size_p": 512, "user_dim": 100, "item_dim": 100, "chop_size": 5, "eval_topk": 50, "optimizer": "sgd"}
    conf = get_conf("synthetic", "best", pd, model_choice="mf")
    helper = get_data(dict(data), conf, rand_seed=5)
    md = get_model(conf, helper, "mf", seed=4)
    head = md["model_neg_shared"]
    ev = Evaluator(helper, helper.data_spec, conf)
    eng = head.engine
    names = ["B fragment in registers", "products done", "ranking (+ appends) done", "next stage stored, loads issued", "barrier", "cut + barrier + refresh"]
    for which, items in (("train", ev.train_items), ("test", ev.test_items)):
        ev._whole(head, which, 50, fused=True)
        torch.cuda.synchronize()
        nw = int(os.environ.get("NNCF_EVAL_WAVES", "16"))
        out = np.zeros(8 * nw, dtype=np.int64)
        _lib.check(eng._L.nncf_model_debug_stamps(eng._h, out.ctypes.data_as(ctypes.POINTER(ctypes.c_longlong)), out.size))
        out = out.reshape(nw, 8)
        nst = (len(items) + 63) // 64
        print("## %s candidates: %d stages; shader cycles per stage and wave (workgroup 0)" % (which, nst))
        print("%-34s" % "phase" + "".join("%7s" % ("w%d" % w) for w in range(nw)) + "    mean")
        for k, nm in enumerate(names):
            print("%-34s" % nm + "".join("%7.0f" % (out[w, k] / nst) for w in range(nw)) + "%8.0f" % (out[:, k].mean() / nst))
        print("%-34s" % "sum" + "".join("%7.0f" % (out[w, :6].sum() / nst) for w in range(nw)) + "%8.0f" % (out[:, :6].sum(axis=1).mean() / nst))


if __name__ == "__main__":
    main()
