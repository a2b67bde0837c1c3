#!/usr/bin/env python3
"""whole@50 evaluation pass (train + test candidate sets) at the BASELINE shape: the one-kernel evaluator (nncf_eval_whole)
against the two-kernel path (nncf_predict_scores + nncf_eval_topk).  Prints one JSON line with a roofline object for the
fused kernel: 2 x users x candidates x d flop on the fp32 matrix pipe (157.3 TFLOP/s)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _traffic(d):
    """HBM bytes per pass from profiles/traffic.json (tools/make_traffic.py) while the kernel sources are the profiled ones"""
    try:
        ent = json.load(open(os.path.join(ROOT, "profiles", "traffic.json"))).get("evaluator/whole@50/16Kx17K/d%d" % d) or {}
        import __graft_entry__ as ge
        return ent.get("bytes_per_pass") if ent.get("csrc_hash") == ge._source_hash() else None
    except Exception:
        return None


def main():
    import torch
    from nncf_amd import synthetic
    from nncf_amd import engine as nengine
    from nncf_amd.conf import get_conf
    from nncf_amd.data_utils import get_data
    from nncf_amd.model_framework import get_model
    from nncf_amd.objectives import Evaluator
    d = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    data = synthetic.make_citeulike_like()
    pd = {"reset_after_getconf": True, "max_epoch": 1, "loss": "skip-gram", "num_negatives": 0, "neg_loss_weight": 16, "loss_gamma": 10,
          "learn_rate": 0.1, "batch_size_p": 512, "user_dim": d, "item_dim": d, "chop_size": 5, "eval_topk": 50, "optimizer": "sgd"}
    conf = get_conf("synthetic", "best", pd, model_choice="mf")
    helper = get_data(dict(data), conf, rand_seed=5)
    md = get_model(conf, helper, "mf", seed=4)
    head = md["model_neg_shared"]
    ev = Evaluator(helper, helper.data_spec, conf)
    out = {"metric": "whole@50 evaluation pass (train + test candidates), 16K users x 17K items, d=%d" % d, "unit": "ms"}
    n_cand = len(ev.train_items) + len(ev.test_items)
    flop = 2.0 * ev.user_count * n_cand * d
    for name, fused in (("fused", True), ("two_kernel", False)):
        for which in ("train", "test"):
            ev._whole(head, which, 50, fused=fused)
        torch.cuda.synchronize()
        reps = 10
        kern = "eval_fused_kernel" if fused else "predict_tile_kernel"
        nengine.profile_begin(kern, 4000)
        t0 = time.perf_counter()
        for _ in range(reps):
            for which in ("train", "test"):
                ev._whole(head, which, 50, fused=fused)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / reps
        raw_ms, launches, ov_ms = nengine.profile_end()
        # profile_end: MEAN bracket per launch and the empty bracket's cost
        out[name] = {"pass_ms_wall": wall * 1e3, "kernel_ms_per_pass": (raw_ms - ov_ms) * launches / reps, "launches_per_pass": launches / reps}
    k_ms = out["fused"]["kernel_ms_per_pass"]
    out["roofline"] = {"kernel": "eval_fused_kernel", "bound": "mfma", "achieved": flop / (k_ms * 1e-3) / 1e12, "peak": 157.3, "unit": "TFLOP/s",
                       "frac": flop / (k_ms * 1e-3) / 1e12 / 157.3, "traffic": _traffic(d),
                       "work": "2 x %d users x %d candidates x %d = %.1f GFLOP per pass (fp32 MFMA, exact fp32)" % (ev.user_count, n_cand, d, flop / 1e9)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
