#!/usr/bin/env python3
"""Is a training trajectory sensitive to rounding?  The product Trainer runs one epoch of a configuration TWICE with two summation
orders of the same arithmetic (cnn_embedding: NNCF_CNN_SPLIT = 0 / 64 -- the conv backward over whole items or over halves of the
word-rich ones) and the per-step losses and final parameters of the two runs are compared with each other.  No oracle involved:
what the two fp32 runs disagree by is what rounding alone does to this configuration over an epoch."""
import contextlib
import importlib
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(model, scheme, loss, out):
    import torch
    from nncf_amd import synthetic
    from nncf_amd.conf import get_conf
    from nncf_amd.data_utils import get_data
    from nncf_amd.model_framework import get_model
    from tests.epoch_parity import CNN_PARAMS, HEAD_OF, TOWER_PARAMS, param_dict
    data = synthetic.make_citeulike_like(content_len=200, vocab=8001)
    conf = get_conf("synthetic_citeulike", "best", param_dict(model, scheme, loss, "sgd", 1), model_choice=model)
    conf.eval_topk = 50
    with contextlib.redirect_stdout(open(os.devnull, "w")):
        helper = get_data(dict(data), conf, rand_seed=12345, device=0)
        md = get_model(conf, helper, model, device=0, seed=0)
        head = md[HEAD_OF[scheme]]
        losses = []
        inner = head.run_epoch

        def rec(*a, **kw):
            o = inner(*a, **kw)
            losses.append(o.detach().double().cpu().numpy().copy())
            return o
        head.run_epoch = rec
        np.random.seed(4224)
        tr = importlib.import_module("nncf_amd.train_" + scheme).Trainer(md, conf, helper)
        tr.train("whole")
    torch.cuda.synchronize()
    tw = head.core.tower
    names = CNN_PARAMS if model == "cnn_embedding" else TOWER_PARAMS
    np.savez(out, losses=np.concatenate([x for x in losses if len(x)]), U=head.engine.U.cpu().numpy(),
             **{n: getattr(tw, n).cpu().numpy() for n in names})


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        return child(*sys.argv[2:6])
    model, scheme, loss = (sys.argv[1:4] + ["cnn_embedding", "group_neg_shared", "log-loss"][len(sys.argv) - 1:])[:3]
    runs = []
    for tag, env in (("a", {"NNCF_CNN_SPLIT": "0"}), ("b", {"NNCF_CNN_SPLIT": "64"})):
        out = "/tmp/chaos_%s.npz" % tag
        subprocess.run([sys.executable, os.path.abspath(__file__), "--child", model, scheme, loss, out], env=dict(os.environ, **env), check=True, cwd=ROOT)
        runs.append(np.load(out))
    a, b = runs
    rel = np.abs(a["losses"] - b["losses"]) / np.maximum(1.0, np.abs(a["losses"]))
    over = np.nonzero(rel > 1e-5)[0]
    rep = {"what": "two fp32 runs of the product Trainer, one epoch, that differ ONLY in a summation order (NNCF_CNN_SPLIT = 0 / 64)",
           "workload": "%s %s %s, CiteULike-a-shaped synthetic, b = 512, sparse SGD" % (model, scheme, loss), "steps": int(rel.size),
           "max_rel_step_loss_diff": float(rel.max()), "first_step_over_1e-5": int(over[0]) if over.size else None,
           "rel_diff_at_steps_50_100_150_200_300_last": [float(rel[min(i, rel.size - 1)]) for i in (50, 100, 150, 200, 300, rel.size - 1)],
           "max_abs_param_diff": {n: float(np.abs(a[n] - b[n]).max()) for n in a.files if n != "losses"}}
    print(json.dumps(rep, indent=1))


if __name__ == "__main__":
    main()
