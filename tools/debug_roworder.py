#!/usr/bin/env python3
"""Developer check: tests/test_tower_gpu.py::test_row_order_does_not_matter_at_baseline_shape[basic_embedding], printing the
differences between the positives-first and the row-wise path instead of asserting them."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from nncf_amd import synthetic
    from nncf_amd.conf import get_conf
    from nncf_amd.data_utils import get_data
    from nncf_amd.model_framework import get_model
    from oracle import step_oracle as so
    data = synthetic.make_citeulike_like(content_len=200, vocab=8001)
    pd = {"reset_after_getconf": True, "max_epoch": 1, "loss": "skip-gram", "num_negatives": 10, "neg_loss_weight": 128,
          "loss_gamma": 10, "learn_rate": 0.01, "neg_dist": "unigram", "neg_sampling_power": 1, "batch_size_p": 512,
          "user_dim": 100, "item_dim": 100, "word_dim": 100, "chop_size": 5, "u_reg": 1e-6, "word_emb_dropout_rate": 0.,
          "interaction_bias": None}
    conf = get_conf("synthetic", "best", dict({"optimizer": "sgd"}, **pd), model_choice="basic_embedding")
    mds = [get_model(conf, get_data(dict(data), conf, rand_seed=5), "basic_embedding", seed=2) for _ in range(2)]
    b, k = 512, 10
    rs = np.random.RandomState(3)
    n_items = data["C"].shape[0]
    for step in range(int(sys.argv[1]) if len(sys.argv) > 1 else 2):
        bt = data["train"][rs.choice(len(data["train"]), b, replace=False)]
        u, c, y = so.assemble_original(bt, rs.randint(0, n_items, b * k), "skip-gram")
        perm = rs.permutation(b * (1 + k))
        l0 = mds[0]["model"].train_on_batch([u, c], [y])
        l1 = mds[1]["model"].train_on_batch([u[perm], c[perm]], [y[perm]])
        t0, t1 = mds[0]["model"].core.tower, mds[1]["model"].core.tower
        line = "step %d loss %.9g / %.9g" % (step, l0, l1)
        for name in ("W", "Wd", "gamma", "beta", "bd"):
            a, bb = getattr(t0, name).cpu().numpy(), getattr(t1, name).cpu().numpy()
            d = np.abs(a - bb)
            i = int(np.argmax(d))
            line += "  %s max|d| %.3g (value %.3g)" % (name, d.max(), a.ravel()[i])
        print(line)


if __name__ == "__main__":
    main()
