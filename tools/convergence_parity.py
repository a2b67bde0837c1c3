"""Whole-run parity record at the BASELINE shape: the product Trainer (HIP) and the reference's trainer loop
on the float64 oracle (oracle/epoch_oracle.py), same links / seeds / initial parameters / dropout masks, for --epochs epochs:
per-step loss differences, both Recall@50 / MAP@50 curves (train candidates and cold test items), final-table errors against
the term-count bounds of tests/tolerances.py.  Writes one JSON (default profiles/r06_convergence_parity.json).

  python tools/convergence_parity.py --epochs 30 --optimizer adam --eval-epochs 1,2,5,10,20,30

Checker tool: imports oracle/ and tests/epoch_parity.py; not part of the product."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="basic_embedding", choices=["basic_embedding", "mf", "cnn_embedding"])
    ap.add_argument("--scheme", default="neg_shared", choices=["neg_shared", "group_neg_shared", "original"])
    ap.add_argument("--loss", default=None)
    ap.add_argument("--optimizer", default="adam", choices=["sgd", "adam", "sparse_adam"])
    ap.add_argument("--learn-rate", type=float, default=None)
    ap.add_argument("--epochs", type=int, default=30)
    ap.add_argument("--eval-epochs", default=None, help="comma list of epochs at which the ORACLE is evaluated (default: all)")
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r06_convergence_parity.json"))
    ap.add_argument("--tag", default=None)
    ap.add_argument("--max-steps", type=int, default=None, help="train on the first max-steps * b links only (cnn_embedding: the float64 conv oracle takes seconds per step)")
    ap.add_argument("--control-fp32", action="store_true", help="also run the oracle loop in float32: the drift precision alone causes")
    args = ap.parse_args()
    import __graft_entry__
    __graft_entry__.build(quiet=True)
    from nncf_amd import synthetic
    from tests import epoch_parity as ep
    loss = args.loss or ("log-loss" if args.scheme == "group_neg_shared" else "skip-gram")
    data = synthetic.make_citeulike_like(content_len=200, vocab=8001)
    ev = [int(v) for v in args.eval_epochs.split(",")] if args.eval_epochs else None
    t0 = time.time()
    res = ep.run_case(data, model=args.model, scheme=args.scheme, loss=loss, optimizer=args.optimizer, epochs=args.epochs,
                      eval_epochs=ev, learn_rate=args.learn_rate, control_fp32=args.control_fp32, max_steps=args.max_steps, log=lambda *a: print("[parity]", *a, file=sys.stderr, flush=True))
    conf = res["conf"]
    b, k = conf.batch_size_p, conf.num_negatives
    rows = b * (1 + k)
    L = data["C"].shape[1]
    terms = {"U": rows, "V": rows, "W": rows * L, "Wd": rows, "bd": rows, "gamma": rows, "beta": rows,
             "Wc": rows * L, "bc": rows * L, "g1": rows * L, "b1": rows * L, "rm1": rows * L, "rv1": rows * L}
    updates = args.epochs * res["n_steps"]
    n = res["n_steps"]
    g, w = res["got_losses"], res["want_losses"]
    rel = np.abs(g - w) / np.maximum(1.0, np.abs(w))
    per_epoch = [{"epoch": e + 1, "mean_cost_product": res["got_cost"][e + 1], "mean_cost_oracle": res["want_cost"][e + 1],
                  "max_rel_step_loss_diff": float(rel[e * n:(e + 1) * n].max())} for e in range(args.epochs)]
    out = {
        "what": "product Trainer (HIP, fp32) vs the reference trainer loop on the float64 oracle; identical links, NumPy seed, "
                "initial parameters and dropout masks",
        "workload": "CiteULike-a-shaped synthetic 16K users x 17K items x 200K links, item text L=200, 8,001 words; %s d=100, %s, %s, "
                    "b=512%s, word-row dropout %.2g, optimizer %s(%g)" % (args.model, args.scheme, loss, (", k=%d" % k) if k else "",
                                                                           conf.word_emb_dropout_rate, args.optimizer, conf.learn_rate),
        "epochs": args.epochs, "steps_per_epoch": n,
        "step_loss": ep.loss_report(res), "per_epoch_cost": per_epoch,
        "metrics_at_epochs": ep.metric_report(res),
        "final_tables_vs_term_count_bound": ep.table_report(res, terms, updates),
        "control_float32_oracle_vs_float64_oracle": ep.control_report(res),
        "product_train_seconds_per_epoch": float(np.mean(res["trainer"].train_time)),
        "wall_seconds": time.time() - t0,
    }
    if args.tag:
        out["tag"] = args.tag
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    with open(args.out, "w") as fh:
        json.dump(out, fh, indent=1)
    print(json.dumps({k2: out[k2] for k2 in ("workload", "step_loss", "final_tables_vs_term_count_bound")}, indent=1))
    for e, row in out["metrics_at_epochs"].items():
        print("epoch %2d: train recall@50 %.6f / %.6f (diff %+.2e)   test %.6f / %.6f (diff %+.2e)" % (
            e, row["train_recall@50"]["product"], row["train_recall@50"]["oracle"], row["train_recall@50"]["diff"],
            row["test_recall@50"]["product"], row["test_recall@50"]["oracle"], row["test_recall@50"]["diff"]))


if __name__ == "__main__":
    main()
