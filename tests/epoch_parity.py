"""Whole-epoch parity at the BASELINE shape: the product `Trainer` (nncf_amd.train_*; HIP kernels behind the C ABI) against the
reference's trainer loop written out over the float64 oracle (oracle/epoch_oracle.py), on identical inputs: same links, same
NumPy seed => same shuffles, same initial parameters, the product's dropout masks fed to the oracle.

Shared by tests/test_epoch_parity_gpu.py (one epoch, asserted) and tools/convergence_parity.py (30 epochs, recorded).
Checker code: imports oracle/; nothing under nncf_amd/ imports this."""
import contextlib
import importlib
import os
import sys
import time

import numpy as np

from oracle import epoch_oracle as eo
from oracle import step_oracle as so
from tests.tolerances import sum_bound

HEAD_OF = {"original": "model", "neg_shared": "model_neg_shared", "group_neg_shared": "model_group_neg_shared"}
TOWER_PARAMS = ("W", "Wd", "bd", "gamma", "beta", "rm", "rv")
CNN_PARAMS = ("W", "Wc", "bc", "g1", "b1", "Wd", "bd", "gamma", "beta", "rm", "rv", "rm1", "rv1")   # cnn_embedding (one conv layer: the 'best' preset)


def param_dict(model, scheme, loss, optimizer, epochs, b=512, d=100, k=10, chop=5, dropout=0.5, learn_rate=None):
    """bench.py's param_dict_for (the demo scripts' values: neg_loss_weight 128, loss_gamma 10, learn_rate 0.01, u_reg 1e-6)."""
    return {"reset_after_getconf": True, "max_epoch": epochs, "loss": loss, "num_negatives": k if scheme == "original" else 0,
            "neg_loss_weight": 8 if loss == "mse" else 128, "loss_gamma": 0.1 if loss == "max-margin" else 10,
            "learn_rate": learn_rate if learn_rate is not None else (0.001 if loss == "mse" else 0.01),
            "neg_dist": "unigram", "neg_sampling_power": 1, "batch_size_p": b, "user_dim": d, "item_dim": d, "word_dim": d,
            "chop_size": chop, "u_reg": 1e-6, "word_emb_dropout_rate": dropout, "interaction_bias": None, "optimizer": optimizer,
            "group_shuffling_trick": True}


def _quiet():
    return contextlib.redirect_stdout(sys.stderr if os.environ.get("NNCF_PARITY_TRACE") == "1" else open(os.devnull, "w"))


def run_case(data, model="basic_embedding", scheme="neg_shared", loss="skip-gram", optimizer="sgd", epochs=1, eval_epochs=None,
             np_seed=4224, sampler_seed=12345, model_seed=0, device=0, learn_rate=None, log=None, control_fp32=False, max_steps=None):
    """Train the product Trainer for `epochs` epochs (evaluating whole@50 after every epoch, as main.py does), then the oracle
    loop from the same initial state; returns a dict of per-step losses (both sides), per-epoch metrics (both sides at
    `eval_epochs`), final tables / parameters (both sides + the initial ones).  max_steps: only the first max_steps * b training
    links are used (the float64 conv oracle of cnn_embedding takes seconds per step)."""
    import torch
    from nncf_amd.conf import get_conf
    from nncf_amd.data_utils import degree_dist, get_data, group_shuffle_train
    from nncf_amd.model_framework import get_model
    from oracle.sampler_oracle import OracleSampler
    say = log or (lambda *a: None)
    eval_epochs = sorted(set(eval_epochs if eval_epochs is not None else range(1, epochs + 1)))
    pd = param_dict(model, scheme, loss, optimizer, epochs, learn_rate=learn_rate)
    conf = get_conf("synthetic_citeulike", "best", pd, model_choice=model)
    conf.eval_topk = 50
    if max_steps:
        data = dict(data, train=data["train"][:max_steps * conf.batch_size_p].copy())
    tower_params = CNN_PARAMS if model == "cnn_embedding" else TOWER_PARAMS
    d_prod = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in data.items()}
    train0 = data["train"].copy()
    b, k = conf.batch_size_p, conf.num_negatives
    n_steps = len(train0) // b
    with _quiet():
        helper = get_data(d_prod, conf, rand_seed=sampler_seed, device=device)
        md = get_model(conf, helper, model, device=device, seed=model_seed)
    head = md[HEAD_OF[scheme]]
    eng = head.engine
    tw = head.core.tower if hasattr(head, "core") else None
    nu, ni = helper.data_spec.user_count, helper.data_spec.item_count
    init = {"U": eng.U.cpu().numpy().copy()}
    if tw is None:
        init["V"] = eng.V.cpu().numpy().copy()
    else:
        for n in tower_params:
            init[n] = getattr(tw, n).cpu().numpy().copy()

    # ---- product ------------------------------------------------------------------------------------------------------
    step_losses = []
    inner = head.run_epoch

    def recording_run_epoch(*a, **kw):
        out = inner(*a, **kw)
        step_losses.append(out.detach().double().cpu().numpy().copy())
        return out
    head.run_epoch = recording_run_epoch
    Trainer = importlib.import_module("nncf_amd.train_" + scheme).Trainer
    t0 = time.time()
    with _quiet():
        np.random.seed(np_seed)
        trainer = Trainer(md, conf, helper)
        trainer.train("whole")
    torch.cuda.synchronize()
    say("product: %d epochs in %.1f s (%.3f s training per epoch)" % (epochs, time.time() - t0, float(np.mean(trainer.train_time))))
    head.run_epoch = inner
    got_losses = np.concatenate([x for x in step_losses if len(x)]) if step_losses else np.zeros(0)
    assert got_losses.size == epochs * n_steps, (got_losses.size, epochs, n_steps)
    got_metrics = {h[0]: (h[3], h[4]) for h in trainer.history}
    got_cost = {h[0]: h[2] for h in trainer.history}
    final = {"U": eng.U.cpu().numpy().copy()}
    if tw is None:
        final["V"] = eng.V.cpu().numpy().copy()
    else:
        for n in tower_params:
            final[n] = getattr(tw, n).cpu().numpy().copy()

    # ---- the reference loop on the oracle (float64; and, as a control, the same loop in float32) ---------------------------
    scales = {}

    def scale_of(step_index):
        if tw is None or conf.word_emb_dropout_rate <= 0:
            return None
        if step_index not in scales:
            scales.clear()
            scales[step_index] = tw.dropout_scale(step_index).cpu().numpy().astype(np.float64)
        return scales[step_index]

    def oracle_loop(dtype, evaluate):
        oc = so.OracleConf(batch_size_p=b, num_negatives=k, loss=conf.loss, loss_gamma=conf.loss_gamma,
                           neg_loss_weight=conf.neg_loss_weight, u_reg=conf.u_reg, learn_rate=conf.learn_rate,
                           emb_normalization=conf.emb_normalization, model_choice="basic_embedding" if tw is not None else model,
                           optimizer=optimizer)
        to = None
        if model == "cnn_embedding":
            to = so.CnnTowerOracle(data["C"], init["W"], init["Wc"], init["bc"], init["g1"], init["b1"], init["Wd"], init["bd"],
                                   init["gamma"], init["beta"], conf.learn_rate, dtype=dtype)
        elif tw is not None:
            to = so.TowerOracle(data["C"], init["W"], init["Wd"], init["bd"], init["gamma"], init["beta"], conf.learn_rate,
                                dtype=dtype).use_bag()
        orc = so.StepOracle(init["U"], init.get("V"), oc, dtype=dtype, tower=to)
        smp = OracleSampler(degree_dist(train0, 1, "unigram"), 1.0, sampler_seed) if scheme == "original" else None
        iidx = np.arange(train0[:, 1].max() + 1)
        train = train0.copy()
        losses, metrics, cost = [], {}, {}
        np.random.seed(np_seed)
        t0 = time.time()
        for epoch in range(epochs + 1):
            train, batches = eo.epoch_batches(train, scheme, b, np.random.shuffle,
                                              lambda t: group_shuffle_train(t, by="item", chop=conf.chop_size, iidx=iidx))
            if epoch == 0:       # train_neg_shared.py:34,43: epoch 0 shuffles, trains nothing, evaluates
                cost[0] = -1
                continue
            base = (epoch - 1) * n_steps

            def sc(t):
                if model == "cnn_embedding":   # the conv Dropout's mask of this train-mode forward (ones without conv dropout)
                    to.conv_mask = tw.conv_mask(base + t, b * (1 + k) if scheme == "original" else b).cpu().numpy().astype(dtype)
                v = scale_of(base + t)
                return v if v is None else v.astype(dtype)
            ls = eo.run_reference_epoch(orc, scheme, batches, step_scale=sc, sampler=smp, k=k, loss_name=conf.loss)
            losses.append(ls)
            it = n_steps + (1 if len(train) % b else 0)          # train_neg_shared.py:44-48: counted before the ragged-tail check
            cost[epoch] = float(ls.sum() / it)
            if evaluate and epoch in eval_epochs:
                metrics[epoch] = eo.evaluate_run(orc, data, 50, nu, ni)
            say("oracle %s: epoch %d, mean cost %.6f (product %.6f), %.1f s so far" % (np.dtype(dtype).name, epoch, cost[epoch],
                                                                                      got_cost[epoch], time.time() - t0))
        params = {"U": orc.U}
        if to is None:
            params["V"] = orc.V
        else:
            for n in tower_params:
                params[n] = getattr(to, n)
        return orc, to, np.concatenate(losses), metrics, cost, params

    orc, to, want_losses, want_metrics, want_cost, want = oracle_loop(np.float64, True)
    control = None
    if control_fp32:   # how far does the SAME loop in float32 drift from float64?  (the yardstick for chaotic configurations)
        _, _, c_losses, c_metrics, _, c_params = oracle_loop(np.float32, True)
        control = dict(losses=c_losses, metrics=c_metrics, params=c_params)
    return dict(conf=conf, md=md, head=head, trainer=trainer, helper=helper, orc=orc, tower_oracle=to, n_steps=n_steps,
                got_losses=got_losses, want_losses=want_losses, got_cost=got_cost, want_cost=want_cost,
                got_metrics=got_metrics, want_metrics=want_metrics, init=init, final=final, want=want, control=control)


def table_report(res, terms_of, updates):
    """worst |product - oracle| / term-count bound per parameter (tests/tolerances.py)."""
    out = {}
    for name, got in res["final"].items():
        want, init = np.asarray(res["want"][name], dtype=np.float64), res["init"][name]
        err = np.abs(got.astype(np.float64) - want)
        if name == "bd":
            # The Dense bias in front of a BatchNormalization has NO gradient in exact arithmetic (the batch mean removes it):
            # what both sides hold is the rounding residue of sum_n dz[n] -- ~1e-16 in the float64 oracle, ~1e-7 in fp32 --
            # and the forward pass is blind to it (z - mean(z)).  No movement-based bound exists; the absolute size is reported.
            out[name] = {"worst_err_over_bound": None, "max_abs_err": float(err.max()), "max_abs_value": float(np.abs(want).max()),
                         "note": "zero true gradient under BatchNorm: rounding residue only"}
            continue
        if name in ("bc",):
            # the conv bias in front of BN1: no gradient in exact arithmetic either (the statistics remove it)
            out[name] = {"worst_err_over_bound": None, "max_abs_err": float(err.max()), "max_abs_value": float(np.abs(want).max()),
                         "note": "zero true gradient under BatchNorm: rounding residue only"}
            continue
        if name in ("rm", "rv", "rm1", "rv1"):       # running statistics: an exponential average of per-step batch statistics, not a gradient sum
            bound = sum_bound(want, init, terms=terms_of.get("Wd", 512), updates=updates)
        else:
            bound = sum_bound(want, init, terms=terms_of[name], updates=updates)
        out[name] = {"worst_err_over_bound": float((err / bound).max()), "max_abs_err": float(err.max()),
                     "max_abs_value": float(np.abs(want).max()),
                     "frac_entries_over_bound": float((err > bound).mean())}
    return out


def loss_report(res):
    g, w = res["got_losses"], res["want_losses"]
    rel = np.abs(g - w) / np.maximum(1.0, np.abs(w))
    over = np.nonzero(rel > 1e-5)[0]
    return {"steps": int(g.size), "max_rel_diff": float(rel.max()), "argmax_step": int(rel.argmax()),
            "first_step_over_1e-5": int(over[0]) if over.size else None, "steps_over_1e-5": int(over.size),
            "rel_diff_quantiles_50_90_99": [float(np.quantile(rel, q)) for q in (0.5, 0.9, 0.99)]}


def metric_report(res):
    out = {}
    for epoch, (w_tr, w_te) in sorted(res["want_metrics"].items()):
        g_tr, g_te = res["got_metrics"][epoch]
        row = {}
        for split, g, w in (("train", g_tr, w_tr), ("test", g_te, w_te)):
            for name in ("recall", "map", "precision"):
                row["%s_%s@50" % (split, name)] = {"product": float(g["%s@50" % name]), "oracle": float(w[name]),
                                                   "diff": float(g["%s@50" % name] - w[name])}
            row["%s_users" % split] = {"product": int(g["users"]), "oracle": int(w["users"])}
        out[int(epoch)] = row
    return out


def control_report(res):
    """float32 oracle vs float64 oracle (the same loop, same masks): the drift that precision alone causes."""
    c = res.get("control")
    if not c:
        return None
    w = res["want_losses"]
    rel = np.abs(c["losses"] - w) / np.maximum(1.0, np.abs(w))
    over = np.nonzero(rel > 1e-5)[0]
    out = {"max_rel_step_loss_diff": float(rel.max()), "first_step_over_1e-5": int(over[0]) if over.size else None,
           "steps_over_1e-5": int(over.size), "tables_max_abs_diff": {}}
    for n, v in c["params"].items():
        out["tables_max_abs_diff"][n] = float(np.abs(np.asarray(v, dtype=np.float64) - np.asarray(res["want"][n], dtype=np.float64)).max())
    out["metrics"] = {}
    for epoch, (w_tr, w_te) in sorted(res["want_metrics"].items()):
        if epoch in c["metrics"]:
            c_tr, c_te = c["metrics"][epoch]
            out["metrics"][int(epoch)] = {"train_recall@50_diff": float(c_tr["recall"] - w_tr["recall"]),
                                          "test_recall@50_diff": float(c_te["recall"] - w_te["recall"]),
                                          "train_map@50_diff": float(c_tr["map"] - w_tr["map"]),
                                          "test_map@50_diff": float(c_te["map"] - w_te["map"])}
    return out
