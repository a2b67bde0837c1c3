#!/usr/bin/env python3
"""Headline benchmark: positive-links/sec of the NNCF training step on MI355X.

    python bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch of b = 512 positive links of the synthetic
CiteULike-a-shaped data set (16K users x 17K items x 200K links, item text L = 200, d = 100) of
BASELINE configs[1]: basic_embedding, train_scheme = neg_shared, as the demo scripts run it
(--conf_choice best: word-row dropout 0.5) --
    duplicate plan + tf.unique -> text tower forward of the batch's unique items (word-row gather +
    mean, Dense; BatchNorm statistics derived by the step kernel, BN + ReLU applied when it loads an
    item row) -> ONE launch: U.V^T shared-negative scores -> loss gradient -> back-GEMMs -> sparse SGD
    on the user rows, item gradients exported -> tower backward (BN / Dense by MFMA, sparse SGD on the
    word table)
with all inputs (link list, content matrix, parameters) resident in HBM before the timed region
and no host synchronisation between steps: the trainer loop (models/train_neg_shared.py:46-59)
runs inside one native call (nncf_tower_epoch / nncf_epoch_*).  --model mf runs the id-table
towers alone.  --gpus N > 1 (the parent starts its own N ranks unless it already runs under torchrun):
data-parallel text tower for the default model, the item-sharded id-table path for --model mf.
Prints ONE JSON line (rank 0).

The line also carries
  roofline     -- the dominant kernel of the step by time: algorithmic bytes / flops per launch
                  (stated in `algorithmic_what`) / its mean duration: the rocprofv3
                  kernel-trace average committed under profiles/ while its source hash matches the
                  current kernels, else HIP events on the launch stream in a second pass over the same
                  steps (`kernel_ms_source` says which; roofline_other_kernels: the other heavy kernels);
  cpu_baseline -- the CPU restatement of the same step (oracle/step_torch.py: torch-CPU autograd at 1 / 8 /
                  32 threads; cnn_embedding: the numpy oracle; kind "port") and the C sampler on one core,
                  timed on this box's host cores on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")   # kernel arguments in device memory (nncf_amd/__init__.py says why); before HIP initialises

PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: fp32-input MFMA = fp32 vector peak
PEAK_F16_MFMA_TFLOPS = 2500.0  # dense f16 / bf16 MFMA (the guide's ~2.5 PFLOP/s)
# the large-batch kernels (fused_wide / pair_wide) form their fp32 products on the f16 matrix pipe: x = hi + lo / 2048, three f16
# MFMAs per 32 k-values with the cross terms in their own fp32 accumulator (csrc/devutil.h split_h; NNCF_WIDE_F16SPLIT=0: fp32 pipe)
WIDE_KERNELS = ("fused_wide_kernel", "pair_wide_kernel")
PEAK_HBM_GBS = 8000.0          # HBM3E spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None, help="GPUs of this node (default: WORLD_SIZE, else 1); > 1 without a "
                    "torchrun environment: bench.py starts the N ranks itself")
    ap.add_argument("--steps", type=int, default=3900)     # 10 epochs of 390 steps
    ap.add_argument("--warmup", type=int, default=390)
    ap.add_argument("--scheme", default="neg_shared", choices=["original", "neg_shared", "group_neg_shared"])
    ap.add_argument("--loss", default=None)
    ap.add_argument("--model", default="basic_embedding", choices=["basic_embedding", "cnn_embedding", "mf"],
                    help="basic_embedding = mean-pool text tower (BASELINE configs[1]); cnn_embedding = CNN text tower "
                         "(configs[4]: use --scheme group_neg_shared); mf = id-table towers only")
    ap.add_argument("--conv-dropout", type=float, default=0.1, help="cnn_embedding: Dropout after the conv ('best' preset)")
    ap.add_argument("--filter-len", type=int, default=3, help="cnn_embedding: Convolution1D filter length ('best' preset)")
    ap.add_argument("--filter-lens", default=None, help="cnn_embedding: a filter BANK, e.g. 1,3,5 with --dim 50 --conv-dropout 0.3 "
                                                        "--dropout 0 = the reference's 'default' preset (cnn_embedding_conf.py:30-50)")
    ap.add_argument("--dim", type=int, default=100)
    ap.add_argument("--batch", type=int, default=512)
    ap.add_argument("--negatives", type=int, default=10)
    ap.add_argument("--chop", type=int, default=5)
    ap.add_argument("--content-len", type=int, default=200)
    ap.add_argument("--vocab", type=int, default=8001)
    ap.add_argument("--dropout", type=float, default=0.5,
                    help="word-row dropout of the text tower in training; 0.5 = the 'best' preset every demo script "
                         "runs with (basic_embedding_conf.py:112, scripts/demos/run_*.sh --conf_choice best)")
    ap.add_argument("--optimizer", default="sgd", choices=["sgd", "adam", "sparse_adam"],
                    help="sgd = the north-star's sparse SGD (headline); adam = Keras Adam, the reference's dense "
                         "default; sparse_adam = utils/optimizer.py")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the cpu_baseline leg (0 = skip)")
    ap.add_argument("--recall-epochs", type=int, default=30,
                    help="epochs of the side run that reports Recall@50 of a TRAINED model (Keras Adam, the reference's "
                         "optimiser; 30 = the demo scripts' max_epoch); 0 = skip")
    ap.add_argument("--profile-launches", type=int, default=1000)
    ap.add_argument("--dump-stamps", default=None, help="developer: with NNCF_DEBUG_TIMING=1, save the step kernel's in-kernel "
                                                        "cycle stamps of the last launch to this .npy (tools/fused_timing.py)")
    ap.add_argument("--roofline-kernel", default=None, help="HIP kernel to bracket (default: per workload)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="--gpus N > 1, --model mf: weak = b links per GPU and step (default); strong = ONE global batch of b "
                         "links per step split by rows over the GPUs (the one-device step's result; SURVEY.md 8(e))")
    ap.add_argument("--small", action="store_true", help="2K x 2K x 20K data (quick functional run)")
    ap.add_argument("--device-links", type=float, default=0,
                    help="--large only: generate this many (user, item) links ON THE DEVICE (Zipf-like over 10M users / 1M items, "
                         "torch's Philox counter PRNG, int32 pairs: 8 GB at 1e9 = SURVEY.md 8(d) input B) and train on their "
                         "first (warmup + steps) * batch; 0 = only the consumed links, generated on the host")
    ap.add_argument("--large", action="store_true",
                    help="BASELINE config-4 shape on one GPU: 10M users x 1M items, d=128, id tables (mf), random "
                         "Zipf links; tables 5.6 GB -- the HBM-bound regime")
    return ap.parse_args()


def param_dict_for(args):
    loss = args.loss or ("log-loss" if args.scheme == "group_neg_shared" else "skip-gram")
    return {"reset_after_getconf": True, "max_epoch": 1, "loss": loss,
            "num_negatives": args.negatives if args.scheme == "original" else 0,
            "neg_loss_weight": 8 if loss == "mse" else 128, "loss_gamma": 0.1 if loss == "max-margin" else 10,
            "learn_rate": 0.001 if loss == "mse" else 0.01, "neg_dist": "unigram", "neg_sampling_power": 1,
            "batch_size_p": args.batch, "user_dim": args.dim, "item_dim": args.dim, "word_dim": args.dim,
            "chop_size": args.chop, "u_reg": 1e-6, "word_emb_dropout_rate": args.dropout, "interaction_bias": None,
            "optimizer": args.optimizer, "num_filters": [args.dim],
            "filter_lengths": [bank_lens(args)] if args.filter_lens else [args.filter_len],
            "conv_dropout_rate": args.conv_dropout}


def bank_lens(args):
    return [int(v) for v in args.filter_lens.split(",")] if args.filter_lens else [args.filter_len]


def oracle_conf_for(args):
    """OracleConf of the run -- ONLY for the cpu_baseline leg (the product path takes nncf_amd.conf.get_conf)."""
    from oracle.step_oracle import OracleConf
    pd = param_dict_for(args)
    return OracleConf(batch_size_p=pd["batch_size_p"], num_negatives=pd["num_negatives"], loss=pd["loss"],
                      loss_gamma=pd["loss_gamma"], neg_loss_weight=pd["neg_loss_weight"], u_reg=pd["u_reg"],
                      learn_rate=pd["learn_rate"], model_choice=args.model, optimizer=args.optimizer)


def product_conf(args):
    from nncf_amd.conf import get_conf
    return get_conf("synthetic_citeulike", "best", param_dict_for(args), model_choice=args.model)


def make_large(args):
    """config-4-shaped id-table problem: only the links the run consumes are materialised."""
    users, items = 10000000, 1000000
    rs = np.random.RandomState(20170813)
    n = (args.warmup + args.steps) * args.batch
    zipf = lambda size, count: np.minimum((count * rs.power(0.35, size)).astype(np.int64), count - 1)  # noqa: E731
    train = np.stack([zipf(n, users), zipf(n, items), np.ones(n, dtype=np.int64)], 1)
    train[0, 0], train[0, 1] = users - 1, items - 1   # table sizes come from the max ids (data_utils.py:23-25)
    return {"C": np.zeros((items, 1), dtype=np.int64), "train": train, "test": train[:1000].copy(),
            "test_seen": train[:1000].copy(), "train_items": np.arange(items), "test_items": np.arange(items)}


def make_data(args):
    from nncf_amd.synthetic import make_citeulike_like
    if args.large:
        return make_large(args)
    text = args.content_len if args.model != "mf" else 0
    if args.small:
        return make_citeulike_like(users=2000, items=2000, train_links=20000, seed=7, content_len=min(text, 40),
                                   vocab=min(args.vocab, 500))
    return make_citeulike_like(content_len=text, vocab=args.vocab)


def build_inputs(args, conf, rank, world):
    """Shuffled (and for group_neg_shared: stratified) link stream long enough for warmup+steps."""
    from nncf_amd.data_utils import group_shuffle_train, degree_dist
    data = make_data(args)
    train = data["train"].copy()
    users = int(train[:, 0].max()) + 1
    items = int(data["C"].shape[0])
    rs = np.random.RandomState(4224 + rank)
    b = conf.batch_size_p
    need = (args.warmup + args.steps) * b
    chunks = []
    have = 0
    iidx = np.arange(int(train[:, 1].max()) + 1)
    while have < need:
        if args.scheme == "group_neg_shared":
            ep = group_shuffle_train(train, by="item", chop=args.chop, iidx=iidx, rng=rs)
        else:
            rs.shuffle(train)
            ep = train
        n = (ep.shape[0] // b) * b          # ragged tail dropped, train_*.py:46-48
        chunks.append(ep[:n].copy())
        have += n
    stream = np.concatenate(chunks, 0)[:need]
    dist = degree_dist(data["train"], column=1, neg_dist="unigram")
    return dict(users=users, items=items, uid=stream[:, 0], cid=stream[:, 1], dist=dist, data=data)


def cpu_model_name():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def cpu_baseline(args, inp, U0, V0, tower_params):
    """the CPU restatement of the same step on this box's host cores, bounded sample.
    basic_embedding / mf: oracle/step_torch.py (torch-CPU graph + autograd, as the reference's Keras-CPU path
    runs it) at all cores and at 1 thread; cnn_embedding: the numpy oracle.  Plus the C restatement of the
    sampler (oracle/nodesampler_oracle.c) on one core -- the reference sampler is single-threaded."""
    from oracle import step_oracle as so
    from oracle.sampler_oracle import OracleSampler
    import torch
    conf = oracle_conf_for(args)
    nproc = os.cpu_count() or 1
    b, k = conf.batch_size_p, conf.num_negatives
    budget = args.cpu_seconds
    max_steps = min(args.steps + args.warmup, len(inp["uid"]) // b)
    drop = args.dropout if tower_params is not None else 0.0
    sampler = OracleSampler(inp["dist"], 1.0, 12345)
    t0 = time.time()
    draws = 0
    while time.time() - t0 < min(2.0, budget / 6):
        sampler.sample_batch(1 << 20)
        draws += 1 << 20
    sampler_rate = draws / (time.time() - t0)

    def batches(stepper, seconds, mask_words):
        mask_rs = np.random.RandomState(29)
        t0 = time.time()
        steps = 0
        while steps < max_steps and (time.time() - t0 < seconds or steps < 1):
            if drop > 0:   # Embedding(dropout): a fresh row mask per batch (mean_pool.py:56-59)
                stepper.row_scale = ((mask_rs.rand(mask_words) < 1 - drop) / (1 - drop)).astype(np.float32)
            u = inp["uid"][steps * b:(steps + 1) * b]
            c = inp["cid"][steps * b:(steps + 1) * b]
            if args.scheme == "original":
                negs = sampler.sample_batch(b * k)
                uu, cc, yy = so.assemble_original(np.stack([u, c, np.ones(b, dtype=np.int64)], 1), negs, conf.loss)
                stepper.step_original(uu, cc, yy)
            elif args.scheme == "neg_shared":
                stepper.step_neg_shared(u, c)
            else:
                stepper.step_group_neg_shared(u, c)
            steps += 1
        return steps, time.time() - t0

    if args.model == "cnn_embedding":
        kc = 1 - args.conv_dropout
        mk_mask = lambda F: ((np.random.RandomState(3).rand(b * (1 + k), inp["data"]["C"].shape[1], F) < kc) / kc).astype(np.float32)  # noqa: E731
        if isinstance(tower_params, dict):   # a filter bank (--filter-lens): parameters at the configured widths
            tower = so.CnnBankTowerOracle(inp["data"]["C"], lr=conf.learn_rate, dtype=np.float32, **tower_params)
            if args.conv_dropout > 0:
                tower.conv_mask = [mk_mask(w.shape[2]) for w in tower.Wc]
        else:
            tower = so.CnnTowerOracle(inp["data"]["C"], *tower_params, lr=conf.learn_rate, dtype=np.float32)
            if args.conv_dropout > 0:   # Dropout(conv_dropout_rate), cnn_model.py:97: one fixed mask (timing only)
                tower.conv_mask = mk_mask(tower.Wc.shape[2])
        o = so.StepOracle(U0, None, conf, dtype=np.float32, tower=tower)
        steps, dt = batches(o, budget, tower.W.shape[0])
        return {"value": steps * b / dt, "unit": "positive-links/s", "cores": int(nproc), "kind": "port",
                "cpu_model": cpu_model_name(), "sampler_samples_per_s_1core": sampler_rate,
                "sample": "%d steps of the same %s/%s workload (numpy fp32 oracle, BLAS threads = all cores, %.1f s)"
                          % (steps, args.model, args.scheme, dt)}
    from oracle.step_torch import TorchStep
    tower = None
    words = 0
    if tower_params is not None:
        tower = dict(zip(("W", "Wd", "bd", "gamma", "beta"), tower_params), C=inp["data"]["C"])
        words = tower["W"].shape[0]
    res = {}
    prev = torch.get_num_threads()
    # a b = 512 step is too small for hundreds of threads (256 threads: 10 s per step on a 2 x 64-core box);
    # 1 thread and a few intra-op thread counts are tried; `value` is the best of them, all are in by_threads
    cand = sorted({t for t in (8, 32, nproc) if t <= nproc and (t <= 64 or t == nproc and nproc <= 64)})
    for threads in [1] + cand:
        st = TorchStep(U0, V0 if tower is None else None, conf, tower=tower, threads=threads)
        res[threads] = batches(st, budget / (2.0 if threads == 1 else 2.0 * len(cand)), words)
    torch.set_num_threads(prev)
    rate = {t: s * b / dt for t, (s, dt) in res.items()}
    best = max(rate, key=lambda t: rate[t])     # the CPU at its best, whichever thread count that is (cores = that count)
    return {"value": rate[best], "unit": "positive-links/s", "cores": int(best), "kind": "port",
            "value_1thread": rate[1], "by_threads": {str(t): rate[t] for t in sorted(rate)}, "host_cores": int(nproc),
            "cpu_model": cpu_model_name(), "sampler_samples_per_s_1core": sampler_rate,
            "sample": "CPU restatement of the reference path (Keras 1.2.2 / TF 1.0 not runnable): torch-CPU graph + autograd "
                      "of the same %s/%s step (oracle/step_torch.py): %s; sampler = oracle/nodesampler_oracle.c on one core"
                      % (args.model, args.scheme, ", ".join("%d steps at %d thread%s in %.1f s" % (res[t][0], t, "s" if t > 1 else "", res[t][1])
                                                            for t in sorted(res)))}


def trained_recall(args, data, device):
    """Second half of BASELINE's metric on a TRAINED state: a fresh model of the same configuration, trained
    through the product Trainer (models/train_*.py mirror) for --recall-epochs epochs with the reference's
    optimiser (Keras Adam(learn_rate), basic_embedding_conf.py:59-60), then whole@50 on the GPU evaluator.
    Outside every timed region."""
    import contextlib
    import importlib
    from nncf_amd.conf import get_conf
    from nncf_amd.data_utils import get_data
    from nncf_amd.model_framework import get_model
    pd = dict(param_dict_for(args), optimizer="adam", max_epoch=int(args.recall_epochs))
    conf = get_conf("synthetic_citeulike", "best", pd, model_choice=args.model)
    conf.eval_topk = 50
    d2 = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in data.items()}
    t0 = time.time()
    with contextlib.redirect_stdout(sys.stderr if os.environ.get("NNCF_BENCH_TRACE") == "1" else open(os.devnull, "w")):
        helper = get_data(d2, conf, rand_seed=12345, device=device)
        md = get_model(conf, helper, args.model, device=device, seed=0)
        Trainer = importlib.import_module("nncf_amd.train_" + args.scheme).Trainer
        np.random.seed(4224)
        tr = Trainer(md, conf, helper)
        tr.train(None)
        tr_res, te_res = tr.evaluater.run(md["model_neg_shared"], verbose=False, eval_scheme="whole")
    return {"train": float(tr_res["recall@50"]), "test_cold_items": float(te_res["recall@50"]),
            "optimizer": "adam (Keras Adam(%g), the reference's optimiser)" % conf.learn_rate,
            "epochs": int(args.recall_epochs), "steps_per_epoch": len(d2["train"]) // conf.batch_size_p,
            "train_scheme": args.scheme, "train_seconds_per_epoch": float(np.mean(tr.train_time)) if tr.train_time else None,
            "wall_seconds": time.time() - t0,
            "note": "whole@50 (objectives.py:296-370,395-421) of a model trained by nncf_amd.train_%s.Trainer on the same "
                    "synthetic data; test items are cold (seen only through their text)" % args.scheme}


def oracle_recall_record(args, got):
    """The float64 oracle's Recall@50 / MAP@50 of the SAME run (same data, seeds, initial parameters, dropout masks, 30 epochs of
    Keras Adam through the reference's trainer loop restated on the oracle): read from the committed record
    profiles/r06_convergence_parity.json (tools/convergence_parity.py; ~9 minutes of host time, so not re-run here).  Attached only
    when the record is of this configuration AND its product-side figures are the ones this run just measured (the kernels are
    deterministic: the same build reproduces them bit for bit) -- otherwise a note says why not."""
    path = os.path.join(ROOT, "profiles", "r06_convergence_parity.json")
    try:
        rec = json.load(open(path))
    except Exception:
        return {"available": False, "why": "profiles/r06_convergence_parity.json not found"}
    want = "%s d=%d, %s" % (args.model, args.dim, args.scheme)
    if want not in rec.get("workload", "") or "optimizer adam" not in rec.get("workload", "") or rec.get("epochs") != got["epochs"]:
        return {"available": False, "why": "the committed record is of another configuration (%s, %s epochs)" % (rec.get("workload"), rec.get("epochs"))}
    last = rec["metrics_at_epochs"].get(str(got["epochs"]))
    if not last:
        return {"available": False, "why": "the record holds no evaluation at epoch %d" % got["epochs"]}
    same = (abs(last["train_recall@50"]["product"] - got["train"]) <= 1e-9 and abs(last["test_recall@50"]["product"] - got["test_cold_items"]) <= 1e-9)
    out = {"available": True, "source": "profiles/r06_convergence_parity.json (tools/convergence_parity.py)",
           "train": last["train_recall@50"]["oracle"], "test_cold_items": last["test_recall@50"]["oracle"],
           "train_map@50": last["train_map@50"]["oracle"], "test_map@50": last["test_map@50"]["oracle"],
           "product_figures_of_the_record_equal_this_run": bool(same),
           "note": "30 epochs of dense Adam are a chaotic trajectory in fp32 (tests/test_epoch_parity_gpu.py pins one epoch and the "
                   "float32-oracle yardstick): the curves agree to ~2e-3, not to 1e-5; one epoch of sparse SGD agrees to 1e-8"}
    if not same:
        out["record_product"] = {"train": last["train_recall@50"]["product"], "test_cold_items": last["test_recall@50"]["product"]}
    return out


_T0 = time.time()


def timed_windows(window, reduce_max=None, min_total=0.2, min_windows=5, max_windows=4096):
    """SURVEY.md 8(d) "several timed windows, report the median": `window()` runs EXACTLY K steps between a
    synchronise (+ barrier for N > 1) on both sides and returns its wall seconds (already the MAX over ranks when
    `reduce_max` is given, so every rank takes the same decisions); repeated until at least `min_windows` windows and
    `min_total` seconds are timed.  Returns the list of window times."""
    times = []
    while True:
        dt = window()
        if reduce_max is not None:
            dt = reduce_max(dt)
        times.append(dt)
        if (len(times) >= min_windows and sum(times) >= min_total) or len(times) >= max_windows:
            return times


def window_stats(times, K):
    ts = sorted(times)
    med = ts[len(ts) // 2] if len(ts) % 2 else 0.5 * (ts[len(ts) // 2 - 1] + ts[len(ts) // 2])
    return med, {"count": len(ts), "steps_each": K, "timed_seconds_total": float(sum(ts)),
                 "ms_per_step_min": ts[0] / K * 1e3, "ms_per_step_median": med / K * 1e3, "ms_per_step_max": ts[-1] / K * 1e3,
                 "ms_per_step_first": times[0] / K * 1e3,
                 "note": "ms_per_step / value are the MEDIAN window; every window is exactly `steps` steps between "
                         "synchronisations (later windows go on training over the same links)"}


def trace(msg):
    """stage timestamps on stderr (NNCF_BENCH_TRACE=1)"""
    if os.environ.get("NNCF_BENCH_TRACE") == "1":
        sys.stderr.write("[bench %7.1fs] %s\n" % (time.time() - _T0, msg))
        sys.stderr.flush()


def emit(line):
    """Print the ONE JSON line last: flush C stdio first (RCCL prints its banner through printf,
    which would otherwise land after Python's output at exit)."""
    import ctypes
    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    sys.stdout.flush()
    print(json.dumps(line), flush=True)


def price_on_f16_pipe(r, split_on):
    """Large-batch kernels (WIDE_KERNELS): the products run on the f16 matrix pipe (three f16 MFMAs per fp32-equivalent product):
    `peak` is THAT pipe's dense peak and `frac` the algorithmic (fp32-equivalent) flop against it.  Rounds 2-4 priced the same
    algorithmic flop against the fp32-MFMA peak: kept as frac_of_fp32_mfma_peak (round-comparable; above 0.75 = 6/8 of it the fp32
    pipe could not have done it at all, and since round 5 it reaches 1); what the pipe is actually asked for is named too."""
    r["pipe"] = ("f16 MFMA: 3 v_mfma_f32_16x16x32_f16 per fp32-equivalent 16x16x32 product (hi.hi + (hi.lo + lo.hi) / 2048), fp32 "
                 "accumulate" if split_on else "f32 MFMA (NNCF_WIDE_F16SPLIT=0)")
    if not split_on:
        return
    r["frac_of_fp32_mfma_peak"] = r.get("frac")
    r["peak"] = PEAK_F16_MFMA_TFLOPS
    if r.get("achieved") is not None:
        r["frac"] = r["achieved"] / PEAK_F16_MFMA_TFLOPS
    for fkey in ("frac_uncorrected_bracket", "frac_empty_bracket_subtracted", "frac_profiles"):
        if r.get(fkey) is not None:
            r[fkey] = r[fkey] * PEAK_F32_MFMA_TFLOPS / PEAK_F16_MFMA_TFLOPS
    issued = 4.0 * r["algorithmic_per_launch"]          # 8/6 (S recomputed) x 3 piece products
    r["issued_f16_tflops"] = issued / (r["kernel_ms"] * 1e-3) / 1e12
    r["frac_issued_of_f16_mfma_peak"] = r["issued_f16_tflops"] / PEAK_F16_MFMA_TFLOPS


def roof_entry(name, bound, work, what, raw_ms, ov_ms, launches, rp_ms=None):
    """roofline object of one kernel from THIS run's HIP-event bracket: `kernel_ms` = mean bracket of its launches on
    the launch stream minus the mean EMPTY bracket (two event records back to back, measured in the same pass: the
    marker packets' own cost); `achieved` = algorithmic work per launch / kernel_ms.  The committed rocprofv3
    kernel-trace average of the same workload, when profiles/traffic.json holds one for the current kernel sources,
    is reported beside it (rocprof_kernel_ms, frac_profiles) -- never used for `frac`."""
    kern_ms = max(raw_ms - ov_ms, 1e-6)
    peak, unit, scale = (PEAK_HBM_GBS, "GB/s", 1e9) if bound == "hbm" else (PEAK_F32_MFMA_TFLOPS, "TFLOP/s", 1e12)
    ach = work / (kern_ms * 1e-3) / scale
    return {"kernel": name, "bound": bound, "achieved": ach, "peak": peak, "unit": unit, "frac": ach / peak, "traffic": None,
            "algorithmic_per_launch": work, "algorithmic_what": what, "kernel_ms": kern_ms,
            "kernel_ms_source": "HIP event bracket of this run minus the empty bracket (live)",
            "event_bracket_ms": raw_ms, "empty_bracket_ms": ov_ms, "launches": launches,
            "frac_uncorrected_bracket": work / (raw_ms * 1e-3) / scale / peak,
            "rocprof_kernel_ms": rp_ms, "frac_profiles": (work / (rp_ms * 1e-3) / scale / peak) if rp_ms else None}


# every kernel name a training step can launch (ProfScope names of nncf_amd/csrc/*.hip; the once-per-64-steps plan
# launches, the sampler, the evaluator and the sharded row movers are not per-step work of the N = 1 paths)
STEP_KERNELS = ("tower_pool_kernel", "tower_stats_kernel", "tower_bn_kernel", "tower_sums_kernel", "tower_merge_kernel",
                "tower_bwd_kernel", "tower_word_kernel", "fused_strip_kernel", "fused_wide_kernel", "fused_commit_kernel",
                "shared_tile_kernel", "shared_apply_kernel", "original_score_kernel", "original_apply_kernel",
                "rowwise_score_kernel", "rowwise_apply_kernel", "pair_anchor_kernel", "pair_wide_kernel", "pair_commit_kernel",
                "bias_update_kernel", "adam_idle_kernel", "cnn_conv_fwd_kernel", "cnn_stats_kernel", "cnn_pool_dense_kernel",
                "cnn_merge_kernel", "cnn_conv_bwd_kernel", "cnn_update_kernel",
                # round 5: the f16-plane guard of the large-batch step (range of the batch's rows; the fp32-pipe twin that returns at once)
                "wide_domain_kernel", "fused_wide_twin_kernel")


def calibrate_brackets(roofs, step_ms, prof_n):
    """VERDICT r3 item 5c.  A HIP-event bracket holds the kernel plus what the two marker packets cost in a CHAIN of
    launches, which is less than the back-to-back empty bracket (the next launch's dispatch overlaps the previous
    kernel): subtracting the whole empty bracket flattered every kernel by 6-20 %.  With every kernel of a step bracketed
    (one pass each) the per-bracket cost c follows from the step itself: sum_k launches_k (bracket_k - c) = ms_per_step of
    the un-bracketed timed windows, 0 <= c <= empty bracket.  kernel_ms / achieved / frac are then the calibrated figures
    (their sum IS the step); when the brackets do not add up to a step within those bounds (a kernel outside
    STEP_KERNELS, launches on another stream) nothing is changed and `bracket_calibration` says so."""
    live = [r for r in roofs if r.get("launches")]
    if not live or prof_n <= 0:
        return
    per_step = [r["launches"] / float(prof_n) for r in live]                     # launches of the kernel per step
    total_raw = sum(n * r["event_bracket_ms"] for n, r in zip(per_step, live))
    n_launch = sum(per_step)
    c = (total_raw - step_ms) / n_launch
    ov = max(r["empty_bracket_ms"] for r in live)
    ok = -0.02 * step_ms / n_launch <= c <= ov * 1.05
    for n, r in zip(per_step, live):
        r["launches_per_step"] = n
        if not ok:
            r["bracket_calibration"] = ("not applied: the brackets of the %d bracketed kernels sum to %.4f ms per step, the step takes "
                                        "%.4f ms (per-bracket cost %.5f ms outside [0, empty bracket])" % (len(live), total_raw, step_ms, c))
            continue
        cc = min(max(c, 0.0), r["event_bracket_ms"] * 0.5)
        scale = 1e9 if r["bound"] == "hbm" else 1e12
        r["frac_empty_bracket_subtracted"] = r["frac"]
        r["kernel_ms"] = r["event_bracket_ms"] - cc
        r["achieved"] = r["algorithmic_per_launch"] / (r["kernel_ms"] * 1e-3) / scale
        r["frac"] = r["achieved"] / r["peak"]
        r["bracket_cost_ms"] = cc
        r["kernel_ms_source"] = ("HIP event bracket of this run minus the per-bracket cost that makes the brackets of ALL the step's "
                                 "kernels sum to ms_per_step (live, calibrated)")


def _profile_entry(args, conf, d, kernel):
    """profiles/traffic.json (tools/make_traffic.py, from the committed rocprofv3 passes of the newest round):
    entry of exactly this workload and kernel, or None."""
    try:
        table = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    except Exception:
        return None
    if args.optimizer != "sgd" or args.large:
        return None   # the committed passes are of the sparse-SGD CiteULike-shaped runs
    key = "%s/%s/%s/b%d/d%d%s" % (args.model, args.scheme, conf.loss, conf.batch_size_p, d, "/small" if args.small else "")
    ent = table.get(key) or {}
    import __graft_entry__ as ge
    if ent.get("csrc_hash") != ge._source_hash():
        return None   # profiled before the kernel sources changed: stale, fall back to the live bracket
    if ent.get("kernel", kernel) == kernel:
        return ent
    return (ent.get("kernels") or {}).get(kernel)


def evaluator_traffic(d):
    """HBM bytes per whole@50 pass of the one-kernel evaluator from the committed PMC passes over tools/eval_bench.py
    (profiles/traffic.json, same corrections and the same source-hash rule as measured_traffic), or None."""
    try:
        ent = json.load(open(os.path.join(ROOT, "profiles", "traffic.json"))).get("evaluator/whole@50/16Kx17K/d%d" % d) or {}
    except Exception:
        return None
    import __graft_entry__ as ge
    return ent.get("bytes_per_pass") if ent.get("csrc_hash") == ge._source_hash() else None


def measured_traffic(args, conf, d, kernel):
    """HBM bytes per launch of `kernel` from the committed PMC passes (FETCH_SIZE + WRITE_SIZE with the
    gfx950 corrections of MI355X_MICROARCH.md), or None."""
    ent = _profile_entry(args, conf, d, kernel)
    return ent.get("bytes_per_launch") if ent else None


def rocprof_duration_ms(args, conf, d, kernel):
    """Average duration (ms) of `kernel` in the committed rocprofv3 --kernel-trace --stats summary, or None."""
    ent = _profile_entry(args, conf, d, kernel)
    us = ent.get("rocprof_avg_us") if ent else None
    return us * 1e-3 if us else None


def rank_windows(run_window, K, device):
    """N > 1: the timed windows of every rank -- barrier + synchronise on both sides of EXACTLY K steps, the window's
    time = MAX over ranks (so all ranks take the same repeat decisions); returns (median seconds, stats)."""
    import torch
    import torch.distributed as dist

    def window():
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run_window()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        dist.barrier()
        return t1 - t0

    def reduce_max(dt):
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    times = timed_windows(window, reduce_max)
    return window_stats(times, K)


def main_data_parallel(args, rank, world, local_rank):
    """N > 1, model_choice = basic_embedding / cnn_embedding: one process per GPU, users + their links
    partitioned by user range, text tower replicated, ONE RCCL all-reduce of the tower gradients per
    global step (nncf_amd/sharded.py: DataParallelTower).  Weak scaling: b positives per rank per step."""
    import torch
    import torch.distributed as dist
    from nncf_amd import engine as nengine
    from nncf_amd.data_utils import get_data
    from nncf_amd.model_framework import get_model
    from nncf_amd.sharded import DataParallelTower, HipTowerReplica, partition_links_by_user
    trace("data-parallel: making data")
    data = make_data(args)
    trace("data made")
    users = int(data["train"][:, 0].max()) + 1
    pconf = product_conf(args)
    b, d = pconf.batch_size_p, args.dim
    mine = partition_links_by_user(data["train"], users, world)[rank].copy()
    rs = np.random.RandomState(4224 + rank)
    W, K = args.warmup, args.steps
    need = (W + K) * b
    chunks, have = [], 0
    while have < need:
        rs.shuffle(mine)
        chunks.append(mine.copy())
        have += len(mine)
    stream = np.concatenate(chunks, 0)[:need]
    helper = get_data(data, pconf, rand_seed=12345, device=local_rank)
    md = get_model(pconf, helper, args.model, device=local_rank, seed=0)   # same seed: identical replicas
    trace("model built")
    rep = HipTowerReplica(md, args.scheme)
    if pconf.word_emb_dropout_rate > 0 or args.model == "cnn_embedding":
        rep.tower.set_dropout(pconf.word_emb_dropout_rate, 29 + rank)            # each rank its own masks
    tr = DataParallelTower(rep)
    eng = rep.engine
    uid, cid = eng.ids(stream[:, 0]), eng.ids(stream[:, 1])
    losses = torch.zeros(K, dtype=torch.float32, device=eng.device)
    want_cpu = rank == 0 and args.cpu_seconds > 0
    U0 = eng.U.cpu().numpy().copy() if want_cpu else None
    tower_params = None
    if want_cpu:
        tw = rep.tower
        names_p = ("W", "Wc", "bc", "g1", "b1", "Wd", "bd", "gamma", "beta") if args.model == "cnn_embedding" else \
            ("W", "Wd", "bd", "gamma", "beta")
        if getattr(tw, "nb", 0) > 1:
            want_cpu = False      # filter banks: the N = 1 line carries their CPU figure
        else:
            tower_params = [getattr(tw, n).cpu().numpy().copy() for n in names_p]

    def run(first, count, keep):
        tr.run_epoch(uid[first * b:(first + count) * b], cid[first * b:(first + count) * b], count, b, losses if keep else None)

    trace("warmup")
    run(0, W, False)
    trace("timed windows")
    tr.wire_bytes = 0
    elapsed, win_stats = rank_windows(lambda: run(W, K, True), K, eng.device)
    trace("timed windows done")
    ml = losses.double().mean()
    dist.all_reduce(ml)
    mean_loss = float(ml.item()) / world
    if not np.isfinite(mean_loss):
        raise SystemExit("ERROR (nan)! cost is %r" % mean_loss)
    # roofline leg (all ranks step together; rank 0's brackets are reported): the SAME kernel as the N = 1 line
    L = args.content_len if not args.small else min(args.content_len, 40)
    cnn = args.model == "cnn_embedding"
    pointwise = pconf.loss in ("skip-gram", "mse")
    name = args.roofline_kernel or ("cnn_conv_bwd_kernel" if cnn else ("tower_word_kernel" if args.scheme == "original" else
                                    ("fused_strip_kernel" if pointwise else "shared_tile_kernel")))
    prof_n = min(K, args.profile_launches)
    nengine.profile_begin(name, prof_n)
    run(W, prof_n, False)
    raw_ms, launches, ov_ms = nengine.profile_end()
    n_u = float(np.mean([len(np.unique(stream[t * b:(t + 1) * b, 1])) for t in range(W, W + prof_n)]))
    keep = 1.0 - args.dropout
    if name.startswith("cnn_conv"):   # n_u = unique items of the bracketed steps (counted here from the id stream)
        mult = 2 if name == "cnn_conv_fwd_kernel" else 4
        bound, work = "mfma", mult * n_u * L * args.filter_len * d * d
        what = "%d*n_u*L*fl*word_dim*F flop, n_u = %.1f unique items per launch" % (mult, n_u)
    elif name in ("fused_strip_kernel", "shared_tile_kernel"):
        bound, work = "mfma", 6.0 * b * (n_u if args.scheme == "group_neg_shared" else b) * d
        what = "6*b*N*d flop (S, G.V, G^T.U); the one-launch kernel issues 8*b*N*d (S recomputed by the item pass)"
    else:
        bound, work = "hbm", 4.0 * n_u * L * d * keep
        what = "4*L*word_dim*(1-dropout) B of gradient rows per unique item, n_u = %.1f" % n_u
    roof = roof_entry(name, bound, work, what, raw_ms, ov_ms, launches) if launches else None
    cpu = None
    if want_cpu:
        inp = dict(uid=stream[:, 0], cid=stream[:, 1], data=data,
                   dist=__import__("nncf_amd.data_utils", fromlist=["degree_dist"]).degree_dist(data["train"], column=1, neg_dist="unigram"))
        cpu = cpu_baseline(args, inp, U0, None, tower_params)
    dist.barrier()      # the other ranks wait for rank 0's CPU leg
    tower_txt = ("cnn_embedding (CNN text tower L=%d, vocab=%d, word-row dropout %g, Conv1D(%d filters, length %d)+BN+ReLU+"
                 "Dropout(%g)+avg-pool, Dense+BN+ReLU)" % (L, args.vocab, args.dropout, d, args.filter_len, args.conv_dropout)
                 if cnn else "basic_embedding (mean-pool text tower L=%d, vocab=%d, word-row dropout %g, Dense+BN+ReLU)"
                 % (L, args.vocab, args.dropout))
    if rank == 0:
        line = {
            "metric": "positive-links/sec (train step)", "value": K * b * world / elapsed, "unit": "positive-links/s",
            "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": elapsed / K * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "windows": win_stats,
            "data": "synthetic CiteULike-a-shaped (16K users x 17K items x 200K links, item text), random-init parameters",
            "config": {"workload": "CiteULike-a-shaped, %s d=%d, train_scheme=%s, loss=%s, b=%d per GPU, sparse SGD"
                                   % (tower_txt, d, args.scheme, pconf.loss, b),
                       "global_batch": b * world,
                       "parallelism": "1 process per GPU; users + their links partitioned by user range, text tower "
                                      "replicated; one RCCL all-reduce of the tower gradients (%.2f MB) per step; transport = %s"
                                      % (rep.grad.numel() * 4 / 1e6, tr.transport),
                       "wire_bytes_per_rank_per_step": tr.wire_bytes / float(max(1, K * win_stats["count"]))},
            "mean_loss": mean_loss,
            "roofline": roof,
            "cpu_baseline": cpu,
        }
        emit(line)
    dist.destroy_process_group()


def main_sharded(args, conf, rank, world, local_rank):
    """N > 1, --model mf: one process per GPU, user/item tables range-sharded, RCCL all-to-alls of ids (once per 64
    steps), rows and gradients (per step) with compact variable splits (nncf_amd/sharded.py; the north-star's config-4
    path).  --scaling weak (default): b positives per rank per step (ShardedNegShared); --scaling strong: ONE global
    batch of b positives per step split by rows over the ranks (ShardedGlobalBatch: the one-device step's result)."""
    import torch
    import torch.distributed as dist
    from nncf_amd import engine as nengine
    from nncf_amd.sharded import HipBackend, ShardedGlobalBatch, ShardedNegShared, partition_links_by_user, shard_size
    if args.model in ("basic_embedding", "cnn_embedding"):
        if args.scaling == "strong":
            raise SystemExit("--scaling strong is built for --model mf (id tables); the text-tower models run data parallel")
        return main_data_parallel(args, rank, world, local_rank)
    if args.model != "mf":
        raise SystemExit("--gpus > 1 is built for the text-tower models (data parallel) and --model mf (item-sharded)")
    if args.scheme != "neg_shared":
        raise SystemExit("--gpus > 1 is built for train_scheme=neg_shared (north-star config 4 path)")
    strong = args.scaling == "strong"
    data = make_data(args)   # --large: BASELINE configs[3], 10M users x 1M items (only the links the run consumes)
    users, items = int(data["train"][:, 0].max()) + 1, int(data["C"].shape[0])
    b, d = conf.batch_size_p, args.dim
    W, K = args.warmup, args.steps
    need = (W + K) * b
    if strong:
        mine = data["train"].copy()                 # the global link stream, the same on every rank
        rs = np.random.RandomState(4224)
    else:
        mine = partition_links_by_user(data["train"], users, world)[rank].copy()
        rs = np.random.RandomState(4224 + rank)
    chunks, have = [], 0
    while have < need:
        rs.shuffle(mine)
        chunks.append(mine.copy())
        have += len(mine)
    stream = np.concatenate(chunks, 0)[:need]
    norm = bool(conf.emb_normalization)
    kw = dict(user_count=users, item_count=items, dim=d, batch_size_p=b, num_negatives=0, loss=conf.loss,
              loss_gamma=conf.loss_gamma, neg_loss_weight=conf.neg_loss_weight, u_reg=conf.u_reg,
              learn_rate=abs(conf.learn_rate), normalize_user=norm, normalize_item=False)   # mf: model_framework.py:85-88
    if args.optimizer != "sgd":
        kw["optimizer"] = args.optimizer      # the owners step their shards with Adam (nncf_shard_set_optimizer)
    us, isz = shard_size(users, world), shard_size(items, world)
    be = HipBackend(kw, max(1, min(us, users - rank * us)), max(1, min(isz, items - rank * isz)), local_rank,
                    world=world, item_shard=isz, item_offset=rank * isz, seed=rank, rank=rank,
                    mode="strong" if strong else "weak", user_shard=us, user_offset=rank * us)
    if strong:
        tr = ShardedGlobalBatch(be, users, items, b, d)
        uid, cid = be.ids(stream[:, 0]), be.ids(stream[:, 1])
    else:
        tr = ShardedNegShared(be, users, items, b, d)
        uid, cid = be.ids(stream[:, 0] - rank * us), be.ids(stream[:, 1])   # user ids relative to my range
    losses = torch.zeros(K, dtype=torch.float32, device=be.device)
    scratch = torch.zeros(max(W, K, 1), dtype=torch.float32, device=be.device)

    def run(first, count, keep):
        # plans + the id all-to-all batched 64 steps at a time; per step: rows -> local step -> gradients -> apply
        tr.run_epoch(uid[first * b:(first + count) * b], cid[first * b:(first + count) * b], count, losses if keep else scratch)

    run(0, W, False)
    tr.wire_bytes = 0
    elapsed, win_stats = rank_windows(lambda: run(W, K, True), K, be.device)
    wire_per_step = tr.wire_bytes / float(K * win_stats["count"])
    be.engine.check()
    ml = losses.double().mean()
    dist.all_reduce(ml)
    mean_loss = float(ml.item()) / (1 if strong else world)     # strong: a step's loss is the SUM of the ranks' shares
    if not np.isfinite(mean_loss):
        raise SystemExit("ERROR (nan)! cost is %r" % mean_loss)
    # roofline leg: the same kernel as the N = 1 line of this model (the one-launch step), rank 0's brackets
    replicated = bool(getattr(be, "replicated", False))   # pairwise loss on the strong form: every rank runs the whole b x b step
    name = args.roofline_kernel or ("shared_tile_kernel" if replicated else
                                    ("fused_strip_kernel" if (b < 4096 or strong) else "fused_wide_kernel"))
    prof_n = min(K, args.profile_launches)
    nengine.profile_begin(name, prof_n)
    run(W, prof_n, False)
    raw_ms, launches, ov_ms = nengine.profile_end()
    rows = b // world if (strong and not replicated) else b
    work = 6.0 * rows * b * d
    roof = roof_entry(name, "mfma", work, "6*rows*b*d flop (S, G.V, G^T.U) of this rank's %d x %d block of the score matrix; the kernel "
                      "issues 8*rows*b*d (S recomputed by the item pass)" % (rows, b), raw_ms, ov_ms, launches) if launches else None
    if roof and name in WIDE_KERNELS and roof.get("kernel_ms"):
        price_on_f16_pipe(roof, os.environ.get("NNCF_WIDE_F16SPLIT", "1") != "0")
    cpu = None
    if rank == 0 and args.cpu_seconds > 0 and not args.large:
        rs2 = np.random.RandomState(5)     # timing only: tables of the global shape, rank 0's batches
        inp = dict(uid=stream[:, 0], cid=stream[:, 1], data=data,
                   dist=__import__("nncf_amd.data_utils", fromlist=["degree_dist"]).degree_dist(data["train"], column=1, neg_dist="unigram"))
        cpu = cpu_baseline(args, inp, rs2.uniform(-.05, .05, (users, d)).astype(np.float32),
                           rs2.uniform(-.05, .05, (items, d)).astype(np.float32), None)
    dist.barrier()
    if rank == 0:
        gb = b if strong else b * world
        line = {
            "metric": "positive-links/sec (train step)", "value": K * gb / elapsed, "unit": "positive-links/s",
            "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": elapsed / K * 1e3, "higher_is_better": True,
            "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": "f32", "windows": win_stats,
            "data": ("synthetic config-4-shaped (10M users x 1M items, Zipf links), random-init tables" if args.large else
                     "synthetic CiteULike-a-shaped (16K users x 17K items x 200K links), random-init tables"),
            "config": {"workload": "%s, id-table towers (model_choice=mf), tables sharded by id "
                                   "range, d=%d, train_scheme=neg_shared, loss=%s, %s, %s"
                                   % ("config-4-shaped 10M x 1M" if args.large else "CiteULike-a-shaped", d, conf.loss,
                                      ("ONE global batch b=%d per step, rows split over the GPUs" % b) if strong else ("b=%d per GPU" % b),
                                      {"sgd": "sparse SGD", "adam": "Keras Adam (dense)", "sparse_adam": "lazy-sparse Adam"}[args.optimizer]),
                       "global_batch": gb,
                       "parallelism": ("1 process per GPU; user AND item rows range-sharded and fetched per step (users of the local "
                                       "rows, items of all columns), both gradients returned to the owners" if strong else
                                       "1 process per GPU; users + their links and item rows range-sharded") +
                                      "; RCCL all-to-alls with compact variable splits: ids + counts once per 64 steps, rows and "
                                      "gradients per step; transport = %s%s" % (
                                          tr.transport + (" (the library issues every exchange itself, one native call per 64-step chunk: "
                                                          "grouped ncclSend / ncclRecv on the step's stream)" if tr.transport == "native" else ""),
                                          "; replicated pairwise form: every rank runs the whole step, no gradients travel" if replicated else ""),
                       "wire_bytes_per_rank_per_step": wire_per_step},
            "mean_loss": mean_loss,
            "roofline": roof, "cpu_baseline": cpu,
        }
        emit(line)
    dist.destroy_process_group()


def launch_ranks(args):
    """`python bench.py --gpus N` outside torchrun: this process never touches a GPU (no torch device
    call, the build runs in a child) -- it starts N fresh rank processes with
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py <same arguments>`,
    lets their stderr through, relays rank 0's ONE JSON line and exits non-zero if any rank failed
    or no line came back.  Never os.exec*: the ranks are children."""
    import subprocess
    n = int(args.gpus)
    if os.environ.get("NNCF_BENCH_DRYRUN") != "1":
        rc = subprocess.call([sys.executable, "-c", "import __graft_entry__ as g; g.build(quiet=True)"], cwd=ROOT)
        if rc:
            raise SystemExit("bench.py: building the HIP extension failed (exit %d)" % rc)
    # --standalone: torchrun's own rendezvous binds port 0 itself (no pick-then-reuse race between concurrent launches)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1",
           "--nproc-per-node=%d" % n, os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL needs it on this host driver
    env.setdefault("OMP_NUM_THREADS", "1")
    trace("launching %d ranks: %s" % (n, " ".join(cmd)))
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, cwd=os.getcwd(), universal_newlines=True)
    line = None
    for ln in p.stdout:
        t = ln.strip()
        if t.startswith("{") and '"metric"' in t:
            try:
                line = json.loads(t)
            except ValueError:
                sys.stderr.write(ln)
        else:
            sys.stderr.write(ln)   # RCCL banners and the like: not the result line
    rc = p.wait()
    if rc != 0:
        raise SystemExit("bench.py: a rank failed (torch.distributed.run exit code %d)" % rc)
    if line is None:
        raise SystemExit("bench.py: the ranks finished but rank 0 printed no result line")
    if line.get("n_gpus") != n:
        raise SystemExit("bench.py: asked for %d GPUs, the ranks report n_gpus = %r" % (n, line.get("n_gpus")))
    emit(line)


class _NullBackend(object):
    """NNCF_BENCH_DRYRUN=1 (no GPU: launcher / rendezvous / timing-contract check on a CPU box): the
    backend interface of nncf_amd/sharded.py with no arithmetic at all -- CPU tensors of the real
    shapes, so ShardedNegShared.step runs its three all-to-alls over gloo."""

    def __init__(self, b, d, world):
        import torch
        self.b, self.cap, self.world = b, b, world
        self.req = torch.full((world * b,), -1, dtype=torch.int32)
        self.slot = torch.zeros(b, dtype=torch.int32)
        self.rows_in = torch.zeros(world * b, d)
        self.rows_out = torch.zeros(world * b, d)
        self.g_out = torch.zeros(world * b, d)
        self._loss = torch.zeros(1)

    def route(self, cid):
        return self.req, self.slot

    def gather_rows(self, req_in):
        return self.rows_out

    def rows_landing(self):
        return self.rows_in

    def local_step(self, uid_local, slot):
        return self._loss, self.g_out

    def scatter_apply(self, req_in, g_in):
        pass


def main_dryrun(args, rank, world):
    """The N-rank contract without a GPU: gloo process group, W untimed + K timed global steps of the
    exchange pattern alone (no kernels -> "value" is null), barrier on both sides, MAX over ranks,
    rank 0 prints the line."""
    import torch
    import torch.distributed as dist
    from nncf_amd.sharded import ShardedNegShared
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29517")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    if os.environ.get("NNCF_BENCH_DRYRUN_FAIL") == str(rank):   # launcher test: a rank that dies
        raise SystemExit(3)
    b, d = args.batch, args.dim
    tr = ShardedNegShared(_NullBackend(b, d, world), 16000, 17000, b, d)
    ids = torch.zeros(b, dtype=torch.int32)
    W, K = args.warmup, args.steps
    for _ in range(W):
        tr.step(ids, ids)
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(K):
        tr.step(ids, ids)
    t1 = time.perf_counter()
    dist.barrier()
    t = torch.tensor([t1 - t0], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if rank == 0:
        emit({"metric": "positive-links/sec (train step)", "value": None, "unit": "positive-links/s", "n_gpus": world,
              "steps": K, "warmup": W, "ms_per_step": float(t.item()) / max(K, 1) * 1e3, "higher_is_better": True,
              "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "none", "dry_run": True,
              "config": {"workload": "NNCF_BENCH_DRYRUN: exchange pattern of the item-sharded step over gloo, no kernels",
                         "global_batch": b * world, "parallelism": "1 process per rank (CPU, gloo)"},
              "roofline": None, "cpu_baseline": None})
    dist.destroy_process_group()


def main():
    args = parse()
    if args.large:
        args.model, args.dim = "mf", 128
    in_torchrun = "WORLD_SIZE" in os.environ
    if args.gpus is None:
        args.gpus = int(os.environ.get("WORLD_SIZE", "1"))
    if not in_torchrun and args.gpus > 1:
        return launch_ranks(args)        # parent: starts the ranks, relays rank 0's line
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE = %d" % (args.gpus, world))
    if os.environ.get("NNCF_BENCH_DRYRUN") == "1":
        return main_dryrun(args, rank, world)
    # build (make: a child process) BEFORE anything initialises the GPU in this process
    import __graft_entry__ as ge
    if local_rank == 0:
        ge.build(quiet=True)
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU fallback of the product path)")
    torch.cuda.set_device(local_rank)
    # NNCF_FORCE_SHARDED=1 runs the sharded (RCCL all-to-all) path even at world_size 1 (smoke run)
    dist_on = world > 1 or os.environ.get("NNCF_FORCE_SHARDED") == "1"
    if dist_on:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        os.environ.setdefault("NCCL_DEBUG", "WARN")
        trace("init_process_group")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    if dist_on:
        dist.barrier()
        trace("process group up")
    from nncf_amd.sampler import MultinomialSampler

    pconf = product_conf(args)
    conf = pconf
    if dist_on:
        return main_sharded(args, pconf, rank, world, local_rank)
    inp = build_inputs(args, pconf, rank, world)
    b, k, d = pconf.batch_size_p, pconf.num_negatives, args.dim
    from nncf_amd.data_utils import get_data
    from nncf_amd.model_framework import get_model
    from nncf_amd import engine as nengine
    helper = get_data(inp["data"], pconf, rand_seed=12345, device=local_rank)
    md = get_model(pconf, helper, args.model, device=local_rank, seed=0)
    head = md[{"original": "model", "neg_shared": "model_neg_shared", "group_neg_shared": "model_group_neg_shared"}[args.scheme]]
    eng = head.engine
    want_cpu = rank == 0 and args.cpu_seconds > 0 and not args.large   # the oracle copies whole tables
    U0 = eng.U.cpu().numpy().copy() if want_cpu else None
    V0, tower_params = None, None
    if args.model == "mf":
        V0 = eng.V.cpu().numpy().copy() if want_cpu else None
    else:
        tw = head.core.tower
        names_p = ("W", "Wc", "bc", "g1", "b1", "Wd", "bd", "gamma", "beta") if args.model == "cnn_embedding" else \
            ("W", "Wd", "bd", "gamma", "beta")
        tower_params = [getattr(tw, n).cpu().numpy().copy() for n in names_p]
        if getattr(tw, "nb", 0) > 1:   # filter bank: per-bank lists, pad rows / columns cut away
            g = lambda t: t.cpu().numpy().copy()  # noqa: E731
            wd_l, F_l, d_l, rows = tw.wd_l, tw.F_l, tw.d_l, tw.dense_rows().numpy()
            tower_params = dict(W=g(tw.W)[:, :wd_l], Wc=[g(tw.bank_Wc(i))[:, :wd_l, :F_l] for i in range(tw.nb)],
                                bc=[g(tw.bc)[i, :F_l] for i in range(tw.nb)], g1=[g(tw.g1)[i, :F_l] for i in range(tw.nb)],
                                b1=[g(tw.b1)[i, :F_l] for i in range(tw.nb)], Wd=g(tw.Wd)[rows][:, :d_l], bd=g(tw.bd)[:d_l],
                                gamma=g(tw.gamma)[:d_l], beta=g(tw.beta)[:d_l])
            if U0 is not None:
                U0 = U0[:, :eng.dim]
    uid = eng.ids(inp["uid"])
    cid = eng.ids(inp["cid"])
    W, K = args.warmup, args.steps
    device_links = None
    if args.large and args.device_links > 0:
        # SURVEY.md 8(d) input B: the link list lives in HBM and is generated there (counter-based PRNG), int32 SoA
        n_links = max(int(args.device_links), (W + K) * b)
        gen = torch.Generator(device=eng.device).manual_seed(20170813)
        ul = torch.empty(n_links, dtype=torch.int32, device=eng.device)
        cl = torch.empty(n_links, dtype=torch.int32, device=eng.device)
        for o in range(0, n_links, 1 << 27):
            m = min(1 << 27, n_links - o)
            for out, count in ((ul, inp["users"]), (cl, inp["items"])):
                r = torch.rand(m, generator=gen, device=eng.device).pow_(1.0 / 0.35).mul_(count)    # numpy's power(0.35)
                out[o:o + m] = r.to(torch.int32).clamp_(max=count - 1)
        uid, cid, device_links = ul, cl, n_links
    sampler = None
    neg = None
    if args.scheme == "original":
        sampler = MultinomialSampler(inp["dist"], inp["dist"].size, 1.0, 12345, device=local_rank)
        neg = torch.empty((W + K) * b * k, dtype=torch.int32, device=eng.device)

    n_u_buf = torch.zeros(W + K + 1, dtype=torch.int32, device=eng.device) if args.model != "mf" else None

    def run(first, count):
        u, c = uid[first * b:], cid[first * b:]
        n = None
        if args.scheme == "original":
            # in-loop sampling is inside the reference's timing window (train_original.py:42-70):
            # the whole window's negatives are drawn on device first, same stream order.
            n = neg[first * b * k:(first + count) * b * k]
            sampler.sample_batch_device(count * b * k, out=n)
        if n_u_buf is not None:
            return head.run_epoch(u, c, count, neg_all=n, n_u_out=n_u_buf[first:first + count])
        return head.run_epoch(u, c, count, neg_all=n)

    def sync():
        torch.cuda.synchronize()

    if W > 0:
        run(0, W)
    sync()
    kept = {}

    def window():
        sync()
        t0 = time.perf_counter()
        kept["losses"] = run(W, K)
        torch.cuda.synchronize()
        return time.perf_counter() - t0

    win_times = timed_windows(window)
    elapsed, win_stats = window_stats(win_times, K)
    losses = kept["losses"]
    if args.dump_stamps:
        import ctypes
        from nncf_amd import _lib
        nblk = 2 * (b // 16 + 8) * 8 + 1
        stamps = np.zeros(nblk * 32, dtype=np.int64)
        _lib.check(eng._L.nncf_model_debug_stamps(eng._h, stamps.ctypes.data_as(ctypes.POINTER(ctypes.c_longlong)), stamps.size))
        np.save(args.dump_stamps, stamps.reshape(nblk, 32))
    mean_loss = float(losses.double().mean().item())
    if not np.isfinite(mean_loss):
        raise SystemExit("ERROR (nan)! cost is %r" % mean_loss)  # utilities.py:33-35

    # ---- roofline leg: same steps again, dominant kernel bracketed by HIP events ----------
    L = args.content_len if not args.small else min(args.content_len, 40)
    keep = 1.0 - (args.dropout if args.model != "mf" else 0.0)
    mean_n_u = float(n_u_buf[W:W + K].double().mean().item()) if n_u_buf is not None else float(b)
    # per LAUNCH: a filter bank launches the conv kernels once per length -- the mean over the banks
    conv_flop = 2.0 * mean_n_u * L * (sum(bank_lens(args)) / float(len(bank_lens(args)))) * d * d
    kernels = {
        "cnn_conv_fwd_kernel": ("mfma", conv_flop, "2*n_u*L*fl*word_dim*F flop (fl: mean over the bank's lengths), n_u = %.1f unique items per launch" % mean_n_u),
        "cnn_conv_bwd_kernel": ("mfma", 2 * conv_flop, "dX and dWk: 4*n_u*L*fl*word_dim*F flop, n_u = %.1f" % mean_n_u),
        # N = the batch's unique items for group_neg_shared (6*b*n_u*d with the measured n_u), b otherwise
        "shared_tile_kernel": ("mfma", 6.0 * b * (mean_n_u if args.scheme == "group_neg_shared" and args.model != "mf" else b) * d,
                               "6*b*N*d flop (S, G.V, G^T.U); N = %.1f" % (mean_n_u if args.scheme == "group_neg_shared" and args.model != "mf" else b)),
        # the one-launch step computes S in both passes (8 b N d flop issued); the ALGORITHMIC work of the step is 6 b N d
        "fused_strip_kernel": ("mfma", 6.0 * b * (mean_n_u if args.scheme == "group_neg_shared" and args.model != "mf" else b) * d,
                               "6*b*N*d flop (S, G.V, G^T.U); the kernel issues 8*b*N*d (S recomputed by the item pass)"),
        "fused_wide_kernel": ("mfma", 6.0 * b * b * d,
                              "6*b*N*d flop (S, G.V, G^T.U); the kernel issues 8*b*N*d (S recomputed by the item pass)"),
        "pair_wide_kernel": ("mfma", 6.0 * b * b * d,
                             "6*b*N*d flop (S, G.V, G^T.U); the kernel issues 8*b*N*d (S recomputed by the item strips)"),
        "original_score_kernel": ("hbm", b * (4.0 * d * (2 + k) + 8 * (1 + k)), "read half of 8d(2+k)+8(1+k) B/link"),
        "tower_pool_kernel": ("hbm", 4.0 * mean_n_u * L * d * keep,
                              "4*L*word_dim*(1-dropout) B per unique item, n_u = %.1f unique items per launch" % mean_n_u),
        "tower_word_kernel": ("hbm", 4.0 * mean_n_u * L * d * keep,
                              "4*L*word_dim*(1-dropout) B of gradient rows per unique item, n_u = %.1f" % mean_n_u),
        "tower_bwd_kernel": ("mfma", 4.0 * b * d * d, "2*m*wd*d (dh) + 2*m*wd*d (dWd) flop"),
    }
    # dominant kernel of the step by time (profiles/): the one-launch step for the shared schemes with a pointwise loss,
    # the tile kernel for their pairwise losses, the per-positive score kernel for `original`, the conv backward for cnn
    pointwise = pconf.loss in ("skip-gram", "mse")
    shared_kernel = "fused_strip_kernel" if (pointwise and os.environ.get("NNCF_SHARED_PATH") != "tile") else "shared_tile_kernel"
    if shared_kernel == "fused_strip_kernel" and args.model == "mf" and b >= 4096 and os.environ.get("NNCF_FUSED_MT") != "1":
        shared_kernel = "fused_wide_kernel"     # step_fused.hip fused_row_tiles: 2 or 4 row tiles per workgroup
    if not pointwise and args.model == "mf" and (b > 2048 or os.environ.get("NNCF_SHARED_PATH") == "pairwide"):
        shared_kernel = "pair_wide_kernel"      # step_pair.hip: anchors -> wide strips -> commit
    default_kernel = {"mf": {"original": "original_score_kernel"}, "basic_embedding": {"original": "tower_word_kernel"}}.get(
        args.model, {}).get(args.scheme) or {"cnn_embedding": "cnn_conv_bwd_kernel"}.get(args.model, shared_kernel)
    names = [args.roofline_kernel or default_kernel]
    if args.model == "basic_embedding":
        names += [n for n in ("tower_pool_kernel", "tower_word_kernel", "tower_bwd_kernel",
                              "fused_strip_kernel", "shared_tile_kernel") if n not in names]
    if args.model == "mf":
        names += [n for n in ("fused_strip_kernel", "fused_wide_kernel", "shared_tile_kernel", "shared_apply_kernel",
                              "pair_commit_kernel", "pair_anchor_kernel") if n not in names]
    if args.model == "cnn_embedding":
        names += [n for n in ("cnn_conv_fwd_kernel", "tower_word_kernel", "shared_tile_kernel") if n not in names]
    # ... and every other kernel a step of this path may launch (those it does not launch report zero launches and drop
    # out): with ALL of a step's kernels bracketed the brackets can be calibrated against the step itself (below)
    names += [n for n in STEP_KERNELS if n not in names]
    prof_n = min(K, args.profile_launches)
    roofs = []
    for name in names:
        nengine.profile_begin(name, prof_n)
        run(W, prof_n)
        raw_ms, launches, ov_ms = nengine.profile_end()
        if launches == 0:
            continue
        # duration: the rocprofv3 --kernel-trace average of this exact workload when one is committed under
        # profiles/ (first kernel only: the table holds the dominant one), else the raw event bracket -- which
        # includes the marker packets' own cost (empty_bracket_ms), i.e. errs on the slow side
        rp_ms = rocprof_duration_ms(args, conf, d, name)
        bound, work, what = kernels.get(name, ("hbm", 0.0, "n/a"))
        roofs.append(roof_entry(name, bound, work, what, raw_ms, ov_ms, launches, rp_ms))
    if not roofs:
        roofs = [{"kernel": names[0], "bound": None, "achieved": None, "peak": None, "unit": None, "frac": None, "traffic": None,
                  "note": "kernel not launched by this configuration"}]
    calibrate_brackets(roofs, elapsed / K * 1e3, prof_n)
    split_on = os.environ.get("NNCF_WIDE_F16SPLIT", "1") != "0"
    # Guarded mode (the default): the bracket named fused_wide_kernel is the f16-plane launch; when the batch's rows fall
    # outside the fp16 domain it returns at once and the work runs under fused_wide_twin_kernel (the fp32-pipe twin).  The
    # roofline line must be the launch that did the work, priced on the pipe it ran on: the larger bracket wins.
    twin = next((r for r in roofs if r.get("kernel") == "fused_wide_twin_kernel" and r.get("kernel_ms")), None)
    for r in roofs:
        if r.get("kernel") in WIDE_KERNELS and r.get("kernel_ms"):
            on_planes = split_on
            if r["kernel"] == "fused_wide_kernel" and twin is not None and twin["kernel_ms"] > r["kernel_ms"]:
                planes_ms = r["kernel_ms"]
                for key in ("kernel_ms", "event_bracket_ms", "empty_bracket_ms", "bracket_cost_ms", "launches", "launches_per_step"):
                    if key in twin:
                        r[key] = twin[key]
                r["achieved"] = r["algorithmic_per_launch"] / (r["kernel_ms"] * 1e-3) / 1e12
                r["frac"] = r["achieved"] / PEAK_F32_MFMA_TFLOPS
                r["frac_uncorrected_bracket"] = r["algorithmic_per_launch"] / (r["event_bracket_ms"] * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS
                r["f16_domain"] = ("OUTSIDE: the f16-plane launch returned at once (%.4f ms) and the fp32-pipe twin took the step; "
                                   "this line is the twin's bracket, priced on the fp32 MFMA" % planes_ms)
                on_planes = False
            elif r["kernel"] == "fused_wide_kernel" and twin is not None:
                r["f16_domain"] = "inside: the f16-plane launch took the step (the twin's bracket: %.4f ms)" % twin["kernel_ms"]
            price_on_f16_pipe(r, on_planes)
    roof = roofs[0]
    roof["traffic"] = measured_traffic(args, conf, d, roof["kernel"])

    cpu = None
    if want_cpu:
        cpu = cpu_baseline(args, inp, U0, V0, tower_params)

    # Recall@50 (the second half of BASELINE's metric) of the model as trained so far: whole@50
    # evaluator on the GPU (U.V^T over the candidate items + per-user top-k), outside every timed region
    recall = None
    if rank == 0 and not args.large:
        from nncf_amd.objectives import Evaluator
        pconf.eval_topk = 50
        ev = Evaluator(helper, helper.data_spec, pconf)
        tr_res, te_res = ev.run(md["model_neg_shared"], verbose=False, eval_scheme="whole")
        # the evaluator's own roofline entry: one whole@50 pass = the train AND the test candidate sets
        # through nncf_eval_whole (csrc/eval.hip eval_fused_kernel), bracketed with HIP events by the library's profile hook
        n_cand_all = len(ev.train_items) + len(ev.test_items)
        e_flop = 2.0 * ev.user_count * n_cand_all * d
        nengine.profile_begin("eval_fused_kernel", 64)
        for _ in range(5):
            ev.run(md["model_neg_shared"], verbose=False, eval_scheme="whole")
        torch.cuda.synchronize()
        e_ms, e_n, e_ov = nengine.profile_end()
        eval_roof = None
        if e_n:
            pass_ms = (e_ms - e_ov) * e_n / 5.0
            eval_roof = {"kernel": "eval_fused_kernel", "what": "whole@50 over %d users x (%d train + %d test candidates), d = %d: U.V^T on the "
                         "fp32 matrix pipe with per-user top-k lists in LDS -- no score matrix (round 4: predict_tile + eval_topk, 2.56 ms, "
                         "1.04 GB of scores written per pass)" % (ev.user_count, len(ev.train_items), len(ev.test_items), d),
                         "pass_ms": pass_ms, "launches_per_pass": e_n / 5.0, "bound": "mfma", "achieved": e_flop / (pass_ms * 1e-3) / 1e12,
                         "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": e_flop / (pass_ms * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS,
                         "traffic": evaluator_traffic(d), "algorithmic_flop_per_pass": e_flop}
        recall = {"train": float(tr_res["recall@50"]), "test_cold_items": float(te_res["recall@50"]),
                  "after_steps": int(W + K * len(win_times) + prof_n * len(names)),
                  "note": "whole@50 after the warm-up, timed and roofline passes of this run (a short run, not a converged model)"
                          + ("; learn_rate 0.01 is the demo scripts' ADAM step size -- plain SGD barely moves with it "
                             "(--optimizer adam: train 0.68 / cold-item test 0.27 after the same steps)"
                             if args.optimizer == "sgd" else "")}

    recall_trained = None
    if rank == 0 and not args.large and args.recall_epochs > 0:
        recall_trained = trained_recall(args, inp["data"], local_rank)
        recall_trained["oracle_float64"] = oracle_recall_record(args, recall_trained)

    if rank == 0:
        value = K * b * world / elapsed
        tower_txt = ("basic_embedding (mean-pool text tower L=%d, vocab=%d, word-row dropout %g, Dense+BN+ReLU)"
                     % (L, args.vocab, args.dropout)
                     if args.model == "basic_embedding" else
                     "cnn_embedding (CNN text tower L=%d, vocab=%d, word-row dropout %g, Conv1D(%d filters, length %s)+BN+ReLU+"
                     "Dropout(%g)+avg-pool, Dense+BN+ReLU)" % (L, args.vocab, args.dropout, d, args.filter_lens or str(args.filter_len), args.conv_dropout)
                     if args.model == "cnn_embedding" else "id-table towers (model_choice=mf)")
        line = {
            "metric": "positive-links/sec (train step)", "value": value, "unit": "positive-links/s",
            "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": elapsed / K * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": ("f32 (products of the large-batch kernel: fp16 hi / lo planes on the f16 matrix pipe, fp32 accumulate -- error "
                      "against float64 below the fp32 MFMA's)" if (roof.get("kernel") in WIDE_KERNELS and split_on) else "f32"),
            "windows": win_stats,
            "data": ("synthetic config-4-shaped (10M users x 1M items, Zipf links), random-init parameters" if args.large
                     else "synthetic CiteULike-a-shaped (16K users x 17K items x 200K links%s), random-init parameters"
                     % (", item text" if args.model != "mf" else "")),
            "config": {"workload": "%s, %s d=%d, train_scheme=%s, loss=%s, b=%d%s, %s"
                                   % ("config-4-shaped 10M x 1M" if args.large else "CiteULike-a-shaped", tower_txt, d,
                                      args.scheme, conf.loss, b, (", k=%d" % k) if k else "",
                                      {"sgd": "sparse SGD", "adam": "Keras Adam (dense)",
                                       "sparse_adam": "lazy-sparse Adam"}[args.optimizer]),
                       "global_batch": b * world, "parallelism": "1 process per GPU"},
            "mean_loss": mean_loss,
            "mean_unique_items_per_step": mean_n_u if args.model != "mf" else None,
            "device_generated_links": device_links,
            "recall_at_50": recall,
            "recall_at_50_trained": recall_trained,
            "evaluator": eval_roof if (rank == 0 and not args.large) else None,
            "roofline": roof,
            "roofline_other_kernels": roofs[1:],
            "cpu_baseline": cpu,
        }
        emit(line)


if __name__ == "__main__":
    main()
