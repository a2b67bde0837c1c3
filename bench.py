#!/usr/bin/env python3
"""Headline benchmark: positive-links/sec of the NNCF training step on MI355X.

    python bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch of b = 512 positive links of the synthetic
CiteULike-a-shaped data set (16K users x 17K items x 200K links, item text L = 200, d = 100) of
BASELINE configs[1]: basic_embedding, train_scheme = neg_shared, as the demo scripts run it
(--conf_choice best: word-row dropout 0.5) --
    duplicate plan + tf.unique -> text tower forward of the batch's unique items (word-row gather +
    mean, Dense, BatchNorm statistics; BN + ReLU applied when the step kernels load an item row)
    -> U.V^T shared-negative tile -> loss gradient -> back-GEMMs -> sparse SGD on the user rows
    -> tower backward (BN / Dense by MFMA, sparse SGD on the word table)
with all inputs (link list, content matrix, parameters) resident in HBM before the timed region
and no host synchronisation between steps: the trainer loop (models/train_neg_shared.py:46-59)
runs inside one native call (nncf_tower_epoch / nncf_epoch_*).  --model mf runs the id-table
towers alone; --gpus N > 1 runs the item-sharded id-table path.  Prints ONE JSON line (rank 0).

The line also carries
  roofline     -- the dominant kernel of the step by time: algorithmic bytes / flops per launch
                  (stated in `algorithmic_what`) / its mean duration measured with
                  HIP events on the launch stream in a second pass over the same steps
                  (roofline_other_kernels: the same for the other heavy kernels);
  cpu_baseline -- the numpy oracle of the same step (oracle/step_oracle.py, kind "port")
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

PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: fp32-input MFMA = fp32 vector peak
PEAK_HBM_GBS = 8000.0          # HBM3E spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3900)     # 10 epochs of 390 steps
    ap.add_argument("--warmup", type=int, default=390)
    ap.add_argument("--scheme", default="neg_shared", choices=["original", "neg_shared", "group_neg_shared"])
    ap.add_argument("--loss", default=None)
    ap.add_argument("--model", default="basic_embedding", choices=["basic_embedding", "cnn_embedding", "mf"],
                    help="basic_embedding = mean-pool text tower (BASELINE configs[1]); cnn_embedding = CNN text tower "
                         "(configs[4]: use --scheme group_neg_shared); mf = id-table towers only")
    ap.add_argument("--conv-dropout", type=float, default=0.1, help="cnn_embedding: Dropout after the conv ('best' preset)")
    ap.add_argument("--filter-len", type=int, default=3, help="cnn_embedding: Convolution1D filter length ('best' preset)")
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
    ap.add_argument("--profile-launches", type=int, default=1000)
    ap.add_argument("--roofline-kernel", default=None, help="HIP kernel to bracket (default: per workload)")
    ap.add_argument("--small", action="store_true", help="2K x 2K x 20K data (quick functional run)")
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
            "optimizer": args.optimizer, "num_filters": [args.dim], "filter_lengths": [args.filter_len],
            "conv_dropout_rate": args.conv_dropout}


def conf_for(args):
    from oracle.step_oracle import OracleConf  # only for the cpu_baseline leg
    pd = param_dict_for(args)
    return OracleConf(batch_size_p=pd["batch_size_p"], num_negatives=pd["num_negatives"], loss=pd["loss"],
                      loss_gamma=pd["loss_gamma"], neg_loss_weight=pd["neg_loss_weight"], u_reg=pd["u_reg"],
                      learn_rate=pd["learn_rate"], model_choice=args.model, optimizer=args.optimizer)


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


def cpu_baseline(args, conf, inp, U0, V0, tower_params):
    """numpy oracle of the same step on the host cores, bounded sample."""
    from oracle import step_oracle as so
    from oracle.sampler_oracle import OracleSampler
    try:
        from threadpoolctl import threadpool_info
        cores = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
    except Exception:
        cores = os.cpu_count() or 1
    tower = None
    if tower_params is not None:
        cls = so.CnnTowerOracle if args.model == "cnn_embedding" else so.TowerOracle
        tower = cls(inp["data"]["C"], *tower_params, lr=conf.learn_rate, dtype=np.float32)
    o = so.StepOracle(U0, V0, conf, dtype=np.float32, tower=tower)
    b, k = conf.batch_size_p, conf.num_negatives
    sampler = OracleSampler(inp["dist"], 1.0, 12345) if args.scheme == "original" else None
    t0 = time.time()
    steps = 0
    max_steps = min(args.steps, len(inp["uid"]) // b)
    drop = args.dropout if tower is not None else 0.0
    mask_rs = np.random.RandomState(29)
    while steps < max_steps and (time.time() - t0 < args.cpu_seconds or steps < 5):
        if drop > 0:   # Embedding(dropout): a fresh row mask per batch (mean_pool.py:56-59)
            o.row_scale = ((mask_rs.rand(tower.W.shape[0]) < 1 - drop) / (1 - drop)).astype(np.float32)
        if args.model == "cnn_embedding" and args.conv_dropout > 0:   # Dropout(conv_dropout_rate), cnn_model.py:97
            kc = 1 - args.conv_dropout
            tower.conv_mask = ((mask_rs.rand(b * (1 + k), tower.C.shape[1], tower.Wc.shape[2]) < kc) / kc).astype(np.float32)
        u = inp["uid"][steps * b:(steps + 1) * b]
        c = inp["cid"][steps * b:(steps + 1) * b]
        if args.scheme == "original":
            negs = sampler.sample_batch(b * k)
            uu, cc, yy = so.assemble_original(np.stack([u, c, np.ones(b, dtype=np.int64)], 1), negs, conf.loss)
            o.step_original(uu, cc, yy)
        elif args.scheme == "neg_shared":
            o.step_neg_shared(u, c)
        else:
            o.step_group_neg_shared(u, c)
        steps += 1
    dt = time.time() - t0
    return {"value": steps * b / dt, "unit": "positive-links/s", "cores": int(cores), "kind": "port",
            "sample": "%d steps of the same %s/%s workload (numpy fp32 oracle, %.1f s)" % (steps, args.model, args.scheme, dt)}


_T0 = time.time()


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


def measured_traffic(args, conf, d):
    """Bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
    (profiles/*_traffic.json: FETCH_SIZE x2 per the gfx950 correction + WRITE_SIZE, KiB -> bytes),
    when one exists for exactly this workload; else None."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        table = json.load(open(path))
    except Exception:
        return None
    if args.optimizer != "sgd" or args.large:
        return None   # the committed PMC passes are of the sparse-SGD CiteULike-shaped runs
    key = "%s/%s/%s/b%d/d%d%s" % (args.model, args.scheme, conf.loss, conf.batch_size_p, d, "/small" if args.small else "")
    ent = table.get(key)
    return ent["bytes_per_launch"] if ent else None


def rocprof_duration_ms(args, conf, d):
    """Average duration of the dominant kernel in the committed rocprofv3 --kernel-trace --stats summary of
    this workload (profiles/traffic.json carries it next to the PMC bytes), for comparison with the live
    event-bracket figure; None when there is no committed profile of exactly this workload."""
    try:
        table = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    except Exception:
        return None
    if args.optimizer != "sgd" or args.large:
        return None
    key = "%s/%s/%s/b%d/d%d%s" % (args.model, args.scheme, conf.loss, conf.batch_size_p, d, "/small" if args.small else "")
    us = (table.get(key) or {}).get("rocprof_avg_us")
    return us * 1e-3 if us else None


def main_data_parallel(args, rank, world, local_rank):
    """N > 1, model_choice = basic_embedding / cnn_embedding: one process per GPU, users + their links
    partitioned by user range, text tower replicated, ONE RCCL all-reduce of the tower gradients per
    global step (nncf_amd/sharded.py: DataParallelTower).  Weak scaling: b positives per rank per step."""
    import torch
    import torch.distributed as dist
    from nncf_amd import engine as nengine
    from nncf_amd.conf import get_conf
    from nncf_amd.data_utils import get_data
    from nncf_amd.model_framework import get_model
    from nncf_amd.sharded import DataParallelTower, HipTowerReplica, partition_links_by_user
    trace("data-parallel: making data")
    data = make_data(args)
    trace("data made")
    users = int(data["train"][:, 0].max()) + 1
    pconf = get_conf("synthetic_citeulike", "best", param_dict_for(args), model_choice=args.model)
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

    def run(first, count, keep):
        tr.run_epoch(uid[first * b:(first + count) * b], cid[first * b:(first + count) * b], count, b, losses if keep else None)

    trace("warmup")
    run(0, W, False)
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    trace("timed region")
    t0 = time.perf_counter()
    run(W, K, True)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    trace("timed region done")
    dist.barrier()
    t = torch.tensor([t1 - t0], dtype=torch.float64, device=eng.device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    ml = losses.double().mean()
    dist.all_reduce(ml)
    mean_loss = float(ml.item()) / world
    if not np.isfinite(mean_loss):
        raise SystemExit("ERROR (nan)! cost is %r" % mean_loss)
    # roofline leg (all ranks step together; rank 0's brackets are reported)
    L = args.content_len if not args.small else min(args.content_len, 40)
    keep = 1.0 - args.dropout
    cnn = args.model == "cnn_embedding"
    name = args.roofline_kernel or ("cnn_conv_bwd_kernel" if cnn else "tower_word_kernel")
    prof_n = min(K, args.profile_launches)
    nengine.profile_begin(name, prof_n)
    run(W, prof_n, False)
    raw_ms, launches, ov_ms = nengine.profile_end()
    kern_ms = max(raw_ms - ov_ms, 1e-9)
    if name.startswith("cnn_conv"):   # n_u = unique items of the bracketed steps (counted here from the id stream)
        n_u = float(np.mean([len(np.unique(stream[t * b:(t + 1) * b, 1])) for t in range(W, W + prof_n)]))
        mult = 2 if name == "cnn_conv_fwd_kernel" else 4
        work = mult * n_u * L * args.filter_len * d * d
        roof = {"bound": "mfma", "achieved": work / (kern_ms * 1e-3) / 1e12, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                "what": "%d*n_u*L*fl*word_dim*F flop, n_u = %.1f unique items per launch" % (mult, n_u)}
    else:
        n_u = float(np.mean([len(np.unique(stream[t * b:(t + 1) * b, 1])) for t in range(W, W + prof_n)]))
        work = 4.0 * n_u * L * d * keep
        roof = {"bound": "hbm", "achieved": work / (kern_ms * 1e-3) / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                "what": "4*L*word_dim*(1-dropout) B of gradient rows per unique item, n_u = %.1f" % n_u}
    tower_txt = ("cnn_embedding (CNN text tower L=%d, vocab=%d, word-row dropout %g, Conv1D(%d filters, length %d)+BN+ReLU+"
                 "Dropout(%g)+avg-pool, Dense+BN+ReLU)" % (L, args.vocab, args.dropout, d, args.filter_len, args.conv_dropout)
                 if cnn else "basic_embedding (mean-pool text tower L=%d, vocab=%d, word-row dropout %g, Dense+BN+ReLU)"
                 % (L, args.vocab, args.dropout))
    if rank == 0:
        line = {
            "metric": "positive-links/sec (train step)", "value": K * b * world / elapsed, "unit": "positive-links/s",
            "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": elapsed / K * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic CiteULike-a-shaped (16K users x 17K items x 200K links, item text), random-init parameters",
            "config": {"workload": "CiteULike-a-shaped, %s d=%d, train_scheme=%s, loss=%s, b=%d per GPU, sparse SGD"
                                   % (tower_txt, d, args.scheme, pconf.loss, b),
                       "global_batch": b * world,
                       "parallelism": "1 process per GPU; users + their links partitioned by user range, text tower "
                                      "replicated; one RCCL all-reduce of the tower gradients (%.2f MB) per step"
                                      % (rep.grad.numel() * 4 / 1e6)},
            "mean_loss": mean_loss,
            "roofline": {"kernel": name, "bound": roof["bound"], "achieved": roof["achieved"], "peak": roof["peak"],
                         "unit": roof["unit"], "frac": roof["achieved"] / roof["peak"], "traffic": None,
                         "algorithmic_per_launch": work, "algorithmic_what": roof["what"],
                         "kernel_ms": kern_ms, "event_bracket_ms": raw_ms, "empty_bracket_ms": ov_ms, "launches": launches},
            "cpu_baseline": None,
        }
        emit(line)
    dist.destroy_process_group()


def main_sharded(args, conf, rank, world, local_rank):
    """N > 1, --model mf: one process per GPU, user/item tables range-sharded, three RCCL all-to-alls
    per global step (nncf_amd/sharded.py: ShardedNegShared; the north-star's config-4 path).
    Weak scaling: b positives per rank per step."""
    import torch
    import torch.distributed as dist
    from nncf_amd.sharded import HipBackend, ShardedNegShared, partition_links_by_user, shard_size
    if args.model in ("basic_embedding", "cnn_embedding"):
        return main_data_parallel(args, rank, world, local_rank)
    if args.model != "mf":
        raise SystemExit("--gpus > 1 is built for the text-tower models (data parallel) and --model mf (item-sharded)")
    if args.scheme != "neg_shared":
        raise SystemExit("--gpus > 1 is built for train_scheme=neg_shared (north-star config 4 path)")
    conf = conf_for(args)
    data = make_data(args)   # --large: BASELINE configs[3], 10M users x 1M items (only the links the run consumes)
    users, items = int(data["train"][:, 0].max()) + 1, int(data["C"].shape[0])
    b, d = conf.batch_size_p, args.dim
    mine = partition_links_by_user(data["train"], users, world)[rank].copy()
    rs = np.random.RandomState(4224 + rank)
    need = (args.warmup + args.steps) * b
    chunks, have = [], 0
    while have < need:
        rs.shuffle(mine)
        chunks.append(mine.copy())
        have += len(mine)
    stream = np.concatenate(chunks, 0)[:need]
    kw = dict(user_count=users, item_count=items, dim=d, batch_size_p=b, num_negatives=0, loss=conf.loss,
              loss_gamma=conf.loss_gamma, neg_loss_weight=conf.neg_loss_weight, u_reg=conf.u_reg,
              learn_rate=conf.learn_rate, normalize_user=conf.normalize_user, normalize_item=conf.normalize_item)
    us, isz = shard_size(users, world), shard_size(items, world)
    be = HipBackend(kw, max(1, min(us, users - rank * us)), max(1, min(isz, items - rank * isz)), local_rank,
                    world=world, item_shard=isz, item_offset=rank * isz, seed=rank)
    tr = ShardedNegShared(be, users, items, b, d)
    uid, cid = be.ids(stream[:, 0] - rank * us), be.ids(stream[:, 1])   # user ids relative to my range
    W, K = args.warmup, args.steps
    losses = torch.zeros(K, dtype=torch.float32, device=be.device)

    def run(first, count, keep):
        for t in range(first, first + count):
            l = tr.step(uid[t * b:(t + 1) * b], cid[t * b:(t + 1) * b])
            if keep:
                losses[t - first].copy_(l[0])

    run(0, W, False)
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(W, K, True)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    dist.barrier()
    t = torch.tensor([t1 - t0], dtype=torch.float64, device=be.device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    ml = losses.double().mean()
    dist.all_reduce(ml)
    mean_loss = float(ml.item()) / world
    if not np.isfinite(mean_loss):
        raise SystemExit("ERROR (nan)! cost is %r" % mean_loss)
    if rank == 0:
        line = {
            "metric": "positive-links/sec (train step)", "value": K * b * world / elapsed, "unit": "positive-links/s",
            "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": elapsed / K * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": ("synthetic config-4-shaped (10M users x 1M items, Zipf links), random-init tables" if args.large else
                     "synthetic CiteULike-a-shaped (16K users x 17K items x 200K links), random-init tables"),
            "config": {"workload": "%s, id-table towers (model_choice=mf), item table sharded by id "
                                   "range, d=%d, train_scheme=neg_shared, loss=%s, b=%d per GPU, sparse SGD"
                                   % ("config-4-shaped 10M x 1M" if args.large else "CiteULike-a-shaped", d, conf.loss, b),
                       "global_batch": b * world,
                       "parallelism": "1 process per GPU; users + their links and item rows range-sharded; "
                                      "3 RCCL all-to-alls per step (ids, rows, gradients)"},
            "mean_loss": mean_loss,
            "roofline": None, "cpu_baseline": None,
        }
        emit(line)
    dist.destroy_process_group()


def main():
    args = parse()
    if args.large:
        args.model, args.dim = "mf", 128
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
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

    conf = conf_for(args)
    if dist_on:
        return main_sharded(args, conf, rank, world, local_rank)
    inp = build_inputs(args, conf, rank, world)
    b, k, d = conf.batch_size_p, conf.num_negatives, args.dim
    from nncf_amd.conf import get_conf
    from nncf_amd.data_utils import get_data
    from nncf_amd.model_framework import get_model
    from nncf_amd import engine as nengine
    pconf = get_conf("synthetic_citeulike", "best", param_dict_for(args), model_choice=args.model)
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
    uid = eng.ids(inp["uid"])
    cid = eng.ids(inp["cid"])
    W, K = args.warmup, args.steps
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
    t0 = time.perf_counter()
    losses = run(W, K)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    mean_loss = float(losses.double().mean().item())
    if not np.isfinite(mean_loss):
        raise SystemExit("ERROR (nan)! cost is %r" % mean_loss)  # utilities.py:33-35

    # ---- roofline leg: same steps again, dominant kernel bracketed by HIP events ----------
    L = args.content_len if not args.small else min(args.content_len, 40)
    keep = 1.0 - (args.dropout if args.model != "mf" else 0.0)
    mean_n_u = float(n_u_buf[W:W + K].double().mean().item()) if n_u_buf is not None else float(b)
    conv_flop = 2.0 * mean_n_u * L * args.filter_len * d * d   # per launch, n_u = mean unique items of the timed steps
    kernels = {
        "cnn_conv_fwd_kernel": ("mfma", conv_flop, "2*n_u*L*fl*word_dim*F flop, n_u = %.1f unique items per launch" % mean_n_u),
        "cnn_conv_bwd_kernel": ("mfma", 2 * conv_flop, "dX and dWk: 4*n_u*L*fl*word_dim*F flop, n_u = %.1f" % mean_n_u),
        "shared_tile_kernel": ("mfma", 6.0 * b * b * d, "6*b*N*d flop (S, G.V, G^T.U; N = b upper bound)"),
        "original_score_kernel": ("hbm", b * (4.0 * d * (2 + k) + 8 * (1 + k)), "read half of 8d(2+k)+8(1+k) B/link"),
        "tower_pool_kernel": ("hbm", 4.0 * mean_n_u * L * d * keep,
                              "4*L*word_dim*(1-dropout) B per unique item, n_u = %.1f unique items per launch" % mean_n_u),
        "tower_word_kernel": ("hbm", 4.0 * mean_n_u * L * d * keep,
                              "4*L*word_dim*(1-dropout) B of gradient rows per unique item, n_u = %.1f" % mean_n_u),
        "tower_bwd_kernel": ("mfma", 4.0 * b * d * d, "2*m*wd*d (dh) + 2*m*wd*d (dWd) flop"),
    }
    default_kernel = {"mf": {"original": "original_score_kernel"}}.get(args.model, {}).get(args.scheme) or \
        {"mf": "shared_tile_kernel", "cnn_embedding": "cnn_conv_bwd_kernel"}.get(args.model, "tower_word_kernel")
    names = [args.roofline_kernel or default_kernel]
    if args.model == "basic_embedding":
        names += [n for n in ("tower_pool_kernel", "tower_word_kernel", "tower_bwd_kernel", "shared_tile_kernel")
                  if n not in names]
    if args.model == "cnn_embedding":
        names += [n for n in ("cnn_conv_fwd_kernel", "tower_word_kernel", "shared_tile_kernel") if n not in names]
    prof_n = min(K, args.profile_launches)
    roofs = []
    for name in names:
        nengine.profile_begin(name, prof_n)
        run(W, prof_n)
        raw_ms, launches, ov_ms = nengine.profile_end()
        if launches == 0:
            continue
        kern_ms = max(raw_ms - ov_ms, 1e-9)   # bracket minus the empty-bracket (marker packet) cost
        bound, work, what = kernels.get(name, ("hbm", 0.0, "n/a"))
        if bound == "hbm":
            ach, peak, unit = work / (kern_ms * 1e-3) / 1e9, PEAK_HBM_GBS, "GB/s"
        else:
            ach, peak, unit = work / (kern_ms * 1e-3) / 1e12, PEAK_F32_MFMA_TFLOPS, "TFLOP/s"
        roofs.append({"kernel": name, "bound": bound, "achieved": ach, "peak": peak,
                      "unit": unit, "frac": ach / peak, "traffic": None, "algorithmic_per_launch": work,
                      "algorithmic_what": what, "kernel_ms": kern_ms, "event_bracket_ms": raw_ms,
                      "empty_bracket_ms": ov_ms, "launches": launches})
    roof = roofs[0]
    roof["traffic"] = measured_traffic(args, conf, d)
    # the profiler's duration of the same kernel (start to end of the dispatch, incl. what the event bracket's
    # empty-bracket correction removes): committed summary under profiles/, None without one
    roof["rocprof_kernel_ms"] = rocprof_duration_ms(args, conf, d)

    cpu = None
    if want_cpu:
        cpu = cpu_baseline(args, conf, inp, U0, V0, tower_params)

    # Recall@50 (the second half of BASELINE's metric) of the model as trained so far: whole@50
    # evaluator on the GPU (U.V^T over the candidate items + per-user top-k), outside every timed region
    recall = None
    if rank == 0 and not args.large:
        from nncf_amd.objectives import Evaluator
        pconf.eval_topk = 50
        tr_res, te_res = Evaluator(helper, helper.data_spec, pconf).run(md["model_neg_shared"], verbose=False,
                                                                        eval_scheme="whole")
        recall = {"train": float(tr_res["recall@50"]), "test_cold_items": float(te_res["recall@50"]),
                  "after_steps": int(W + K + prof_n * len(names)),
                  "note": "whole@50 after the warm-up, timed and roofline passes of this run (a short run, not a converged model)"
                          + ("; learn_rate 0.01 is the demo scripts' ADAM step size -- plain SGD barely moves with it "
                             "(--optimizer adam: train 0.68 / cold-item test 0.27 after the same steps)"
                             if args.optimizer == "sgd" else "")}

    if rank == 0:
        value = K * b * world / elapsed
        tower_txt = ("basic_embedding (mean-pool text tower L=%d, vocab=%d, word-row dropout %g, Dense+BN+ReLU)"
                     % (L, args.vocab, args.dropout)
                     if args.model == "basic_embedding" else
                     "cnn_embedding (CNN text tower L=%d, vocab=%d, word-row dropout %g, Conv1D(%d filters, length %d)+BN+ReLU+"
                     "Dropout(%g)+avg-pool, Dense+BN+ReLU)" % (L, args.vocab, args.dropout, d, args.filter_len, args.conv_dropout)
                     if args.model == "cnn_embedding" else "id-table towers (model_choice=mf)")
        line = {
            "metric": "positive-links/sec (train step)", "value": value, "unit": "positive-links/s",
            "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": elapsed / K * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
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
            "recall_at_50": recall,
            "roofline": roof,
            "roofline_other_kernels": roofs[1:],
            "cpu_baseline": cpu,
        }
        emit(line)


if __name__ == "__main__":
    main()
