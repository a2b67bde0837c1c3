#!/usr/bin/env python3
"""Headline benchmark: positive-links/sec of the NNCF training step on MI355X.

    python bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch of b = 512 positive links of the
synthetic CiteULike-a-shaped data set (16K users x 17K items x 200K links, d = 100):
gather -> U.V^T shared-negative tile -> loss gradient -> back-GEMMs -> sparse SGD scatter
update, with all inputs (link list, tables) resident in HBM before the timed region and no
host synchronisation between steps (the trainer loop models/train_neg_shared.py:46-59 runs
inside nncf_epoch_neg_shared).  Prints ONE JSON line (rank 0).

The line also carries
  roofline     -- the dominant kernel (the U.V^T tile kernel): algorithmic flops per launch
                  (6*b*b*d) / its mean duration measured with HIP events on
                  the launch stream in a second pass over the same steps;
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
    ap.add_argument("--model", default="mf", choices=["mf"])
    ap.add_argument("--dim", type=int, default=100)
    ap.add_argument("--batch", type=int, default=512)
    ap.add_argument("--negatives", type=int, default=10)
    ap.add_argument("--chop", type=int, default=5)
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the cpu_baseline leg (0 = skip)")
    ap.add_argument("--profile-launches", type=int, default=2000)
    ap.add_argument("--small", action="store_true", help="2K x 2K x 20K data (quick functional run)")
    return ap.parse_args()


def conf_for(args):
    from oracle.step_oracle import OracleConf  # only for the cpu_baseline leg / defaults table
    loss = args.loss or ("log-loss" if args.scheme == "group_neg_shared" else "skip-gram")
    gamma = {"max-margin": 0.1, "log-loss": 10.0}.get(loss, 10.0)
    lam = 8.0 if loss == "mse" else 128.0
    lr = 0.001 if loss == "mse" else 0.01
    k = args.negatives if args.scheme == "original" else 0
    return OracleConf(batch_size_p=args.batch, num_negatives=k, loss=loss, loss_gamma=gamma, neg_loss_weight=lam,
                      u_reg=1e-6, learn_rate=lr, model_choice=args.model)


def build_inputs(args, conf, rank, world):
    """Shuffled (and for group_neg_shared: stratified) link stream long enough for warmup+steps."""
    from nncf_amd.synthetic import make_citeulike_like
    from nncf_amd.data_utils import group_shuffle_train, degree_dist
    if args.small:
        data = make_citeulike_like(users=2000, items=2000, train_links=20000, seed=7)
    else:
        data = make_citeulike_like()
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


def cpu_baseline(args, conf, inp, U0, V0):
    """numpy oracle of the same step on the host cores, bounded sample."""
    from oracle import step_oracle as so
    from oracle.sampler_oracle import OracleSampler
    try:
        from threadpoolctl import threadpool_info
        cores = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
    except Exception:
        cores = os.cpu_count() or 1
    o = so.StepOracle(U0, V0, conf, dtype=np.float32)
    b, k = conf.batch_size_p, conf.num_negatives
    sampler = OracleSampler(inp["dist"], 1.0, 12345) if args.scheme == "original" else None
    t0 = time.time()
    steps = 0
    max_steps = min(args.steps, len(inp["uid"]) // b)
    while steps < max_steps and (time.time() - t0 < args.cpu_seconds or steps < 10):
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
            "sample": "%d steps of the same %s workload (numpy fp32 oracle, %.1f s)" % (steps, args.scheme, dt)}


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
    key = "%s/%s/b%d/d%d%s" % (args.scheme, conf.loss, conf.batch_size_p, d, "/small" if args.small else "")
    ent = table.get(key)
    return ent["bytes_per_launch"] if ent else None


def main_sharded(args, conf, rank, world, local_rank):
    """N > 1: one process per GPU, user/item tables range-sharded, three RCCL all-to-alls per
    global step (nncf_amd/sharded.py).  Weak scaling: b positives per rank per step."""
    import torch
    import torch.distributed as dist
    from nncf_amd.sharded import HipBackend, ShardedNegShared, partition_links_by_user, shard_size
    from nncf_amd.synthetic import make_citeulike_like
    if args.scheme != "neg_shared":
        raise SystemExit("--gpus > 1 is built for train_scheme=neg_shared (north-star config 4 path)")
    data = make_citeulike_like(users=2000, items=2000, train_links=20000, seed=7) if args.small else make_citeulike_like()
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
            "data": "synthetic CiteULike-a-shaped (16K users x 17K items x 200K links), random-init tables",
            "config": {"workload": "CiteULike-a-shaped, id-table towers (model_choice=%s) d=%d, train_scheme=neg_shared, "
                                   "loss=%s, b=%d per GPU, sparse SGD" % (args.model, d, conf.loss, b),
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
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    if dist_on:
        dist.barrier()
    from nncf_amd.engine import Engine
    from nncf_amd.sampler import MultinomialSampler

    conf = conf_for(args)
    if dist_on:
        return main_sharded(args, conf, rank, world, local_rank)
    inp = build_inputs(args, conf, rank, world)
    b, k, d = conf.batch_size_p, conf.num_negatives, args.dim
    eng = Engine(inp["users"], inp["items"], d, b, k, conf.loss, conf.loss_gamma, conf.neg_loss_weight, conf.u_reg,
                 conf.learn_rate, conf.normalize_user, conf.normalize_item, device=local_rank, seed=0)
    U0, V0 = eng.U.cpu().numpy().copy(), eng.V.cpu().numpy().copy()
    uid = eng.ids(inp["uid"])
    cid = eng.ids(inp["cid"])
    W, K = args.warmup, args.steps
    sampler = None
    neg = None
    if args.scheme == "original":
        sampler = MultinomialSampler(inp["dist"], inp["dist"].size, 1.0, 12345, device=local_rank)
        neg = torch.empty((W + K) * b * k, dtype=torch.int32, device=eng.device)

    def run(first, count):
        u, c = uid[first * b:], cid[first * b:]
        if args.scheme == "original":
            # in-loop sampling is inside the reference's timing window (train_original.py:42-70):
            # the whole window's negatives are drawn on device first, same stream order.
            n = neg[first * b * k:(first + count) * b * k]
            sampler.sample_batch_device(count * b * k, out=n)
            return eng.epoch_original(u, c, n, count)
        if args.scheme == "neg_shared":
            return eng.epoch_neg_shared(u, c, count)
        return eng.epoch_group_neg_shared(u, c, count)

    def sync():
        torch.cuda.synchronize()
        if dist_on:
            dist.barrier()
            torch.cuda.synchronize()

    if W > 0:
        run(0, W)
    sync()
    t0 = time.perf_counter()
    losses = run(W, K)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    if dist_on:
        dist.barrier()
    elapsed = t1 - t0
    if dist_on:
        t = torch.tensor([elapsed], dtype=torch.float64, device=eng.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    mean_loss = float(losses.double().mean().item())
    if not np.isfinite(mean_loss):
        raise SystemExit("ERROR (nan)! cost is %r" % mean_loss)  # utilities.py:33-35

    # ---- roofline leg: same steps again, dominant kernel bracketed by HIP events ----------
    prof_n = min(K, args.profile_launches)
    eng.profile_begin(prof_n)
    run(W, prof_n)
    raw_ms, launches, ov_ms = eng.profile_end()
    kern_ms = max(raw_ms - ov_ms, 1e-9)   # bracket minus the empty-bracket (marker packet) cost
    n_cols = b
    if args.scheme == "original":
        # HBM-gather bound: 8d(2+k) + 8(1+k) bytes per link; the score kernel
        # carries the read half of it: (2+k) row reads + ids per link.
        bytes_per_launch = b * (4 * d * (2 + k) + 8 * (1 + k))
        achieved = bytes_per_launch / (kern_ms * 1e-3) / 1e9 if kern_ms > 0 else 0.0
        roof = {"kernel": "original_score_kernel", "bound": "hbm", "achieved": achieved, "peak": PEAK_HBM_GBS,
                "unit": "GB/s", "frac": achieved / PEAK_HBM_GBS, "traffic": None,
                "algorithmic_bytes_per_launch": bytes_per_launch}
    else:
        flops_per_launch = 6.0 * b * n_cols * d     # S, G.V, G^T.U: 3 GEMMs of 2*b*N*d (upper bound N = b)
        achieved = flops_per_launch / (kern_ms * 1e-3) / 1e12 if kern_ms > 0 else 0.0
        roof = {"kernel": "shared_tile_kernel", "bound": "mfma", "achieved": achieved, "peak": PEAK_F32_MFMA_TFLOPS,
                "unit": "TFLOP/s", "frac": achieved / PEAK_F32_MFMA_TFLOPS, "traffic": None,
                "algorithmic_flops_per_launch": flops_per_launch}
    roof["kernel_ms"] = kern_ms
    roof["event_bracket_ms"] = raw_ms
    roof["empty_bracket_ms"] = ov_ms
    roof["launches"] = launches
    roof["traffic"] = measured_traffic(args, conf, d)

    cpu = None
    if rank == 0 and args.cpu_seconds > 0:
        cpu = cpu_baseline(args, conf, inp, U0, V0)

    if rank == 0:
        value = K * b * world / elapsed
        line = {
            "metric": "positive-links/sec (train step)", "value": value, "unit": "positive-links/s",
            "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": elapsed / K * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic CiteULike-a-shaped (16K users x 17K items x 200K links), random-init tables",
            "config": {"workload": "CiteULike-a-shaped, id-table towers (model_choice=%s) d=%d, train_scheme=%s, "
                                   "loss=%s, b=%d%s, sparse SGD" % (args.model, d, args.scheme, conf.loss, b,
                                                                   (", k=%d" % k) if k else ""),
                       "global_batch": b * world,
                       "parallelism": "1 process per GPU%s" % ("" if world == 1 else ", independent replicas (no exchange yet)")},
            "mean_loss": mean_loss,
            "roofline": roof,
            "cpu_baseline": cpu,
        }
        emit(line)
    if dist_on:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
