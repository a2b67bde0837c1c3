"""One rank of tests/test_sharded_two_ranks_gpu.py: `python sharded_rank_worker.py CASE RANK WORLD PORT OUTDIR [LOSS [OPTIMIZER]]`.
All ranks share GPU 0 (the one-GPU box), collectives over gloo with the payloads staged through host memory
(nncf_amd/sharded.py: _Exchanger.stage_host); the HIP kernels are the product's."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from tests import sharded_cases as sc  # noqa: E402


def main():
    case, rank, world, port, outdir = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5]
    loss = sys.argv[6] if len(sys.argv) > 6 else "skip-gram"
    optimizer = sys.argv[7] if len(sys.argv) > 7 else "sgd"
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", port
    backend = os.environ.get("NNCF_TEST_BACKEND", "gloo")    # "nccl": RCCL (one rank per GPU: the world-1 tests of the native transport)
    if backend == "nccl":
        torch.cuda.set_device(0)
    dist.init_process_group(backend, rank=rank, world_size=world)
    from nncf_amd.sharded import HipBackend, ShardedGlobalBatch, ShardedNegShared
    out = {}
    if case in ("weak", "strong"):
        U, V = sc.tables()
        us, isz = sc.shard_size(sc.NU, world), sc.shard_size(sc.NI, world)
        ul, il = min(us, sc.NU - rank * us), min(isz, sc.NI - rank * isz)
        bias = os.environ.get("NNCF_TEST_BIAS") or None
        be = HipBackend(sc.engine_kwargs(loss, optimizer, bias), ul, il, 0, world=world, item_shard=isz, item_offset=rank * isz, seed=rank, rank=rank,
                        mode=case, user_shard=us, user_offset=rank * us, shared_device=True)
        be.U[:, :sc.D].copy_(torch.from_numpy(U[rank * us:rank * us + ul]))
        be.V[:, :sc.D].copy_(torch.from_numpy(V[rank * isz:rank * isz + il]))
        losses = torch.zeros(sc.STEPS, dtype=torch.float32, device=be.device)
        if case == "weak":
            tr = ShardedNegShared(be, sc.NU, sc.NI, sc.B, sc.D)
            bt = sc.weak_batches(world, skew=os.environ.get("NNCF_TEST_SKEW") == "1")[rank]
            uid = be.ids(np.concatenate([u for u, _ in bt]) - rank * us)
            cid = be.ids(np.concatenate([c for _, c in bt]))
        else:
            tr = ShardedGlobalBatch(be, sc.NU, sc.NI, sc.B, sc.D)
            bt = sc.global_batches()
            uid = be.ids(np.concatenate([u for u, _ in bt]))
            cid = be.ids(np.concatenate([c for _, c in bt]))
        tr.run_epoch(uid, cid, sc.STEPS, losses)
        be.engine.check()
        want = os.environ.get("NNCF_SHARD_TRANSPORT")
        assert not want or tr.transport == want, (tr.transport, want)
        lc = losses.cpu()
        if case == "strong" and world > 1 and not os.environ.get("NNCF_DBG_NOREDUCE"):
            dist.all_reduce(lc)               # a step's loss = the sum of the ranks' shares
        out = dict(U=be.U[:, :sc.D].cpu().numpy(), V=be.V[:, :sc.D].cpu().numpy(), losses=lc.numpy(), wire=np.int64(tr.wire_bytes))
        for name in ("bu", "bi"):        # this rank's shards of the InteractionDot bias vectors
            if getattr(be, name, None) is not None:
                out[name] = getattr(be, name).cpu().numpy()
    elif case == "tower":
        from nncf_amd import synthetic
        from nncf_amd.conf import get_conf
        from nncf_amd.data_utils import get_data
        from nncf_amd.model_framework import get_model
        from nncf_amd.sharded import DataParallelTower, HipTowerReplica
        data = synthetic.make_citeulike_like(**sc.TW)
        conf = get_conf("synthetic", "best", dict(sc.TW_PD), model_choice="basic_embedding")
        md = get_model(conf, get_data(dict(data), conf, rand_seed=5), "basic_embedding", seed=2)
        md["model_neg_shared"].engine.set_shared_device(True)
        rep = HipTowerReplica(md, "neg_shared")
        dp = DataParallelTower(rep)
        stream = sc.tower_streams(data["train"], sc.TW["users"], world)[rank]
        e = rep.engine
        losses = torch.zeros(sc.STEPS, dtype=torch.float32, device=e.device)
        dp.run_epoch(e.ids(stream[:, 0]), e.ids(stream[:, 1]), sc.STEPS, 64, losses)
        e.check()
        assert dp.reduce == (world > 1 or os.environ.get("NNCF_SHARD_NOALIAS", "0") == "1")
        tw = rep.tower
        out = dict(U=e.U.cpu().numpy(), losses=losses.cpu().numpy(),
                   **{k: getattr(tw, k).cpu().numpy() for k in ("W", "Wd", "bd", "gamma", "beta")})
    else:
        raise SystemExit("unknown case %r" % case)
    np.savez(os.path.join(outdir, "rank%d.npz" % rank), **out)
    if world > 1:
        dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
