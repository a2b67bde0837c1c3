"""GPU: the multi-rank paths with a REAL second rank -- two processes sharing the one GPU of the box, each running the
product's HIP kernels (HipBackend / HipTowerReplica), collectives over gloo with the payloads staged through host memory.
What crosses between the ranks is what RCCL carries on a multi-GPU node: a row gathered by rank 0's gather kernel lands
in rank 1's step kernel as its item (or user) table, rank 1's exported gradient rows come back into rank 0's
scatter-merge.  70 steps (> one 64-step chunk) against the single-process restatements on the oracle
(tests/sharded_cases.py): loss per step to 1e-5, every shard row.  No reference counterpart (single device,
/root/reference/main.py:34-35); contract = SURVEY.md 8(e).

The rank processes are started with subprocess (fresh interpreters: nothing of this process's GPU state is shared), and
the one-launch step runs in its shared-device form (no in-launch wait: the two ranks' kernels are co-tenants)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from tests import sharded_cases as sc  # noqa: E402
from tests.tolerances import assert_within_sum_bound  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run_ranks(case, world, tmp_path, loss="skip-gram", optimizer="sgd", **env_extra):
    port = str(_free_port())
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="1")
    for k in ("NNCF_SHARD_TRANSPORT", "NNCF_SHARD_NOALIAS", "NNCF_COMM_SELF_VIA_RCCL", "NNCF_TEST_BACKEND", "NNCF_TEST_BIAS", "NNCF_TEST_SKEW"):
        env.pop(k, None)
    env.update(env_extra)
    os.makedirs(str(tmp_path), exist_ok=True)
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "sharded_rank_worker.py"), case, str(r), str(world), port,
                               str(tmp_path), loss, optimizer], cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
             for r in range(world)]
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(o.decode(errors="replace"))
    for r, p in enumerate(procs):
        assert p.returncode == 0, "rank %d failed:\n%s" % (r, outs[r][-3000:])
    return [dict(np.load(os.path.join(str(tmp_path), "rank%d.npz" % r))) for r in range(world)]


@pytest.mark.parametrize("loss,transport", [("skip-gram", "torch"), ("mse", "torch"), ("skip-gram", "callback")])
def test_weak_form_two_ranks_on_one_gpu(tmp_path, loss, transport):
    """item-sharded weak form: every rank its own batch per step, compact variable-split exchanges.  transport = callback: ONE
    native call per 64-step chunk (nncf_shard_run_chunk) issues every exchange -- what runs over RCCL on a multi-GPU node, here
    over the callback communicator (gloo + host staging: RCCL refuses two ranks on one device)"""
    world = 2
    got = _run_ranks("weak", world, tmp_path, loss, NNCF_SHARD_TRANSPORT=transport)
    U, V, losses = sc.simulate_weak(loss, world)
    us, isz = sc.shard_size(sc.NU, world), sc.shard_size(sc.NI, world)
    U0, V0 = sc.tables()
    for r in range(world):
        np.testing.assert_allclose(got[r]["losses"], losses[:, r], rtol=1e-5)
        # rows: fp32 sums of up to b terms per step over STEPS steps (the restatement itself runs in fp32: both sides round)
        assert_within_sum_bound(got[r]["U"], U[r * us:(r + 1) * us], U0[r * us:(r + 1) * us], sc.B, 2 * sc.STEPS, "U shard %d" % r)
        assert_within_sum_bound(got[r]["V"], V[r * isz:(r + 1) * isz], V0[r * isz:(r + 1) * isz], world * sc.B, 2 * sc.STEPS, "V shard %d" % r)
    # the wire carries the rows that are needed, not world * b padded blocks: count what all_to_all_single was handed
    uniq = sum(len(np.unique(c)) for _, c in sc.weak_batches(world)[0])
    rows_bytes = 2 * uniq * 4 * ((sc.D + 3) // 4 * 4)       # rows out (as an owner ~ as a requester) + gradients
    assert got[0]["wire"] < 1.6 * rows_bytes + 4 * uniq + 2 * 2 * 64 * 4 * 2


@pytest.mark.parametrize("transport", ["torch", "callback"])
def test_weak_form_two_ranks_with_unequal_splits_and_an_empty_peer(tmp_path, transport):
    """VERDICT r5 item 7b on the product's kernels: rank 0 requests nothing from rank 1 (an EMPTY split in every exchange of
    every step), rank 1 sends ~nine ids to shard 0 for each of its own and, twice, a single row -- through the HIP plan /
    gather / scatter-merge kernels and (transport = callback) nncf_shard_run_chunk's own variable-split exchange code, the code
    that runs over RCCL on a multi-GPU node.  The one-rank RCCL run of the box can only ever send everything to itself."""
    world = 2
    got = _run_ranks("weak", world, tmp_path, "skip-gram", NNCF_SHARD_TRANSPORT=transport, NNCF_TEST_SKEW="1")
    U, V, losses = sc.simulate_weak("skip-gram", world, skew=True)
    us, isz = sc.shard_size(sc.NU, world), sc.shard_size(sc.NI, world)
    assert all((c < isz).all() for _, c in sc.weak_batches(world, skew=True)[0])      # the premise: rank 0 never leaves its shard
    U0, V0 = sc.tables()
    for r in range(world):
        np.testing.assert_allclose(got[r]["losses"], losses[:, r], rtol=1e-5)
        assert_within_sum_bound(got[r]["U"], U[r * us:(r + 1) * us], U0[r * us:(r + 1) * us], sc.B, 2 * sc.STEPS, "U shard %d" % r)
        assert_within_sum_bound(got[r]["V"], V[r * isz:(r + 1) * isz], V0[r * isz:(r + 1) * isz], world * sc.B, 2 * sc.STEPS, "V shard %d" % r)


@pytest.mark.parametrize("loss,bias,transport", [("skip-gram", "both", "torch"), ("skip-gram", "both", "callback"), ("mse", "item", "callback"),
                                                 ("log-loss", "user", "torch")])
def test_weak_form_two_ranks_with_an_interaction_bias(tmp_path, loss, bias, transport):
    """InteractionDot bias (interaction_dot.py:62-71,96-107) on the WEAK sharded form over two HIP ranks (round 6): the user bias is
    the engine's own vector (local users), the item bias values travel with the rows (gather_bias_kernel), the tile + apply pair
    exports d(loss)/d(cbias) per landing slot (StepArgs::gcb), the values come back next to the gradient rows and
    merge_bias_kernel adds the ranks' values in rank order and steps the owner's shard -- with the skewed batches (an empty peer).
    transport = callback: all of it from nncf_shard_run_chunk."""
    world = 2
    got = _run_ranks("weak", world, tmp_path, loss, NNCF_SHARD_TRANSPORT=transport, NNCF_TEST_BIAS=bias, NNCF_TEST_SKEW="1")
    U, V, losses, ub, cb = sc.simulate_weak(loss, world, skew=True, bias=bias)
    us, isz = sc.shard_size(sc.NU, world), sc.shard_size(sc.NI, world)
    U0, V0 = sc.tables()
    for r in range(world):
        np.testing.assert_allclose(got[r]["losses"], losses[:, r], rtol=1e-5)
        assert_within_sum_bound(got[r]["U"], U[r * us:(r + 1) * us], U0[r * us:(r + 1) * us], sc.B, 2 * sc.STEPS, "U shard %d" % r)
        assert_within_sum_bound(got[r]["V"], V[r * isz:(r + 1) * isz], V0[r * isz:(r + 1) * isz], world * sc.B, 2 * sc.STEPS, "V shard %d" % r)
        for name, want, lo, hi, terms in (("bu", ub, r * us, (r + 1) * us, sc.B), ("bi", cb, r * isz, (r + 1) * isz, world * sc.B)):
            assert (name in got[r]) == (want is not None), name
            if want is not None:
                w = want[lo:hi]
                assert_within_sum_bound(got[r][name], w, np.zeros_like(w), terms, 2 * sc.STEPS, "%s shard %d" % (name, r))
    if ub is not None:
        assert np.abs(ub).max() > 1e-3
    if cb is not None:
        assert np.abs(cb).max() > 1e-3


@pytest.mark.parametrize("loss,transport", [("skip-gram", "torch"), ("mse", "torch"), ("skip-gram", "callback"),
                                            ("log-loss", "torch"), ("max-margin", "torch"), ("log-loss", "callback")])
def test_strong_form_two_ranks_equal_the_single_device_step(tmp_path, loss, transport):
    """same-global-batch form: the b x b step split by rows over two ranks == the one-device step on identical inputs.
    Pairwise losses (log-loss, max-margin: objectives.py:90-98) run the strong form REPLICATED -- every rank fetches all
    2 b rows and runs the whole step, the owners write their rows back, no gradient travels (csrc/rows.hip)."""
    world = 2
    got = _run_ranks("strong", world, tmp_path, loss, NNCF_SHARD_TRANSPORT=transport)
    U, V, losses = sc.simulate_strong(loss)
    U0, V0 = sc.tables()
    us, isz = sc.shard_size(sc.NU, world), sc.shard_size(sc.NI, world)
    for r in range(world):
        np.testing.assert_allclose(got[r]["losses"], losses, rtol=1e-5)
        # against the float64 one-device step: a row's gradient is an fp32 sum of up to b terms, STEPS times
        assert_within_sum_bound(got[r]["U"], U[r * us:(r + 1) * us], U0[r * us:(r + 1) * us], sc.B, sc.STEPS, "U shard %d" % r)
        assert_within_sum_bound(got[r]["V"], V[r * isz:(r + 1) * isz], V0[r * isz:(r + 1) * isz], sc.B, sc.STEPS, "V shard %d" % r)


@pytest.mark.parametrize("loss,bias,transport", [("skip-gram", "both", "torch"), ("skip-gram", "user", "callback"),
                                                 ("log-loss", "both", "callback"), ("mse", "item", "torch")])
def test_strong_form_two_ranks_with_an_interaction_bias(tmp_path, loss, bias, transport):
    """InteractionDot bias (interaction_dot.py:62-71,96-107) on the same-global-batch form over two ranks: the bias values travel
    with the rows of the replicated form (a second all-to-all of 4-byte elements with the same counts; nncf_shard_set_bias),
    both ranks run the whole step incl. the bias update, the owners write rows AND bias values back == the one-device step."""
    world = 2
    got = _run_ranks("strong", world, tmp_path, loss, NNCF_SHARD_TRANSPORT=transport, NNCF_TEST_BIAS=bias)
    U, V, losses, ub, cb = sc.simulate_strong(loss, bias=bias)
    U0, V0 = sc.tables()
    us, isz = sc.shard_size(sc.NU, world), sc.shard_size(sc.NI, world)
    for r in range(world):
        np.testing.assert_allclose(got[r]["losses"], losses, rtol=1e-5)
        assert_within_sum_bound(got[r]["U"], U[r * us:(r + 1) * us], U0[r * us:(r + 1) * us], sc.B, sc.STEPS, "U shard %d" % r)
        assert_within_sum_bound(got[r]["V"], V[r * isz:(r + 1) * isz], V0[r * isz:(r + 1) * isz], sc.B, sc.STEPS, "V shard %d" % r)
        for name, want, lo, hi in (("bu", ub, r * us, (r + 1) * us), ("bi", cb, r * isz, (r + 1) * isz)):
            assert (name in got[r]) == (want is not None)
            if want is not None:    # a bias value: an fp32 sum of up to b score gradients per step
                w = want[lo:hi]
                if loss == "log-loss" and name == "bi":   # neg_shared pairwise: the item bias cancels in S[i, j] - S[j, j]; its
                    assert np.abs(w).max() < 1e-12 and np.abs(got[r][name]).max() < 1e-7   # gradient is 0 up to float64 rounding
                else:
                    assert_within_sum_bound(got[r][name], w, np.zeros_like(w), sc.B, sc.STEPS, "%s shard %d" % (name, r))
    assert (ub is None or np.abs(ub).max() > 1e-3) and (cb is None or loss == "log-loss" or np.abs(cb).max() > 1e-3)


@pytest.mark.parametrize("kind", ["adam", "sparse_adam"])
def test_strong_form_two_ranks_with_adam_equal_the_single_device_step(tmp_path, kind):
    """the same with the owners stepping their shards by Adam (Keras dense / lazy-sparse) on the rank-ordered sums"""
    world = 2
    got = _run_ranks("strong", world, tmp_path, "skip-gram", kind)
    U, V, losses = sc.simulate_strong("skip-gram", kind)
    us, isz = sc.shard_size(sc.NU, world), sc.shard_size(sc.NI, world)
    for r in range(world):
        np.testing.assert_allclose(got[r]["losses"], losses, rtol=1e-5)
        # Adam: a step is lr m / (sqrt(v) + eps) -- a RELATIVE rounding error of the summed gradient turns into an absolute
        # error of the step (up to lr = 0.01 per step where v is tiny), so the term-count bound of the SGD cases does not
        # apply; 5e-4 / 5e-5 over 70 steps of lr 0.01 is 0.7 % of the distance a row can travel
        np.testing.assert_allclose(got[r]["U"], U[r * us:(r + 1) * us], rtol=5e-4, atol=5e-5)
        np.testing.assert_allclose(got[r]["V"], V[r * isz:(r + 1) * isz], rtol=5e-4, atol=5e-5)


@pytest.mark.parametrize("kind,transport", [("adam", "torch"), ("sparse_adam", "callback"), ("adam", "callback")])
def test_replicated_strong_form_two_ranks_with_adam(tmp_path, kind, transport):
    """pairwise loss + Adam over two ranks: rows AND their moment rows to every rank, the whole step everywhere, rows / moments /
    (dense Adam) stamps written back by the owners, idle shard rows take the zero-gradient step == the one-device step"""
    world = 2
    got = _run_ranks("strong", world, tmp_path, "log-loss", kind, NNCF_SHARD_TRANSPORT=transport)
    U, V, losses = sc.simulate_strong("log-loss", kind)
    us, isz = sc.shard_size(sc.NU, world), sc.shard_size(sc.NI, world)
    for r in range(world):
        np.testing.assert_allclose(got[r]["losses"], losses, rtol=1e-5)
        np.testing.assert_allclose(got[r]["U"], U[r * us:(r + 1) * us], rtol=5e-4, atol=5e-5)    # (Adam: see the test above)
        np.testing.assert_allclose(got[r]["V"], V[r * isz:(r + 1) * isz], rtol=5e-4, atol=5e-5)


@pytest.mark.parametrize("transport", ["torch", "callback"])
def test_data_parallel_tower_two_ranks_on_one_gpu(tmp_path, transport):
    """basic_embedding, data parallel: both ranks' HIP towers export their gradients, ONE all-reduce per step, the
    replicas apply the sum -- against the same loop on two oracle replicas (callback: the all-reduce issued through the
    native communicator interface)"""
    import torch
    from oracle import step_oracle as so
    from nncf_amd import synthetic
    from nncf_amd.conf import get_conf
    from nncf_amd.data_utils import get_data
    from nncf_amd.model_framework import get_model
    world = 2
    got = _run_ranks("tower", world, tmp_path, NNCF_SHARD_TRANSPORT=transport)
    data = synthetic.make_citeulike_like(**sc.TW)
    conf = get_conf("synthetic", "best", dict(sc.TW_PD), model_choice="basic_embedding")
    md = get_model(conf, get_data(dict(data), conf, rand_seed=5), "basic_embedding", seed=2)    # the ranks' initial state
    core = md["model"].core
    tw = core.tower
    init = {k: getattr(tw, k).cpu().numpy().copy() for k in ("W", "Wd", "gamma", "beta")}
    U_init = core.engine.U.cpu().numpy().copy()
    oc = so.OracleConf(batch_size_p=64, num_negatives=0, loss="skip-gram", loss_gamma=10, neg_loss_weight=16, u_reg=1e-4,
                       learn_rate=0.05, emb_normalization=conf.emb_normalization, model_choice="basic_embedding")
    reps = []
    for _ in range(world):
        to = so.TowerOracle(data["C"], tw.W.cpu().numpy(), tw.Wd.cpu().numpy(), tw.bd.cpu().numpy(), tw.gamma.cpu().numpy(),
                            tw.beta.cpu().numpy(), 0.05, dtype=np.float64)
        to.export_only = True
        reps.append((so.StepOracle(core.engine.U.cpu().numpy(), None, oc, dtype=np.float64, tower=to), to))
    streams = sc.tower_streams(data["train"], sc.TW["users"], world)
    want = np.zeros((sc.STEPS, world))
    for t in range(sc.STEPS):
        flats = []
        for r, (orc, to) in enumerate(reps):
            bt = streams[r][t * 64:(t + 1) * 64]
            want[t, r] = orc.step_neg_shared(bt[:, 0], bt[:, 1])
            flats.append(to.flat_grads())
        total = np.sum(flats, axis=0)
        for _, to in reps:
            to.apply_flat(total)
    us = sc.shard_size(sc.TW["users"], world)
    for r in range(world):
        np.testing.assert_allclose(got[r]["losses"], want[:, r], rtol=1e-5)
        # against the float64 replicas: a word row sums (rank, item, occurrence) terms -- up to world x 64 items x L tokens --, a
        # Dense / BatchNorm entry world x 64 block rows, a user row its <= 64 positions x 64 columns; STEPS updates each
        terms = {"W": world * 64 * sc.TW["content_len"], "Wd": world * 64, "gamma": world * 64, "beta": world * 64}
        for k in ("W", "Wd", "gamma", "beta"):
            assert_within_sum_bound(got[r][k], getattr(reps[r][1], k), init[k], terms[k], sc.STEPS, k)
            np.testing.assert_array_equal(got[r][k], got[0][k])                      # replicas bit-identical
        own = slice(r * us, (r + 1) * us)
        assert_within_sum_bound(got[r]["U"][own], reps[r][0].U[own], U_init[own], 64 * 64, sc.STEPS, "U")
    del torch


# ---- ONE rank over RCCL: the multi-GPU code paths on the one-GPU box ----------------------------------------------------
RCCL1 = dict(NNCF_TEST_BACKEND="nccl", NNCF_SHARD_NOALIAS="1", NNCF_COMM_SELF_VIA_RCCL="1")


@pytest.mark.parametrize("case", ["weak", "strong", "tower"])
@pytest.mark.parametrize("transport", ["native", "torch"])
def test_one_rank_drives_the_real_rccl_collectives(tmp_path, case, transport):
    """before this round no RCCL collective had executed -- one rank aliased both ends of every exchange.
    With NNCF_SHARD_NOALIAS=1 the session keeps them apart and the ONE rank of an nccl process group runs the real thing:
      native  nncf_shard_run_chunk issues grouped ncclSend / ncclRecv with the variable splits on the step's stream (the rank's own
              share through RCCL too: NNCF_COMM_SELF_VIA_RCCL=1); the data-parallel tower's all-reduce is ncclAllReduce
      torch   torch.distributed.all_to_all_single / all_reduce on the nccl backend over the library's device views
    Result == the aliased run of the same rank, BITWISE (the exchanges only move bytes)."""
    ref = _run_ranks(case, 1, tmp_path / "alias")[0]
    got = _run_ranks(case, 1, tmp_path / "rccl", NNCF_SHARD_TRANSPORT=transport, **RCCL1)[0]
    for k in ref:
        if k != "wire":
            np.testing.assert_array_equal(got[k], ref[k], err_msg=k)
    if case != "tower":
        assert ref["wire"] == 0 and got["wire"] > 0       # nothing travels when the ends alias; the real exchanges carry the rows
