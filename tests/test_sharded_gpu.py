"""GPU: HipBackend of the sharded path (export-gradient step, row gather, deterministic
scatter-apply) at world_size 1 against the plain single-GPU step and the oracle."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import torch  # noqa: E402

from oracle import step_oracle as so  # noqa: E402
from nncf_amd.engine import Engine  # noqa: E402
from nncf_amd.sharded import HipBackend, ShardedNegShared  # noqa: E402


@pytest.mark.parametrize("loss", ["skip-gram", "log-loss", "max-margin"])
def test_sharded_world1_matches_plain_engine_and_oracle(loss):
    b, d, nu, ni = 64, 20, 300, 500
    gamma = {"max-margin": 0.1, "log-loss": 10.}.get(loss, 10.)
    kw = dict(user_count=nu, item_count=ni, dim=d, batch_size_p=b, num_negatives=0, loss=loss, loss_gamma=gamma,
              neg_loss_weight=16., u_reg=1e-3, learn_rate=0.1, normalize_user=loss != "skip-gram", normalize_item=False)
    plain = Engine(seed=5, init_scale=0.5, **kw)
    be = HipBackend(kw, nu, ni, 0, world=1, item_shard=ni, item_offset=0, seed=0)
    be.U.copy_(plain.U)
    be.V.copy_(plain.V)
    conf = so.OracleConf(batch_size_p=b, num_negatives=0, loss=loss, loss_gamma=gamma, neg_loss_weight=16., u_reg=1e-3,
                         learn_rate=0.1, model_choice="mf")
    orc = so.StepOracle(plain.U.cpu().numpy(), plain.V.cpu().numpy(), conf, dtype=np.float32)
    tr = ShardedNegShared(be, nu, ni, b, d)
    rs = np.random.RandomState(3)
    for step in range(4):
        uid, cid = rs.randint(0, nu, b), rs.randint(0, 40 if step % 2 else ni, b)  # odd steps: many duplicates
        got = tr.step(be.ids(uid), be.ids(cid)).item()
        ref = plain.step_neg_shared(plain.ids(uid), plain.ids(cid)).item()
        want = orc.step_neg_shared(uid, cid)
        assert abs(got - want) <= 1e-5 * max(1, abs(want))
        assert abs(got - ref) <= 1e-6 * max(1, abs(ref))
        np.testing.assert_allclose(be.U.cpu().numpy(), orc.U, rtol=1e-5, atol=2e-6)
        np.testing.assert_allclose(be.V.cpu().numpy(), orc.V, rtol=1e-5, atol=2e-6)


@pytest.mark.parametrize("loss", ["skip-gram", "log-loss"])
def test_sharded_world1_chunked_epoch_equals_step_by_step_and_oracle(loss):
    """ShardedNegShared.run_epoch over HipBackend's native chunk session (nncf_shard_*: routes, plans and the id
    exchange of 64 steps batched; one rank: exchange buffers aliased) == the per-step protocol == the oracle;
    70 steps = one full chunk and a partial one."""
    b, d, nu, ni, steps = 64, 20, 300, 500, 70
    gamma = {"log-loss": 10.}.get(loss, 10.)
    kw = dict(user_count=nu, item_count=ni, dim=d, batch_size_p=b, num_negatives=0, loss=loss, loss_gamma=gamma,
              neg_loss_weight=16., u_reg=1e-3, learn_rate=0.1, normalize_user=loss != "skip-gram", normalize_item=False)
    be1 = HipBackend(kw, nu, ni, 0, world=1, item_shard=ni, item_offset=0, seed=0)
    be2 = HipBackend(kw, nu, ni, 0, world=1, item_shard=ni, item_offset=0, seed=0)
    assert torch.equal(be1.U, be2.U) and torch.equal(be1.V, be2.V)
    conf = so.OracleConf(batch_size_p=b, num_negatives=0, loss=loss, loss_gamma=gamma, neg_loss_weight=16., u_reg=1e-3,
                         learn_rate=0.1, model_choice="mf")
    orc = so.StepOracle(be1.U.cpu().numpy(), be1.V.cpu().numpy(), conf, dtype=np.float32)
    rs = np.random.RandomState(3)
    uid = rs.randint(0, nu, steps * b)
    cid = np.where(np.arange(steps * b) // b % 2 == 1, rs.randint(0, 40, steps * b), rs.randint(0, ni, steps * b))
    tr1, tr2 = ShardedNegShared(be1, nu, ni, b, d), ShardedNegShared(be2, nu, ni, b, d)
    ua, ca = be1.ids(uid), be1.ids(cid)
    losses = torch.zeros(steps, dtype=torch.float32, device=be1.device)
    tr1.run_epoch(ua, ca, steps, losses)
    l1 = losses.cpu().numpy()
    for t in range(steps):
        l2 = tr2.step(ua[t * b:(t + 1) * b], ca[t * b:(t + 1) * b]).item()
        want = orc.step_neg_shared(uid[t * b:(t + 1) * b], cid[t * b:(t + 1) * b])
        assert abs(l1[t] - l2) <= 1e-6 * max(1, abs(l2)), (t, l1[t], l2)
        assert abs(l1[t] - want) <= 1e-5 * max(1, abs(want)), (t, l1[t], want)
    np.testing.assert_allclose(be1.U.cpu().numpy(), be2.U.cpu().numpy(), rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(be1.V.cpu().numpy(), be2.V.cpu().numpy(), rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(be1.U.cpu().numpy(), orc.U, rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(be1.V.cpu().numpy(), orc.V, rtol=2e-4, atol=2e-5)


def test_sharded_world1_config4_shape():
    """BASELINE configs[3] through the sharded path at world = 1: 10M users x 1M items, d = 128, b = 512, neg_shared;
    3 steps, loss and every touched row of the user shard and of the item shard against the oracle."""
    users, items, b, d = 10000000, 1000000, 512, 128
    kw = dict(user_count=users, item_count=items, dim=d, batch_size_p=b, num_negatives=0, loss="skip-gram", loss_gamma=10.,
              neg_loss_weight=128., u_reg=1e-6, learn_rate=0.05, normalize_user=False, normalize_item=False)
    be = HipBackend(kw, users, items, 0, world=1, item_shard=items, item_offset=0, seed=3)
    assert be.U.shape == (users, d) and be.V.shape == (items, d)
    conf = so.OracleConf(batch_size_p=b, num_negatives=0, loss="skip-gram", u_reg=1e-6, learn_rate=0.05, model_choice="mf")
    rs = np.random.RandomState(5)
    zipf = lambda n, cnt: np.minimum((cnt * rs.power(0.35, n)).astype(np.int64), cnt - 1)  # noqa: E731
    steps = 3
    uid, cid = zipf(steps * b, users), zipf(steps * b, items)
    uid[0], cid[0] = users - 1, items - 1
    us, its = np.unique(uid), np.unique(cid)
    dev = be.device
    orc = so.StepOracle(be.U[torch.as_tensor(us, device=dev)].cpu().numpy(), be.V[torch.as_tensor(its, device=dev)].cpu().numpy(),
                        conf, dtype=np.float32)
    tr = ShardedNegShared(be, users, items, b, d)
    losses = torch.zeros(steps, dtype=torch.float32, device=dev)
    tr.run_epoch(be.ids(uid), be.ids(cid), steps, losses)
    for t in range(steps):
        want = orc.step_neg_shared(np.searchsorted(us, uid[t * b:(t + 1) * b]), np.searchsorted(its, cid[t * b:(t + 1) * b]))
        assert abs(losses[t].item() - want) <= 1e-5 * max(1, abs(want)), (t, losses[t].item(), want)
    np.testing.assert_allclose(be.U[torch.as_tensor(us, device=dev)].cpu().numpy(), orc.U, rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(be.V[torch.as_tensor(its, device=dev)].cpu().numpy(), orc.V, rtol=1e-5, atol=2e-6)


def test_gather_and_scatter_apply_rows():
    import ctypes
    from nncf_amd import _lib
    L = _lib.lib()
    d, n_rows, n = 24, 50, 700
    rs = np.random.RandomState(0)
    table = torch.as_tensor(rs.rand(n_rows, d).astype(np.float32)).cuda()
    ids = rs.randint(-1, n_rows, n).astype(np.int32)
    rows = rs.rand(n, d).astype(np.float32)
    ids_t, rows_t = torch.as_tensor(ids).cuda(), torch.as_tensor(rows).cuda()
    out = torch.empty(n, d, device="cuda")
    p = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
    _lib.check(L.nncf_gather_rows(p(table), d, d, p(ids_t), n, 0, p(out), _lib.stream_ptr()))
    want = np.where(ids[:, None] >= 0, table.cpu().numpy()[np.maximum(ids, 0)], 0)
    assert np.array_equal(out.cpu().numpy(), want)
    ws = torch.empty(int(L.nncf_scatter_apply_ws_ints(n)), dtype=torch.int32, device="cuda")
    t0 = table.cpu().numpy().copy()
    _lib.check(L.nncf_scatter_apply(p(table), d, d, p(ids_t), n, 0, p(rows_t), 0.5, p(ws), _lib.stream_ptr()))
    ref = t0.copy()
    acc = np.zeros_like(ref)
    for i in range(n):  # list order
        if ids[i] >= 0:
            acc[ids[i]] += rows[i]
    ref -= np.float32(0.5) * acc
    np.testing.assert_allclose(table.cpu().numpy(), ref, rtol=1e-6, atol=1e-6)
    # run-to-run deterministic
    table2 = torch.as_tensor(t0).cuda()
    _lib.check(L.nncf_scatter_apply(p(table2), d, d, p(ids_t), n, 0, p(rows_t), 0.5, p(ws), _lib.stream_ptr()))
    assert torch.equal(table, table2)


@pytest.mark.parametrize("ranks,cap,n_rows", [(8, 512, 3000), (3, 100, 40), (8, 8192, 200000), (1, 64, 500)])
def test_scatter_merge_of_sorted_request_blocks(ranks, cap, n_rows):
    """Owner side of a sharded step (nncf_scatter_merge, what nncf_shard_apply runs): `ranks` request blocks, each
    ascending in id with -1 padding at its tail, ids shared between ranks; the rows of an id are added in rank order
    and the table row is stepped once -- against numpy, bitwise run to run, for 8 x 8,192 requests too (no plan)."""
    import ctypes
    from nncf_amd import _lib
    L = _lib.lib()
    d = 24
    rs = np.random.RandomState(ranks * 1000 + cap)
    req = np.full((ranks, cap), -1, dtype=np.int32)
    for r in range(ranks):
        k = min(rs.randint(cap // 2, cap + 1), n_rows)    # few table rows: most ids are shared by several ranks
        req[r, :k] = np.sort(rs.choice(n_rows, k, replace=False))
    rows = rs.rand(ranks * cap, d).astype(np.float32)
    table0 = rs.rand(n_rows, d).astype(np.float32)
    p = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
    req_t, rows_t = torch.as_tensor(req).cuda(), torch.as_tensor(rows).cuda()
    outs = []
    for _ in range(2):
        table = torch.as_tensor(table0).cuda()
        _lib.check(L.nncf_scatter_merge(p(table), d, d, p(req_t), ranks, cap, 0, p(rows_t), 0.5, _lib.stream_ptr()))
        outs.append(table)
    assert torch.equal(outs[0], outs[1])
    acc = np.zeros_like(table0)
    flat = req.reshape(-1)
    for r in range(ranks):        # rank order, float32 like the kernel
        sl = slice(r * cap, (r + 1) * cap)
        ok = flat[sl] >= 0
        acc[flat[sl][ok]] += rows[sl][ok]
    ref = table0 - np.float32(0.5) * acc
    np.testing.assert_allclose(outs[0].cpu().numpy(), ref, rtol=1e-6, atol=1e-6)
    untouched = np.setdiff1d(np.arange(n_rows), flat[flat >= 0])
    assert np.array_equal(outs[0].cpu().numpy()[untouched], table0[untouched])


@pytest.mark.parametrize("model,optimizer", [("basic_embedding", "sgd"), ("cnn_embedding", "sgd"), ("cnn_embedding", "adam"),
                                             ("cnn_bank", "sgd"), ("cnn_bank", "adam")])
def test_data_parallel_replica_world1_equals_plain_step(model, optimizer):
    """HipTowerReplica + DataParallelTower without a process group (gradients exported to the flat
    buffer, then applied by nncf_tower_apply_grads) == the plain in-place step of the text-tower models."""
    from nncf_amd import synthetic
    from nncf_amd.conf import get_conf
    from nncf_amd.data_utils import get_data
    from nncf_amd.model_framework import get_model
    from nncf_amd.sharded import DataParallelTower, HipTowerReplica
    data = synthetic.make_citeulike_like(users=300, items=260, train_links=5000, seed=3, content_len=24, vocab=120)
    pd = {"reset_after_getconf": True, "max_epoch": 1, "loss": "skip-gram", "num_negatives": 0, "neg_loss_weight": 16,
          "loss_gamma": 10, "learn_rate": 0.1, "neg_dist": "unigram", "neg_sampling_power": 1, "batch_size_p": 64,
          "user_dim": 24, "item_dim": 24, "word_dim": 24, "chop_size": 4, "u_reg": 1e-4, "word_emb_dropout_rate": 0.,
          "num_filters": [20], "filter_lengths": [3], "conv_dropout_rate": 0., "optimizer": optimizer}
    if optimizer != "sgd":
        pd["learn_rate"] = 0.01
    if model == "cnn_bank":    # the `default` preset's filter bank: the flat gradient carries one conv block per length
        model = "cnn_embedding"
        pd["filter_lengths"] = [[1, 3, 5]]
    conf = get_conf("synthetic", "best", dict({"optimizer": "sgd"}, **pd), model_choice=model)
    mds = [get_model(conf, get_data(dict(data), conf, rand_seed=5), model, seed=2) for _ in range(2)]
    dp = DataParallelTower(HipTowerReplica(mds[0], "neg_shared"))
    plain = mds[1]["model_neg_shared"]
    e = plain.engine
    rs = np.random.RandomState(1)
    for step in range(4):
        bt = data["train"][rs.choice(len(data["train"]), 64, replace=False)]
        u, c = e.ids(bt[:, 0]), e.ids(bt[:, 1])
        got = float(dp.step(u, c).item())
        want = float(plain.core.step("neg_shared", u, c).item())
        assert abs(got - want) <= 1e-5 * max(1, abs(want)), (step, got, want)
    t0, t1 = dp.be.tower, plain.core.tower
    names = ("W", "Wd", "bd", "gamma", "beta", "rm", "rv")
    if model == "cnn_embedding":
        names += ("Wc", "bc", "g1", "b1", "rm1", "rv1")
    for name in names:
        if optimizer != "sgd" and name in ("bd", "bc"):
            continue   # their gradient is rounding noise (BatchNorm follows), which Adam normalises to +-lr
        np.testing.assert_allclose(getattr(t0, name).cpu().numpy(), getattr(t1, name).cpu().numpy(), rtol=1e-5, atol=1e-6,
                                   err_msg=name)
    np.testing.assert_allclose(dp.be.engine.U.cpu().numpy(), e.U.cpu().numpy(), rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("model", ["basic_embedding", "cnn_embedding"])
def test_data_parallel_replica_with_c_reg_and_a_dense_tail_variant(model):
    """The gradient-export road of round 5's fidelity items: conf.c_reg (the word kernel's tail adds the regulariser's share to
    the exported row, the loss carries the term) and item_dense_transform = tanh + dropout 0.2 (the materialised item block;
    the backward starts from dL/dE * mask * act') through HipTowerReplica + DataParallelTower == the plain in-place step."""
    from nncf_amd import synthetic
    from nncf_amd.conf import get_conf
    from nncf_amd.data_utils import get_data
    from nncf_amd.model_framework import get_model
    from nncf_amd.sharded import DataParallelTower, HipTowerReplica
    data = synthetic.make_citeulike_like(users=300, items=260, train_links=5000, seed=3, content_len=24, vocab=120)
    pd = {"reset_after_getconf": True, "max_epoch": 1, "loss": "skip-gram", "num_negatives": 0, "neg_loss_weight": 16,
          "loss_gamma": 10, "learn_rate": 0.1, "neg_dist": "unigram", "neg_sampling_power": 1, "batch_size_p": 64,
          "user_dim": 24, "item_dim": 24, "word_dim": 24, "chop_size": 4, "u_reg": 1e-4, "word_emb_dropout_rate": 0.3,
          "num_filters": [20], "filter_lengths": [3], "conv_dropout_rate": 0.2, "optimizer": "sgd", "c_reg": 0.05}
    conf = get_conf("synthetic", "best", pd, model_choice=model)
    conf.item_dense_transform = {"dense_hidden_dim": conf.item_dim, "dense_hidden_dropout": 0.2, "dense_hidden_actv": "tanh"}
    mds = [get_model(conf, get_data(dict(data), conf, rand_seed=5), model, seed=2) for _ in range(2)]
    dp = DataParallelTower(HipTowerReplica(mds[0], "neg_shared"))
    plain = mds[1]["model_neg_shared"]
    e = plain.engine
    rs = np.random.RandomState(1)
    for step in range(4):
        bt = data["train"][rs.choice(len(data["train"]), 64, replace=False)]
        u, c = e.ids(bt[:, 0]), e.ids(bt[:, 1])
        got = float(dp.step(u, c).item())
        want = float(plain.core.step("neg_shared", u, c).item())
        assert abs(got - want) <= 1e-5 * max(1, abs(want)), (step, got, want)
    t0, t1 = dp.be.tower, plain.core.tower
    names = ("W", "Wd", "bd", "gamma", "beta", "rm", "rv")
    if model == "cnn_embedding":
        names += ("Wc", "bc", "g1", "b1", "rm1", "rv1")
    for name in names:
        np.testing.assert_allclose(getattr(t0, name).cpu().numpy(), getattr(t1, name).cpu().numpy(), rtol=1e-5, atol=1e-6,
                                   err_msg=name)
    np.testing.assert_allclose(dp.be.engine.U.cpu().numpy(), e.U.cpu().numpy(), rtol=1e-6, atol=1e-7)


def test_gradient_buffer_alignment_is_checked():
    """nncf_tower_set_grad_export / nncf_tower_apply_grads read and write the flat gradient four floats at a time: a pointer off
    16 bytes is refused (NNCF_ERR_INVALID -> ValueError), not dereferenced."""
    import ctypes
    from nncf_amd import _lib, synthetic
    from nncf_amd.conf import get_conf
    from nncf_amd.data_utils import get_data
    from nncf_amd.model_framework import get_model
    data = synthetic.make_citeulike_like(users=100, items=80, train_links=1000, seed=3, content_len=12, vocab=40)
    pd = {"reset_after_getconf": True, "max_epoch": 1, "loss": "skip-gram", "num_negatives": 0, "batch_size_p": 32, "user_dim": 8,
          "item_dim": 8, "word_dim": 8, "optimizer": "sgd"}
    conf = get_conf("synthetic", "best", pd, model_choice="basic_embedding")
    tw = get_model(conf, get_data(dict(data), conf, rand_seed=5), "basic_embedding", seed=2)["model"].core.tower
    L = _lib.lib()
    n = int(L.nncf_tower_grad_floats(tw.handle))
    buf = torch.zeros(n + 8, device="cuda")
    with torch.cuda.device(buf.device):
        with pytest.raises(ValueError, match="16-byte"):
            _lib.check(L.nncf_tower_set_grad_export(tw.handle, ctypes.c_void_p(buf.data_ptr() + 4)))
        with pytest.raises(ValueError, match="16-byte"):
            _lib.check(L.nncf_tower_apply_grads(tw.handle, ctypes.c_void_p(buf.data_ptr() + 8), 1.0, 0, None))
        _lib.check(L.nncf_tower_apply_grads(tw.handle, ctypes.c_void_p(buf.data_ptr()), 1.0, 0, None))   # aligned: a zero gradient
    torch.cuda.synchronize()


def test_data_parallel_run_epoch_equals_plain_epoch():
    """DataParallelTower.run_epoch (plans of 64 steps per launch, planned step -> apply) over 70 steps ==
    the single-GPU epoch runner, bitwise on the user table and to rounding on the tower (the gradient
    takes the export buffer's detour)."""
    from nncf_amd import synthetic
    from nncf_amd.conf import get_conf
    from nncf_amd.data_utils import get_data
    from nncf_amd.model_framework import get_model
    from nncf_amd.sharded import DataParallelTower, HipTowerReplica
    data = synthetic.make_citeulike_like(users=300, items=260, train_links=5000, seed=3, content_len=24, vocab=120)
    pd = {"reset_after_getconf": True, "max_epoch": 1, "loss": "skip-gram", "num_negatives": 0, "neg_loss_weight": 16,
          "loss_gamma": 10, "learn_rate": 0.05, "neg_dist": "unigram", "neg_sampling_power": 1, "batch_size_p": 64,
          "user_dim": 24, "item_dim": 24, "word_dim": 24, "chop_size": 4, "u_reg": 1e-4, "word_emb_dropout_rate": 0.}
    conf = get_conf("synthetic", "best", dict({"optimizer": "sgd"}, **pd), model_choice="basic_embedding")
    mds = [get_model(conf, get_data(dict(data), conf, rand_seed=5), "basic_embedding", seed=2) for _ in range(2)]
    dp = DataParallelTower(HipTowerReplica(mds[0], "neg_shared"))
    plain = mds[1]["model_neg_shared"]
    e = plain.engine
    rs = np.random.RandomState(2)
    steps, b = 70, 64
    bt = data["train"][rs.choice(len(data["train"]), steps * b, replace=True)]
    u, c = e.ids(bt[:, 0]), e.ids(bt[:, 1])
    losses = torch.zeros(steps, device="cuda")
    dp.run_epoch(u, c, steps, b, losses)
    want = plain.run_epoch(u, c, steps)
    np.testing.assert_allclose(losses.cpu().numpy(), want.cpu().numpy(), rtol=2e-5, atol=1e-6)
    for name in ("W", "Wd", "gamma", "beta"):
        np.testing.assert_allclose(getattr(dp.be.tower, name).cpu().numpy(), getattr(plain.core.tower, name).cpu().numpy(),
                                   rtol=1e-4, atol=2e-6, err_msg=name)
    np.testing.assert_allclose(dp.be.engine.U.cpu().numpy(), e.U.cpu().numpy(), rtol=1e-5, atol=1e-6)


def test_data_parallel_native_runner_equals_the_python_loop_bitwise(monkeypatch):
    """nncf_tower_epoch_dp (round 5: planned step -> all-reduce -> apply for every step in ONE native call) against
    DataParallelTower's Python loop of the same three library calls per step (NNCF_DP_NATIVE=0): losses, user table and every
    tower parameter bit for bit, over 70 steps (two plan chunks) of neg_shared and group_neg_shared."""
    from nncf_amd import synthetic
    from nncf_amd.conf import get_conf
    from nncf_amd.data_utils import get_data
    from nncf_amd.model_framework import get_model
    from nncf_amd.sharded import DataParallelTower, HipTowerReplica
    data = synthetic.make_citeulike_like(users=300, items=260, train_links=5000, seed=3, content_len=24, vocab=120)
    pd = {"reset_after_getconf": True, "max_epoch": 1, "loss": "skip-gram", "num_negatives": 0, "neg_loss_weight": 16,
          "loss_gamma": 10, "learn_rate": 0.05, "neg_dist": "unigram", "neg_sampling_power": 1, "batch_size_p": 64,
          "user_dim": 24, "item_dim": 24, "word_dim": 24, "chop_size": 4, "u_reg": 1e-4, "word_emb_dropout_rate": 0.3}
    conf = get_conf("synthetic", "best", dict({"optimizer": "sgd"}, **pd), model_choice="basic_embedding")
    rs = np.random.RandomState(2)
    steps, b = 70, 64
    bt = data["train"][rs.choice(len(data["train"]), steps * b, replace=True)]
    for scheme in ("neg_shared", "group_neg_shared"):
        cid_np = np.repeat(bt[::4, 1], 4)[:steps * b] if scheme == "group_neg_shared" else bt[:, 1]
        out = []
        for native in ("1", "0"):
            monkeypatch.setenv("NNCF_DP_NATIVE", native)
            md = get_model(conf, get_data(dict(data), conf, rand_seed=5), "basic_embedding", seed=2)
            dp = DataParallelTower(HipTowerReplica(md, scheme))
            e = dp.be.engine
            losses = torch.zeros(steps, device="cuda")
            dp.run_epoch(e.ids(bt[:, 0]), e.ids(cid_np), steps, b, losses)
            torch.cuda.synchronize()
            tw = dp.be.tower
            out.append([losses.clone(), e.U.clone()] + [getattr(tw, n).clone() for n in ("W", "Wd", "bd", "gamma", "beta")])
        for x, y in zip(*out):
            assert torch.equal(x, y), scheme


def test_sharded_large_batch_world1():
    """The large-batch regime of config 4 through the sharded path at world 1: b = 4,096 links per step -- the
    4,096-entry request plan, the wide one-launch step exporting the item gradients, gather / scatter of 4,096 landing
    rows -- three chunked steps == the serial oracle; rows nobody touched stay bit-identical."""
    users, items, b, d, steps = 6000, 5000, 4096, 32, 3
    kw = dict(user_count=users, item_count=items, dim=d, batch_size_p=b, num_negatives=0, loss="skip-gram", loss_gamma=10.,
              neg_loss_weight=128., u_reg=1e-6, learn_rate=0.05, normalize_user=False, normalize_item=False)
    be = HipBackend(kw, users, items, 0, world=1, item_shard=items, item_offset=0, seed=3)
    U0, V0 = be.U.cpu().numpy().copy(), be.V.cpu().numpy().copy()
    conf = so.OracleConf(batch_size_p=b, num_negatives=0, loss="skip-gram", u_reg=1e-6, learn_rate=0.05, model_choice="mf")
    orc = so.StepOracle(U0, V0, conf, dtype=np.float64)
    rs = np.random.RandomState(3)
    uid = rs.randint(0, users, steps * b)
    cid = np.minimum((items * rs.power(0.5, steps * b)).astype(np.int64), items - 1)
    tr = ShardedNegShared(be, users, items, b, d)
    losses = torch.zeros(steps, dtype=torch.float32, device=be.device)
    tr.run_epoch(be.ids(uid), be.ids(cid), steps, losses)
    for t in range(steps):
        want = orc.step_neg_shared(uid[t * b:(t + 1) * b], cid[t * b:(t + 1) * b])
        assert abs(losses[t].item() - want) <= 1e-5 * max(1, abs(want)), (t, losses[t].item(), want)
    U1, V1 = be.U.cpu().numpy(), be.V.cpu().numpy()
    np.testing.assert_allclose(U1, orc.U, rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(V1, orc.V, rtol=1e-5, atol=1e-5)
    idle_u, idle_v = np.setdiff1d(np.arange(users), uid), np.setdiff1d(np.arange(items), cid)
    assert np.array_equal(U1[idle_u], U0[idle_u]) and np.array_equal(V1[idle_v], V0[idle_v])


@pytest.mark.parametrize("loss,norm", [("skip-gram", False), ("mse", False), ("skip-gram", True), ("log-loss", False),
                                       ("max-margin", False), ("log-loss", True)])
def test_strong_form_world1_equals_plain_step_and_oracle(loss, norm):
    """ShardedGlobalBatch at world 1: both engine tables are landing buffers, the rectangular form of the one-launch step
    (all rows) exports both gradients, the owner-side merge steps the shards -- 70 steps (a full chunk + a partial one)
    == the oracle's one-device step on the same batches; duplicates on both sides."""
    from nncf_amd.sharded import ShardedGlobalBatch
    b, d, nu, ni, steps = 64, 20, 300, 500, 70
    kw = dict(user_count=nu, item_count=ni, dim=d, batch_size_p=b, num_negatives=0, loss=loss, loss_gamma=10.,
              neg_loss_weight=16., u_reg=1e-3, learn_rate=0.1, normalize_user=norm, normalize_item=False)
    be = HipBackend(kw, nu, ni, 0, world=1, item_shard=ni, item_offset=0, seed=0, mode="strong", user_shard=nu)
    assert be.engine.U.shape[0] == b and be.engine.V.shape[0] == b and be.U.shape[0] == nu and be.V.shape[0] == ni
    assert be.replicated == (loss in ("log-loss", "max-margin"))   # pairwise losses: the whole step on every rank, owners write back
    conf = so.OracleConf(batch_size_p=b, num_negatives=0, loss=loss, loss_gamma=10., neg_loss_weight=16., u_reg=1e-3,
                         learn_rate=0.1, model_choice="mf", emb_normalization=norm)
    orc = so.StepOracle(be.U.cpu().numpy(), be.V.cpu().numpy(), conf, dtype=np.float64)
    rs = np.random.RandomState(3)
    uid = rs.randint(0, nu, steps * b)
    uid[:6] = uid[0]
    cid = np.where(np.arange(steps * b) // b % 2 == 1, rs.randint(0, 40, steps * b), rs.randint(0, ni, steps * b))
    tr = ShardedGlobalBatch(be, nu, ni, b, d)
    losses = torch.zeros(steps, dtype=torch.float32, device=be.device)
    tr.run_epoch(be.ids(uid), be.ids(cid), steps, losses)
    be.engine.check()
    l = losses.cpu().numpy()
    for t in range(steps):
        want = orc.step_neg_shared(uid[t * b:(t + 1) * b], cid[t * b:(t + 1) * b])
        assert abs(l[t] - want) <= 1e-5 * max(1, abs(want)), (t, l[t], want)
    np.testing.assert_allclose(be.U.cpu().numpy(), orc.U, rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(be.V.cpu().numpy(), orc.V, rtol=1e-4, atol=2e-5)


@pytest.mark.parametrize("bias", ["user", "item", "both"])
@pytest.mark.parametrize("loss", ["skip-gram", "log-loss"])
def test_strong_form_with_an_interaction_bias_equals_the_oracle(loss, bias):
    """InteractionDot bias (interaction_dot.py:62-71,96-107) on ShardedGlobalBatch.  A model with a bias runs
    the REPLICATED form for every loss (the rectangular one-launch step has no bias form): the bias values travel with the rows
    (nncf_shard_set_bias), the one-device tile + apply + bias-update kernels step rows and bias on the landing buffers, the
    owners write both back -- 70 steps == the oracle's one-device step with the same bias, tables and bias vectors."""
    from nncf_amd.sharded import ShardedGlobalBatch
    b, d, nu, ni, steps = 64, 20, 300, 500, 70
    kw = dict(user_count=nu, item_count=ni, dim=d, batch_size_p=b, num_negatives=0, loss=loss, loss_gamma=10.,
              neg_loss_weight=16., u_reg=1e-3, learn_rate=0.1, normalize_user=False, normalize_item=False, interaction_bias=bias)
    be = HipBackend(kw, nu, ni, 0, world=1, item_shard=ni, item_offset=0, seed=0, mode="strong", user_shard=nu)
    assert be.replicated and (be.bu is not None) == (bias != "item") and (be.bi is not None) == (bias != "user")
    conf = so.OracleConf(batch_size_p=b, num_negatives=0, loss=loss, loss_gamma=10., neg_loss_weight=16., u_reg=1e-3,
                         learn_rate=0.1, model_choice="mf", emb_normalization=False, interaction_bias=bias)
    orc = so.StepOracle(be.U.cpu().numpy(), be.V.cpu().numpy(), conf, dtype=np.float64)
    rs = np.random.RandomState(3)
    uid = rs.randint(0, nu, steps * b)
    uid[:6] = uid[0]
    cid = np.where(np.arange(steps * b) // b % 2 == 1, rs.randint(0, 40, steps * b), rs.randint(0, ni, steps * b))
    tr = ShardedGlobalBatch(be, nu, ni, b, d)
    losses = torch.zeros(steps, dtype=torch.float32, device=be.device)
    tr.run_epoch(be.ids(uid), be.ids(cid), steps, losses)
    be.engine.check()
    l = losses.cpu().numpy()
    for t in range(steps):
        want = orc.step_neg_shared(uid[t * b:(t + 1) * b], cid[t * b:(t + 1) * b])
        assert abs(l[t] - want) <= 1e-5 * max(1, abs(want)), (t, l[t], want)
    np.testing.assert_allclose(be.U.cpu().numpy(), orc.U, rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(be.V.cpu().numpy(), orc.V, rtol=1e-4, atol=2e-5)
    if be.bu is not None:
        assert np.abs(orc.ub).max() > 1e-3
        np.testing.assert_allclose(be.bu.cpu().numpy(), orc.ub, rtol=1e-4, atol=2e-5)
    if be.bi is not None:
        assert np.abs(orc.cb).max() > 1e-3 or loss == "log-loss"   # (neg_shared pairwise: the item bias cancels, its gradient is 0)
        np.testing.assert_allclose(be.bi.cpu().numpy(), orc.cb, rtol=1e-4, atol=2e-5)


@pytest.mark.parametrize("bias", ["user", "item", "both"])
@pytest.mark.parametrize("loss", ["skip-gram", "log-loss"])
def test_weak_form_with_an_interaction_bias_equals_the_oracle(loss, bias):
    """InteractionDot bias (interaction_dot.py:62-71,96-107) on the WEAK sharded form (ShardedNegShared), world 1:
    the user bias is the engine's own vector, the item bias values are gathered with the rows into the landing vector, the
    tile + apply + bias-update kernels export d(loss)/d(cbias) per landing slot (StepArgs::gcb) and merge_bias_kernel steps the
    owner's shard -- 70 steps == the oracle's one-device step with the same bias."""
    b, d, nu, ni, steps = 64, 20, 300, 500, 70
    kw = dict(user_count=nu, item_count=ni, dim=d, batch_size_p=b, num_negatives=0, loss=loss, loss_gamma=10.,
              neg_loss_weight=16., u_reg=1e-3, learn_rate=0.1, normalize_user=False, normalize_item=False, interaction_bias=bias)
    be = HipBackend(kw, nu, ni, 0, world=1, item_shard=ni, item_offset=0, seed=0)
    assert (be.bu is not None) == (bias != "item") and (be.bi is not None) == (bias != "user")
    conf = so.OracleConf(batch_size_p=b, num_negatives=0, loss=loss, loss_gamma=10., neg_loss_weight=16., u_reg=1e-3,
                         learn_rate=0.1, model_choice="mf", emb_normalization=False, interaction_bias=bias)
    orc = so.StepOracle(be.U.cpu().numpy(), be.V.cpu().numpy(), conf, dtype=np.float64)
    rs = np.random.RandomState(3)
    uid = rs.randint(0, nu, steps * b)
    uid[:6] = uid[0]
    cid = np.where(np.arange(steps * b) // b % 2 == 1, rs.randint(0, 40, steps * b), rs.randint(0, ni, steps * b))
    tr = ShardedNegShared(be, nu, ni, b, d)
    losses = torch.zeros(steps, dtype=torch.float32, device=be.device)
    tr.run_epoch(be.ids(uid), be.ids(cid), steps, losses)
    be.engine.check()
    l = losses.cpu().numpy()
    for t in range(steps):
        want = orc.step_neg_shared(uid[t * b:(t + 1) * b], cid[t * b:(t + 1) * b])
        assert abs(l[t] - want) <= 1e-5 * max(1, abs(want)), (t, l[t], want)
    np.testing.assert_allclose(be.U.cpu().numpy(), orc.U, rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(be.V.cpu().numpy(), orc.V, rtol=1e-4, atol=2e-5)
    if be.bu is not None:
        assert np.abs(orc.ub).max() > 1e-3
        np.testing.assert_allclose(be.bu.cpu().numpy(), orc.ub, rtol=1e-4, atol=2e-5)
    if be.bi is not None:
        assert np.abs(orc.cb).max() > 1e-3 or loss == "log-loss"   # (neg_shared pairwise: the item bias cancels, its gradient is 0)
        np.testing.assert_allclose(be.bi.cpu().numpy(), orc.cb, rtol=1e-4, atol=2e-5)


def test_the_per_step_protocol_refuses_an_interaction_bias():
    """ShardedNegShared.step() (static splits, nncf_step_neg_shared_export) would step the LANDING bias vector in place and the
    owners would never see it: refused loudly; run_epoch (the chunked protocol) is the bias-carrying road."""
    kw = dict(user_count=300, item_count=500, dim=20, batch_size_p=64, num_negatives=0, loss="skip-gram", loss_gamma=10.,
              neg_loss_weight=16., u_reg=1e-3, learn_rate=0.1, normalize_user=False, normalize_item=False, interaction_bias="item")
    be = HipBackend(kw, 300, 500, 0, world=1, item_shard=500, item_offset=0, seed=0)
    tr = ShardedNegShared(be, 300, 500, 64, 20)
    rs = np.random.RandomState(0)
    with pytest.raises(NotImplementedError, match="chunked protocol"):
        tr.step(be.ids(rs.randint(0, 300, 64)), be.ids(rs.randint(0, 500, 64)))


def test_the_weak_form_with_a_bias_refuses_adam():
    kw = dict(user_count=300, item_count=500, dim=20, batch_size_p=64, num_negatives=0, loss="skip-gram", loss_gamma=10.,
              neg_loss_weight=16., u_reg=1e-3, learn_rate=0.1, normalize_user=False, normalize_item=False, interaction_bias="both",
              optimizer="adam")
    with pytest.raises(NotImplementedError, match="sparse SGD"):
        HipBackend(kw, 300, 500, 0, world=1, item_shard=500, item_offset=0, seed=0)


def test_strong_form_refuses_the_tile_path_loudly(monkeypatch):
    """with NNCF_SHARED_PATH=tile the one-launch step is off; the strong form's rectangular step (bc, exported user
    gradient) has no tile + apply form and must fail loudly instead of running a square step over the first b / G columns."""
    from nncf_amd.sharded import ShardedGlobalBatch
    monkeypatch.setenv("NNCF_SHARED_PATH", "tile")
    b, d, nu, ni = 64, 20, 300, 500
    kw = dict(user_count=nu, item_count=ni, dim=d, batch_size_p=b, num_negatives=0, loss="skip-gram", loss_gamma=10.,
              neg_loss_weight=16., u_reg=1e-3, learn_rate=0.1, normalize_user=False, normalize_item=False)
    be = HipBackend(kw, nu, ni, 0, world=1, item_shard=ni, item_offset=0, seed=0, mode="strong", user_shard=nu)
    rs = np.random.RandomState(3)
    uid, cid = rs.randint(0, nu, 2 * b), rs.randint(0, ni, 2 * b)
    tr = ShardedGlobalBatch(be, nu, ni, b, d)
    losses = torch.zeros(2, dtype=torch.float32, device=be.device)
    with pytest.raises(Exception, match="one-launch path"):
        tr.run_epoch(be.ids(uid), be.ids(cid), 2, losses)


def test_rectangular_step_shares_add_up_to_the_square_step():
    """The rectangular form of the one-launch step (StepArgs::bc; rows g, g + G, ... of the b x b matrix) run for
    g = 0 .. G-1 on the SAME landed rows: the loss shares and the exported gradient rows add up to the square step's
    (nncf_shard_step at world 1 in the strong form == all rows).  b = 512, d = 100, G = 4: split strips on both sides,
    user strips of 128 rows against 512 columns."""
    import ctypes
    from nncf_amd import _lib
    from nncf_amd.sharded import ShardedGlobalBatch
    b, d, nu, ni, G = 512, 100, 700, 600, 4
    kw = dict(user_count=nu, item_count=ni, dim=d, batch_size_p=b, num_negatives=0, loss="skip-gram", loss_gamma=10.,
              neg_loss_weight=128., u_reg=1e-3, learn_rate=0.05, normalize_user=True, normalize_item=False)
    rs = np.random.RandomState(5)
    uid = rs.randint(0, nu, b)
    uid[:9] = uid[0]
    cid = np.minimum((ni * rs.power(0.4, b)).astype(np.int64), ni - 1)
    U0 = rs.uniform(-.5, .5, (nu, d)).astype(np.float32)
    V0 = rs.uniform(-.5, .5, (ni, d)).astype(np.float32)
    conf = so.OracleConf(batch_size_p=b, num_negatives=0, loss="skip-gram", neg_loss_weight=128., u_reg=1e-3, learn_rate=0.05,
                         model_choice="mf", emb_normalization=True)
    want_loss, gu_want, gv_want = 0.0, np.zeros((nu, d)), np.zeros((ni, d))
    for g in range(G):
        rows = g + G * np.arange(b // G)
        l, gu, gv = so.neg_shared_row_block(U0[uid[rows]], V0[cid], rows, conf, dtype=np.float64)
        want_loss += l
        np.add.at(gu_want, uid[rows], gu)
        np.add.at(gv_want, cid, gv)
    orc = so.StepOracle(U0, V0, conf, dtype=np.float64)
    sq = orc.step_neg_shared(uid, cid)
    assert abs(want_loss - sq) < 1e-9 * abs(sq)          # the oracle's row blocks add up to its square step
    # each "rank" g as its own world-G backend on this one GPU; the exchanges are emulated by hand from the global tables
    gu_got, gv_got = np.zeros((nu, d)), np.zeros((ni, d))   # (the loss shares are checked by the two-rank tests)
    us, isz = (nu + G - 1) // G, (ni + G - 1) // G
    p = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
    L = _lib.lib()
    for g in range(G):
        be = HipBackend(kw, min(us, nu - g * us), min(isz, ni - g * isz), 0, world=G, item_shard=isz, item_offset=g * isz, seed=0,
                        rank=g, mode="strong", user_shard=us, user_offset=g * us, shared_device=True)
        mine = be.ids(uid[g::G])          # (no process group here: ShardedGlobalBatch would take rank 0 of 1)
        bufs = be.chunk_buffers()
        be.plan_chunk(mine, be.ids(cid), 1)
        torch.cuda.synchronize()
        uq_v, uq_u = np.unique(cid), np.unique(uid[g::G])
        xv, xu = bufs["x"]
        xv["rows_in"][:len(uq_v), :d] = torch.from_numpy(V0[uq_v]).cuda()       # what the owners would have sent
        xu["rows_in"][:len(uq_u), :d] = torch.from_numpy(U0[uq_u]).cuda()
        loss = torch.zeros(1, device="cuda")
        _lib.check(L.nncf_shard_step(be._session(), 0, None, p(loss), _lib.stream_ptr()))
        _lib.check(L.nncf_model_check(be.engine._h, _lib.stream_ptr()))   # synchronises; no in-launch wait timed out
        gu_got[uq_u] += xu["g_out"][:len(uq_u), :d].cpu().numpy()
        gv_got[uq_v] += xv["g_out"][:len(uq_v), :d].cpu().numpy()
        be.close()
    scale = max(np.abs(gu_want).max(), np.abs(gv_want).max())
    np.testing.assert_allclose(gu_got, gu_want, rtol=0, atol=2e-5 * scale)
    np.testing.assert_allclose(gv_got, gv_want, rtol=0, atol=2e-5 * scale)


@pytest.mark.parametrize("loss", ["log-loss", "max-margin"])
@pytest.mark.parametrize("kind", ["adam", "sparse_adam"])
def test_replicated_strong_form_with_adam_equals_the_oracle(loss, kind):
    """Adam on the REPLICATED strong form (pairwise losses): the moment rows travel with the table rows into the model's own
    moment arrays, the one-device step updates rows and moments, the owners write both back; Keras' dense Adam: the touched rows'
    stamps come home too and the rest of every shard takes the zero-gradient step -- 70 steps at world 1 == the oracle."""
    from nncf_amd.sharded import ShardedGlobalBatch
    b, d, nu, ni, steps = 64, 20, 300, 500, 70
    kw = dict(user_count=nu, item_count=ni, dim=d, batch_size_p=b, num_negatives=0, loss=loss, loss_gamma=10.,
              neg_loss_weight=16., u_reg=1e-3, learn_rate=0.01, normalize_user=False, normalize_item=False, optimizer=kind)
    be = HipBackend(kw, nu, ni, 0, world=1, item_shard=ni, item_offset=0, seed=0, mode="strong", user_shard=nu)
    assert be.replicated
    conf = so.OracleConf(batch_size_p=b, num_negatives=0, loss=loss, loss_gamma=10., neg_loss_weight=16., u_reg=1e-3,
                         learn_rate=0.01, model_choice="mf", emb_normalization=False, optimizer=kind)
    orc = so.StepOracle(be.U.cpu().numpy(), be.V.cpu().numpy(), conf, dtype=np.float64)
    rs = np.random.RandomState(3)
    uid = rs.randint(0, nu, steps * b)
    uid[:6] = uid[0]
    cid = np.where(np.arange(steps * b) // b % 2 == 1, rs.randint(0, 40, steps * b), rs.randint(0, ni, steps * b))
    tr = ShardedGlobalBatch(be, nu, ni, b, d)
    losses = torch.zeros(steps, dtype=torch.float32, device=be.device)
    tr.run_epoch(be.ids(uid), be.ids(cid), steps, losses)
    be.engine.check()
    l = losses.cpu().numpy()
    for t in range(steps):
        want = orc.step_neg_shared(uid[t * b:(t + 1) * b], cid[t * b:(t + 1) * b])
        assert abs(l[t] - want) <= 1e-5 * max(1, abs(want)), (t, l[t], want)
    for got, want in ((be.U, orc.U), (be.V, orc.V)):   # (Adam: flat tolerance, see the two-rank test)
        np.testing.assert_allclose(got.cpu().numpy()[:, :d], want, rtol=5e-4, atol=5e-5)
    m_u, v_u, _ = be._adam[1]
    np.testing.assert_allclose(m_u.cpu().numpy()[:, :d], orc.slot_u.m, rtol=2e-3, atol=1e-6)   # the moments went home as well
    np.testing.assert_allclose(v_u.cpu().numpy()[:, :d], orc.slot_u.v, rtol=2e-3, atol=1e-9)


@pytest.mark.parametrize("form", ["weak", "strong"])
@pytest.mark.parametrize("kind", ["adam", "sparse_adam"])
def test_sharded_world1_adam_equals_the_plain_engine_and_the_oracle(form, kind):
    """Adam on the sharded paths (nncf_shard_set_optimizer): the owners step their shards with the model's Adam on the summed
    gradient rows -- Keras (dense) Adam incl. the zero-gradient step of every untouched shard row, and the lazy-sparse form;
    70 steps of both forms at world 1 == the plain engine's epoch with the same optimiser == the oracle."""
    from nncf_amd.sharded import ShardedGlobalBatch
    b, d, nu, ni, steps = 64, 20, 300, 500, 70
    kw = dict(user_count=nu, item_count=ni, dim=d, batch_size_p=b, num_negatives=0, loss="skip-gram", loss_gamma=10.,
              neg_loss_weight=16., u_reg=1e-3, learn_rate=0.01, normalize_user=False, normalize_item=False)
    plain = Engine(seed=5, init_scale=0.5, **kw)
    plain.set_optimizer(kind)
    be = HipBackend(dict(kw, optimizer=kind), nu, ni, 0, world=1, item_shard=ni, item_offset=0, seed=0, mode=form, user_shard=nu)
    be.U.copy_(plain.U)
    be.V.copy_(plain.V)
    conf = so.OracleConf(batch_size_p=b, num_negatives=0, loss="skip-gram", loss_gamma=10., neg_loss_weight=16., u_reg=1e-3,
                         learn_rate=0.01, model_choice="mf", optimizer=kind)
    orc = so.StepOracle(plain.U.cpu().numpy(), plain.V.cpu().numpy(), conf, dtype=np.float64)
    rs = np.random.RandomState(3)
    uid = rs.randint(0, nu, steps * b)
    uid[:6] = uid[0]
    cid = np.where(np.arange(steps * b) // b % 2 == 1, rs.randint(0, 40, steps * b), rs.randint(0, ni, steps * b))
    tr = (ShardedGlobalBatch if form == "strong" else ShardedNegShared)(be, nu, ni, b, d)
    losses = torch.zeros(steps, dtype=torch.float32, device=be.device)
    tr.run_epoch(be.ids(uid), be.ids(cid), steps, losses)
    be.engine.check()
    ref = plain.epoch_neg_shared(plain.ids(uid), plain.ids(cid), steps).cpu().numpy()
    l = losses.cpu().numpy()
    for t in range(steps):
        want = orc.step_neg_shared(uid[t * b:(t + 1) * b], cid[t * b:(t + 1) * b])
        assert abs(l[t] - want) <= 1e-5 * max(1, abs(want)), (t, l[t], want)
        assert abs(l[t] - ref[t]) <= 1e-5 * max(1, abs(ref[t])), (t, l[t], ref[t])
    for got, want in ((be.U, orc.U), (be.V, orc.V), (plain.U, orc.U), (plain.V, orc.V)):
        np.testing.assert_allclose(got.cpu().numpy()[:, :d], want, rtol=5e-4, atol=5e-5)
    if form == "weak":   # the per-step protocol has no Adam state for the owners: refused, not silently stepped with SGD
        with pytest.raises(NotImplementedError):
            tr.step(be.ids(uid[:b]), be.ids(cid[:b]))


@pytest.mark.timeout(300)
@pytest.mark.parametrize("seed", range(int(os.environ.get("NNCF_SOAK_SEEDS", "12"))))
def test_sharded_world1_random_shapes_match_the_oracle(seed):
    """Random shapes through both sharded forms at world 1 (ShardedNegShared: item-sharded weak form; ShardedGlobalBatch: strong form
    incl. its replicated variant) -- every loss, normalised users or not, interaction bias (strong form), Adam / lazy-sparse Adam,
    tiny and large catalogues, b = 32 .. 512 -- five steps through run_epoch against the float64 oracle.  Tolerance: the fixed one
    plus four times the float32 oracle's own distance from the float64 one.  NNCF_SOAK_SEEDS widens the sweep (tools/soak.sh)."""
    from nncf_amd.sharded import ShardedGlobalBatch
    rs = np.random.RandomState(13000 + seed)
    form = str(rs.choice(["weak", "strong"]))
    loss = str(rs.choice(["skip-gram", "mse", "log-loss", "max-margin"]))
    norm = bool(rs.randint(2))
    bias = str(rs.choice(["none", "none", "user", "item", "both"])) if form == "strong" else "none"
    opt = str(rs.choice(["sgd", "sgd", "adam", "sparse_adam"]))
    if bias != "none":
        opt = "sgd"          # Adam together with an interaction bias is refused on the strong form
    b = int(rs.choice([32, 64, 96, 160, 256, 512]))
    d = int(rs.choice([4, 20, 64, 100, 128]))
    nu, ni = int(rs.choice([2, 7, 40, 400, 3000])), int(rs.choice([2, 5, 33, 300, 2000]))
    steps = 5
    lr = 0.05 if opt == "sgd" else 0.01
    gamma = 0.1 if loss == "max-margin" else 10.
    kw = dict(user_count=nu, item_count=ni, dim=d, batch_size_p=b, num_negatives=0, loss=loss, loss_gamma=gamma,
              neg_loss_weight=16., u_reg=1e-3, learn_rate=lr, normalize_user=norm, normalize_item=False)
    ekw, okw = dict(kw), {}
    if bias != "none":
        ekw["interaction_bias"] = okw["interaction_bias"] = bias
    if opt != "sgd":
        ekw["optimizer"] = okw["optimizer"] = opt
    info = (form, loss, norm, bias, opt, b, d, nu, ni)
    be = HipBackend(ekw, nu, ni, 0, world=1, item_shard=ni, item_offset=0, seed=seed, mode=form, user_shard=nu)
    conf = so.OracleConf(batch_size_p=b, num_negatives=0, loss=loss, loss_gamma=gamma, neg_loss_weight=16., u_reg=1e-3,
                         learn_rate=lr, model_choice="mf", emb_normalization=norm, **okw)
    U0, V0 = be.U.cpu().numpy()[:, :d].copy(), be.V.cpu().numpy()[:, :d].copy()
    o64, o32 = so.StepOracle(U0, V0, conf, dtype=np.float64), so.StepOracle(U0, V0, conf, dtype=np.float32)
    uid, cid = rs.randint(0, nu, steps * b), rs.randint(0, ni, steps * b)
    if len(np.unique(cid[:b])) < 2:
        cid[0], cid[1] = 0, 1
    tr = (ShardedGlobalBatch if form == "strong" else ShardedNegShared)(be, nu, ni, b, d)
    losses = torch.zeros(steps, dtype=torch.float32, device=be.device)
    tr.run_epoch(be.ids(uid), be.ids(cid), steps, losses)
    be.engine.check()
    l = losses.cpu().numpy()
    for t in range(steps):
        sl = slice(t * b, (t + 1) * b)
        if len(np.unique(cid[sl])) < 2:
            return           # a one-item batch: the shared-negative losses are not defined (the tests above pin that case)
        want, w32 = o64.step_neg_shared(uid[sl], cid[sl]), o32.step_neg_shared(uid[sl], cid[sl])
        assert abs(l[t] - want) <= 1e-5 * max(1, abs(want)) + 4 * abs(w32 - want), (t, info, l[t], want)
    far = lambda a32, a64: 4 * float(np.abs(np.asarray(a32, dtype=np.float64) - a64).max())   # noqa: E731
    base = (2e-5 if opt == "sgd" else 5e-5) * max(1.0, b / (512.0 * min(nu, ni)) * 8)
    pairs = [(be.U.cpu().numpy()[:, :d], o64.U, o32.U), (be.V.cpu().numpy()[:, :d], o64.V, o32.V)]
    if bias in ("user", "both"):
        pairs.append((be.bu.cpu().numpy(), o64.ub, o32.ub))
    if bias in ("item", "both"):
        pairs.append((be.bi.cpu().numpy(), o64.cb, o32.cb))
    for g, w64, w32_t in pairs:
        bad = np.abs(g - w64) > base + far(w32_t, w64) + (1e-4 if opt == "sgd" else 5e-4) * np.abs(w64)
        allowed = 0 if opt == "sgd" else max(d, int(1e-5 * bad.size))   # Adam: see test_random_shapes_with_interaction_bias_and_adam_match
        assert bad.sum() <= allowed, (info, int(bad.sum()), float(np.abs(g - w64).max()))
