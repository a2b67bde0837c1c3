"""GPU: the HIP step kernels (through the C ABI) against the numpy oracle on identical inputs.
Tolerances: loss 1e-5 relative (north_star), table rows 1e-5 relative / 2e-6 absolute after the
update (fp32 summation order differs between MFMA/shuffle reductions and numpy)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import torch  # noqa: E402

from oracle import step_oracle as so  # noqa: E402
from nncf_amd.engine import Engine, plan_ids  # noqa: E402

LOSSES = ["skip-gram", "mse", "log-loss", "max-margin"]
RTOL_LOSS = 1e-5


def _mk(loss, b, k, d, nu, ni, model, seed=0, lr=0.05, u_reg=1e-3):
    gamma = {"max-margin": 0.1, "log-loss": 10.}.get(loss)
    conf = so.OracleConf(batch_size_p=b, num_negatives=k, loss=loss, loss_gamma=gamma, u_reg=u_reg, learn_rate=lr,
                         model_choice=model)
    eng = Engine(nu, ni, d, b, k, loss, conf.loss_gamma, conf.neg_loss_weight, conf.u_reg, conf.learn_rate,
                 conf.normalize_user, conf.normalize_item, seed=seed, init_scale=0.5)
    orc = so.StepOracle(eng.U.cpu().numpy(), eng.V.cpu().numpy(), conf, dtype=np.float32)
    return conf, eng, orc


def _check_tables(eng, orc):
    np.testing.assert_allclose(eng.U.cpu().numpy(), orc.U, rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(eng.V.cpu().numpy(), orc.V, rtol=1e-5, atol=2e-6)


def test_plan_ids_matches_tf_unique():
    rs = np.random.RandomState(0)
    for n, hi in [(1, 5), (2, 1), (37, 10), (512, 60), (512, 100000), (1024, 300), (1500, 200), (5632, 900), (8192, 17000),
                  # above 8192 the per-entry arrays leave LDS; above 16384 the sort keys too (one 16384-key tile, csrc/plan.hip)
                  (12000, 3000), (16384, 50), (16385, 40000), (20000, 7), (30001, 2000), (32768, 1), (32768, 100000), (32768, 3)]:
        ids = rs.randint(0, hi, n)
        order, seg_start, uniq, inverse, nseg = plan_ids(torch.as_tensor(ids, device="cuda"))
        u_ref, x_ref = so.tf_unique(ids)
        assert nseg == len(u_ref)
        assert np.array_equal(uniq, u_ref) and np.array_equal(inverse, x_ref)
        # order: positions sorted by (id, position); segments are maximal runs of equal ids
        keys = ids[order].astype(np.int64) * 100000 + order
        assert np.all(np.diff(keys) > 0)
        assert seg_start[0] == 0 and seg_start[-1] == n
        desc = plan_ids.last_desc
        for s in range(nseg):
            seg = ids[order[seg_start[s]:seg_start[s + 1]]]
            assert len(seg) > 0 and np.all(seg == seg[0])
            assert desc[s].tolist() == [seg[0], order[seg_start[s]], seg_start[s], len(seg)]
            assert np.all(plan_ids.last_slot_seg[seg_start[s]:seg_start[s + 1]] == s)
        # chunks: whole segments, consecutive, covering [0, n); a chunk only exceeds 16 slots by
        # the tail of its last segment
        ch = plan_ids.last_chunks
        assert ch[0] == 0 and ch[-1] == n and np.all(np.diff(ch) > 0)
        assert set(ch.tolist()) <= set(seg_start.tolist())
        for c0, c1 in zip(ch[:-1], ch[1:]):
            last_head = seg_start[np.searchsorted(seg_start, c1, side="left") - 1]
            assert last_head - c0 < 16


@pytest.mark.parametrize("loss", LOSSES)
@pytest.mark.parametrize("model", ["mf", "content"])
def test_original_step_matches_oracle(loss, model):
    b, k, d = 64, 5, 20
    conf, eng, orc = _mk(loss, b, k, d, 50, 40, model)
    rs = np.random.RandomState(1)
    for step in range(3):
        pu, pi = rs.randint(0, 50, b), rs.randint(0, 40, b)
        negs = rs.randint(0, 40, b * k)
        uid, cid, y = so.assemble_original(np.stack([pu, pi, np.ones(b, dtype=int)], 1), negs, loss)
        want = orc.step_original(uid, cid, y)
        got = eng.step_original(eng.ids(uid[:b]), eng.ids(cid)).item()
        assert abs(got - want) <= RTOL_LOSS * max(1.0, abs(want)), (step, got, want)
        _check_tables(eng, orc)


@pytest.mark.parametrize("loss", LOSSES)
@pytest.mark.parametrize("model", ["mf", "content"])
def test_neg_shared_step_matches_oracle(loss, model):
    b, d = 64, 20
    conf, eng, orc = _mk(loss, b, 0, d, 50, 300, model, seed=2)
    rs = np.random.RandomState(3)
    for step in range(3):
        uid, cid = rs.randint(0, 50, b), rs.randint(0, 300, b)
        cid[5] = cid[40]  # duplicate item => extra negative column, shared row update
        want = orc.step_neg_shared(uid, cid)
        got = eng.step_neg_shared(eng.ids(uid), eng.ids(cid)).item()
        assert abs(got - want) <= RTOL_LOSS * max(1.0, abs(want)), (step, got, want)
        _check_tables(eng, orc)


@pytest.mark.parametrize("loss", LOSSES)
@pytest.mark.parametrize("model", ["mf", "content"])
def test_group_neg_shared_step_matches_oracle(loss, model):
    b, d = 64, 20
    conf, eng, orc = _mk(loss, b, 0, d, 50, 300, model, seed=4)
    rs = np.random.RandomState(5)
    for step in range(3):
        uid = rs.randint(0, 50, b)
        cid = np.repeat(rs.choice(300, b // 4, replace=False), 4)
        cid[-1] = cid[0]
        want = orc.step_group_neg_shared(uid, cid)
        out, cu, cx, nu = eng.step_group_neg_shared(eng.ids(uid), eng.ids(cid), want_unique=True)
        got = out.item()
        n_u = int(nu.item())
        assert n_u == len(orc.cid_u)
        assert np.array_equal(cu.cpu().numpy()[:n_u], orc.cid_u)  # tf.unique order, bit-exact
        assert np.array_equal(cx.cpu().numpy(), orc.cid_x)
        assert abs(got - want) <= RTOL_LOSS * max(1.0, abs(want)), (step, got, want)
        _check_tables(eng, orc)


@pytest.mark.parametrize("scheme", ["original", "neg_shared", "group_neg_shared"])
def test_citeulike_shape_d100_b512(scheme):
    """BASELINE shapes: b = 512, d = 100 (k = 10 for original), 16K users x 17K items."""
    b, d, k = 512, 100, 10
    loss = "log-loss" if scheme == "group_neg_shared" else "skip-gram"
    conf, eng, orc = _mk(loss, b, k, d, 16000, 17000, "mf", seed=6, lr=0.01, u_reg=1e-6)
    rs = np.random.RandomState(7)
    zipf = 1.0 / np.arange(1, 17001)
    zipf /= zipf.sum()
    for step in range(2):
        uid = rs.randint(0, 16000, b)
        if scheme == "group_neg_shared":
            cid = np.repeat(rs.choice(17000, b // 4, p=zipf), 4)
        else:
            cid = rs.choice(17000, b, p=zipf)
        if scheme == "original":
            negs = rs.choice(17000, b * k, p=zipf)
            u, c, y = so.assemble_original(np.stack([uid, cid, np.ones(b, dtype=int)], 1), negs, loss)
            want = orc.step_original(u, c, y)
            got = eng.step_original(eng.ids(uid), eng.ids(c)).item()
        elif scheme == "neg_shared":
            want = orc.step_neg_shared(uid, cid)
            got = eng.step_neg_shared(eng.ids(uid), eng.ids(cid)).item()
        else:
            want = orc.step_group_neg_shared(uid, cid)
            got = eng.step_group_neg_shared(eng.ids(uid), eng.ids(cid)).item()
        assert abs(got - want) <= RTOL_LOSS * max(1.0, abs(want)), (step, got, want)
        _check_tables(eng, orc)


def test_forward_only_leaves_tables_untouched():
    conf, eng, orc = _mk("skip-gram", 64, 0, 20, 50, 300, "mf", seed=8)
    U0, V0 = eng.U.clone(), eng.V.clone()
    rs = np.random.RandomState(9)
    uid, cid = rs.randint(0, 50, 64), rs.randint(0, 300, 64)
    want = orc.step_neg_shared(uid, cid, update=False)
    got = eng.step_neg_shared(eng.ids(uid), eng.ids(cid), update=False).item()
    assert abs(got - want) <= RTOL_LOSS * max(1.0, abs(want))
    assert torch.equal(U0, eng.U) and torch.equal(V0, eng.V)


def test_predict_scores_matches_oracle():
    for model, loss in [("mf", "skip-gram"), ("mf", "log-loss"), ("content", "log-loss")]:
        conf, eng, orc = _mk(loss, 64, 0, 100, 300, 500, model, seed=10)
        rs = np.random.RandomState(11)
        uid, cid = rs.randint(0, 300, 77), rs.randint(0, 500, 45)  # ragged: not multiples of the tile
        got = eng.predict_scores(eng.ids(uid), eng.ids(cid)).cpu().numpy()
        want = orc.predict_neg_shared(uid, cid)
        np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-6)


def test_epoch_runner_equals_step_by_step():
    """nncf_epoch_* (no host sync, chunked plans) == the same steps issued one by one; bitwise."""
    b, d, k, steps = 64, 20, 3, 70  # > one plan chunk (64)
    rs = np.random.RandomState(12)
    uid_all, cid_all = rs.randint(0, 50, b * steps), rs.randint(0, 300, b * steps)
    neg_all = rs.randint(0, 300, b * k * steps)
    for scheme in ("original", "neg_shared", "group_neg_shared"):
        _, e1, _ = _mk("log-loss", b, k, d, 50, 300, "mf", seed=13)
        _, e2, _ = _mk("log-loss", b, k, d, 50, 300, "mf", seed=13)
        ua, ca, na = e1.ids(uid_all), e1.ids(cid_all), e1.ids(neg_all)
        if scheme == "original":
            l1 = e1.epoch_original(ua, ca, na, steps).cpu().numpy()
        elif scheme == "neg_shared":
            l1 = e1.epoch_neg_shared(ua, ca, steps).cpu().numpy()
        else:
            l1 = e1.epoch_group_neg_shared(ua, ca, steps).cpu().numpy()
        l2 = []
        for t in range(steps):
            u, c = ua[t * b:(t + 1) * b], ca[t * b:(t + 1) * b]
            if scheme == "original":
                cc = torch.cat([c, na[t * b * k:(t + 1) * b * k]])
                l2.append(e2.step_original(u, cc).item())
            elif scheme == "neg_shared":
                l2.append(e2.step_neg_shared(u, c).item())
            else:
                l2.append(e2.step_group_neg_shared(u, c).item())
        assert np.array_equal(l1, np.array(l2, dtype=np.float32)), scheme
        assert torch.equal(e1.U, e2.U) and torch.equal(e1.V, e2.V), scheme


@pytest.mark.parametrize("kind", ["adam", "sparse_adam"])
@pytest.mark.parametrize("scheme", ["original", "neg_shared", "group_neg_shared"])
def test_adam_steps_match_oracle(kind, scheme):
    """Keras Adam (dense: idle rows move too) and the lazy-sparse TF variant against the oracle over
    several steps (moments, bias correction, duplicate ids summed), tables and moments compared."""
    b, k, d, nu, ni = 64, 3, 24, 90, 200
    loss = "log-loss" if scheme == "group_neg_shared" else "skip-gram"
    gamma = {"log-loss": 10.}.get(loss)
    conf = so.OracleConf(batch_size_p=b, num_negatives=k, loss=loss, loss_gamma=gamma, u_reg=1e-3, learn_rate=0.01,
                         model_choice="mf", optimizer=kind)
    eng = Engine(nu, ni, d, b, k, loss, conf.loss_gamma, conf.neg_loss_weight, conf.u_reg, conf.learn_rate,
                 conf.normalize_user, conf.normalize_item, seed=3, init_scale=0.5)
    eng.set_optimizer(kind)
    orc = so.StepOracle(eng.U.cpu().numpy(), eng.V.cpu().numpy(), conf, dtype=np.float32)
    rs = np.random.RandomState(5)
    for step in range(5):
        uid, cid = rs.randint(0, nu, b), rs.randint(0, ni // 2, b)      # half of the items stay idle
        if scheme == "group_neg_shared":
            cid = np.repeat(cid[::4], 4)
        if scheme == "original":
            negs = rs.randint(0, ni // 2, b * k)
            u, c, y = so.assemble_original(np.stack([uid, cid, np.ones(b, dtype=np.int64)], 1), negs, loss)
            want = orc.step_original(u, c, y)
            got = eng.step_original(eng.ids(uid), eng.ids(c)).item()
        elif scheme == "neg_shared":
            want = orc.step_neg_shared(uid, cid)
            got = eng.step_neg_shared(eng.ids(uid), eng.ids(cid)).item()
        else:
            want = orc.step_group_neg_shared(uid, cid)
            got = eng.step_group_neg_shared(eng.ids(uid), eng.ids(cid)).item()
        assert abs(got - want) <= 1e-5 * max(1, abs(want)), (step, got, want)
        # Adam divides by sqrt(v): a tiny gradient difference is amplified where v ~ g^2
        np.testing.assert_allclose(eng.U.cpu().numpy(), orc.U, rtol=2e-4, atol=2e-5)
        np.testing.assert_allclose(eng.V.cpu().numpy(), orc.V, rtol=2e-4, atol=2e-5)
        np.testing.assert_allclose(eng.opt_state["mU"].cpu().numpy(), orc.slot_u.m, rtol=2e-4, atol=1e-7)
        np.testing.assert_allclose(eng.opt_state["vV"].cpu().numpy(), orc.slot_v.v, rtol=2e-4, atol=1e-9)
    idle = np.arange(ni // 2, ni)
    if kind == "sparse_adam":
        assert np.all(eng.opt_state["mV"].cpu().numpy()[idle] == 0)


def test_adam_epoch_runner_equals_step_by_step():
    b, d, k, steps = 64, 20, 3, 70
    rs = np.random.RandomState(12)
    uid_all, cid_all = rs.randint(0, 50, b * steps), rs.randint(0, 300, b * steps)
    for kind in ("adam", "sparse_adam"):
        _, e1, _ = _mk("skip-gram", b, k, d, 50, 300, "mf", seed=13, lr=0.01)
        _, e2, _ = _mk("skip-gram", b, k, d, 50, 300, "mf", seed=13, lr=0.01)
        e1.set_optimizer(kind)
        e2.set_optimizer(kind)
        ua, ca = e1.ids(uid_all), e1.ids(cid_all)
        l1 = e1.epoch_neg_shared(ua, ca, steps).cpu().numpy()
        l2 = [e2.step_neg_shared(ua[t * b:(t + 1) * b], ca[t * b:(t + 1) * b]).item() for t in range(steps)]
        assert np.array_equal(l1, np.array(l2, dtype=np.float32)), kind
        assert torch.equal(e1.U, e2.U) and torch.equal(e1.V, e2.V), kind


def test_dim_50_is_padded_exactly():
    """user_dim = item_dim = 50 (the reference's default): tables are 52 wide with zero pad columns;
    steps equal the oracle run on the 50 logical columns; pads stay exactly zero (SGD and Adam)."""
    b, k, nu, ni = 64, 3, 80, 120
    for kind in ("sgd", "adam"):
        conf = so.OracleConf(batch_size_p=b, num_negatives=k, loss="log-loss", loss_gamma=10., u_reg=1e-3, learn_rate=0.02,
                             model_choice="mf", optimizer=kind)
        eng = Engine(nu, ni, 50, b, k, "log-loss", conf.loss_gamma, conf.neg_loss_weight, conf.u_reg, conf.learn_rate,
                     conf.normalize_user, conf.normalize_item, seed=3, init_scale=0.5)
        assert (eng.dim, eng.dim_p, tuple(eng.U.shape)) == (50, 52, (nu, 52))
        if kind != "sgd":
            eng.set_optimizer(kind)
        orc = so.StepOracle(eng.U.cpu().numpy()[:, :50], eng.V.cpu().numpy()[:, :50], conf, dtype=np.float32)
        rs = np.random.RandomState(5)
        for step in range(3):
            uid, cid = rs.randint(0, nu, b), rs.randint(0, ni, b)
            want = orc.step_neg_shared(uid, cid)
            got = eng.step_neg_shared(eng.ids(uid), eng.ids(cid)).item()
            assert abs(got - want) <= 1e-5 * max(1, abs(want)), (kind, step, got, want)
        tol = dict(rtol=2e-4, atol=2e-5) if kind == "adam" else dict(rtol=1e-5, atol=2e-6)
        np.testing.assert_allclose(eng.U.cpu().numpy()[:, :50], orc.U, **tol)
        np.testing.assert_allclose(eng.V.cpu().numpy()[:, :50], orc.V, **tol)
        assert np.all(eng.U.cpu().numpy()[:, 50:] == 0) and np.all(eng.V.cpu().numpy()[:, 50:] == 0)


def test_edge_cases_degenerate_batches():
    """Collisions and empties: a batch that is one (user, item) pair repeated, a batch with one user and
    all-distinct items, an epoch of zero steps, a prediction with zero rows, invalid shapes."""
    b, k, d, nu, ni = 64, 3, 16, 40, 90
    for loss in ("skip-gram", "log-loss"):
        conf, eng, orc = _mk(loss, b, k, d, nu, ni, "mf", seed=2)
        uid, cid = np.full(b, 7), np.full(b, 11)                          # every row the same pair
        assert abs(eng.step_neg_shared(eng.ids(uid), eng.ids(cid)).item() - orc.step_neg_shared(uid, cid)) < 1e-5 * 10
        _check_tables(eng, orc)
        want = orc.step_group_neg_shared(uid, cid)                        # n_u = 1: a single shared item
        got, cu, cx, n_u = eng.step_group_neg_shared(eng.ids(uid), eng.ids(cid), want_unique=True)
        assert int(n_u.item()) == 1 and abs(got.item() - want) <= 1e-5 * max(1, abs(want))
        _check_tables(eng, orc)
        uid2, cid2 = np.full(b, 3), np.arange(b)                          # one user, b different items
        negs = np.full(b * k, 5)                                          # one negative for everybody
        u, c, y = so.assemble_original(np.stack([uid2, cid2, np.ones(b, dtype=np.int64)], 1), negs, loss)
        want = orc.step_original(u, c, y)
        got = eng.step_original(eng.ids(uid2), eng.ids(c)).item()
        assert abs(got - want) <= 1e-5 * max(1, abs(want))
        _check_tables(eng, orc)
    U0, V0 = eng.U.clone(), eng.V.clone()
    empty = torch.zeros(0, dtype=torch.int32, device=eng.device)
    assert eng.epoch_neg_shared(empty, empty, 0).numel() == 0           # zero steps: nothing runs
    assert eng.epoch_original(empty, empty, empty, 0).numel() == 0
    assert torch.equal(eng.U, U0) and torch.equal(eng.V, V0)
    assert tuple(eng.predict_scores(empty, eng.ids(np.arange(5))).shape) == (0, 5)
    with pytest.raises(ValueError):                                       # b not a multiple of the tile edge
        Engine(nu, ni, d, 48, 0, "skip-gram", 10., 128., 1e-6, 0.1, False, False).step_neg_shared(
            eng.ids(np.zeros(48)), eng.ids(np.zeros(48)))
    with pytest.raises(AssertionError):
        Engine(nu, ni, d, 64, 0, "hinge", 10., 128., 1e-6, 0.1, False, False)   # objectives.py:72


@pytest.mark.parametrize("loss", LOSSES)
@pytest.mark.parametrize("scheme", ["original", "neg_shared", "group_neg_shared"])
def test_interaction_bias_matches_oracle(scheme, loss):
    """interaction_bias = 'both' (interaction_dot.py:62-71): scores, losses, table rows and both bias
    vectors after several steps; duplicates in users and items; predictions include the biases."""
    b, k, d, nu, ni = 64, 3, 16, 50, 90
    gamma = {"max-margin": 0.1, "log-loss": 10.}.get(loss)
    conf = so.OracleConf(batch_size_p=b, num_negatives=k, loss=loss, loss_gamma=gamma, u_reg=1e-3, learn_rate=0.05,
                         model_choice="mf", interaction_bias="both")
    eng = Engine(nu, ni, d, b, k, loss, conf.loss_gamma, conf.neg_loss_weight, conf.u_reg, conf.learn_rate,
                 conf.normalize_user, conf.normalize_item, seed=1, init_scale=0.5, interaction_bias="both")
    orc = so.StepOracle(eng.U.cpu().numpy(), eng.V.cpu().numpy(), conf, dtype=np.float32)
    rs = np.random.RandomState(3)
    eng.bias["ub"].copy_(torch.as_tensor(rs.randn(nu).astype(np.float32) * 0.2))
    eng.bias["cb"].copy_(torch.as_tensor(rs.randn(ni).astype(np.float32) * 0.2))
    orc.ub[:] = eng.bias["ub"].cpu().numpy()
    orc.cb[:] = eng.bias["cb"].cpu().numpy()
    for step in range(3):
        uid, cid = rs.randint(0, nu, b), rs.randint(0, ni, b)
        if scheme == "group_neg_shared":
            cid = np.repeat(cid[::4], 4)
        if scheme == "original":
            negs = rs.randint(0, ni, b * k)
            u, c, y = so.assemble_original(np.stack([uid, cid, np.ones(b, dtype=np.int64)], 1), negs, loss)
            want = orc.step_original(u, c, y)
            got = eng.step_original(eng.ids(uid), eng.ids(c)).item()
        elif scheme == "neg_shared":
            want = orc.step_neg_shared(uid, cid)
            got = eng.step_neg_shared(eng.ids(uid), eng.ids(cid)).item()
        else:
            want = orc.step_group_neg_shared(uid, cid)
            got = eng.step_group_neg_shared(eng.ids(uid), eng.ids(cid)).item()
        assert abs(got - want) <= 1e-5 * max(1, abs(want)), (step, got, want)
        _check_tables(eng, orc)
        np.testing.assert_allclose(eng.bias["ub"].cpu().numpy(), orc.ub, rtol=1e-5, atol=2e-6)
        np.testing.assert_allclose(eng.bias["cb"].cpu().numpy(), orc.cb, rtol=1e-5, atol=2e-6)
    users, items = np.arange(20), np.arange(0, ni, 2)
    got = eng.predict_scores(eng.ids(users), eng.ids(items)).cpu().numpy()
    np.testing.assert_allclose(got, orc.predict_neg_shared(users, items), rtol=1e-5, atol=2e-6)


def test_interaction_bias_adam_and_rowwise():
    """bias under Keras Adam / lazy-sparse Adam (moments, idle entries) and through the row-wise kernels."""
    b, k, d, nu, ni = 64, 3, 16, 50, 90
    for kind in ("adam", "sparse_adam"):
        conf = so.OracleConf(batch_size_p=b, num_negatives=k, loss="skip-gram", u_reg=1e-3, learn_rate=0.01, model_choice="mf",
                             interaction_bias="item", optimizer=kind)
        eng = Engine(nu, ni, d, b, k, "skip-gram", conf.loss_gamma, conf.neg_loss_weight, conf.u_reg, conf.learn_rate,
                     conf.normalize_user, conf.normalize_item, seed=1, init_scale=0.5, interaction_bias="item")
        eng.set_optimizer(kind)
        orc = so.StepOracle(eng.U.cpu().numpy(), eng.V.cpu().numpy(), conf, dtype=np.float32)
        rs = np.random.RandomState(3)
        for step in range(4):
            uid, cid = rs.randint(0, nu, b), rs.randint(0, ni // 2, b)
            want = orc.step_neg_shared(uid, cid)
            got = eng.step_neg_shared(eng.ids(uid), eng.ids(cid)).item()
            assert abs(got - want) <= 1e-5 * max(1, abs(want)), (kind, step, got, want)
            np.testing.assert_allclose(eng.bias["cb"].cpu().numpy(), orc.cb, rtol=2e-4, atol=2e-5)
        assert eng.bias["ub"] is None
    conf = so.OracleConf(batch_size_p=b, num_negatives=k, loss="mse", u_reg=1e-3, learn_rate=0.05, model_choice="mf",
                         interaction_bias="both")
    eng = Engine(nu, ni, d, b, k, "mse", conf.loss_gamma, conf.neg_loss_weight, conf.u_reg, conf.learn_rate,
                 conf.normalize_user, conf.normalize_item, seed=1, init_scale=0.5, interaction_bias="both")
    orc = so.StepOracle(eng.U.cpu().numpy(), eng.V.cpu().numpy(), conf, dtype=np.float32)
    n = b * (1 + k)
    rs = np.random.RandomState(0)
    for step in range(3):
        uid, cid = rs.randint(0, nu, n), rs.randint(0, ni, n)
        y = np.where(rs.rand(n) < 0.3, 1, 0)
        want = orc.step_original(uid, cid, y)
        got = eng.step_rowwise(eng.ids(uid), eng.ids(cid), eng.ids(y)).item()
        assert abs(got - want) <= 1e-5 * max(1, abs(want))
        np.testing.assert_allclose(eng.bias["ub"].cpu().numpy(), orc.ub, rtol=1e-5, atol=2e-6)
        np.testing.assert_allclose(eng.bias["cb"].cpu().numpy(), orc.cb, rtol=1e-5, atol=2e-6)


# ---- BASELINE configs[3]: 10M users x 1M items, d = 128, neg_shared (and original), id tables ---------------
def _compact_oracle(eng, conf, uid_steps, cid_steps, V=None, item_offset=0):
    """Oracle over ONLY the rows the steps touch (the step reads and writes nothing else): gathers those rows from
    the device tables, remaps the ids; returns (oracle, user ids, item ids, remap functions)."""
    users = np.unique(np.concatenate(uid_steps))
    items = np.unique(np.concatenate(cid_steps))
    Vt = eng.V if V is None else V
    Us = eng.U[torch.as_tensor(users, device=eng.U.device)].cpu().numpy()
    Vs = Vt[torch.as_tensor(items - item_offset, device=Vt.device)].cpu().numpy()
    orc = so.StepOracle(Us, Vs, conf, dtype=np.float32)
    return orc, users, items, (lambda u: np.searchsorted(users, u)), (lambda c: np.searchsorted(items, c))


@pytest.mark.parametrize("scheme,loss", [("neg_shared", "skip-gram"), ("neg_shared", "log-loss"), ("original", "skip-gram")])
def test_config4_shape_10m_x_1m_d128(scheme, loss):
    """BASELINE configs[3] on one GPU: 10M x 128 user table (5.1 GB), 1M x 128 item table, b = 512, Zipf links;
    3 steps, loss and every touched row against the oracle."""
    users, items, b, d = 10000000, 1000000, 512, 128
    k = 10 if scheme == "original" else 0
    conf = so.OracleConf(batch_size_p=b, num_negatives=k, loss=loss, u_reg=1e-6, learn_rate=0.05, model_choice="mf")
    eng = Engine(users, items, d, b, k, loss, conf.loss_gamma, conf.neg_loss_weight, conf.u_reg, conf.learn_rate,
                 conf.normalize_user, conf.normalize_item, seed=11)
    assert eng.U.shape == (users, d) and eng.V.shape == (items, d)
    rs = np.random.RandomState(5)
    zipf = lambda n, cnt: np.minimum((cnt * rs.power(0.35, n)).astype(np.int64), cnt - 1)  # noqa: E731
    steps = []
    for _ in range(3):
        u, c = zipf(b, users), zipf(b, items)
        u[0], c[0] = users - 1, items - 1                 # the last rows of both tables
        if scheme == "original":
            uu, cc, yy = so.assemble_original(np.stack([u, c, np.ones(b, dtype=np.int64)], 1), zipf(b * k, items), loss)
            steps.append((uu, cc, yy))
        else:
            steps.append((u, c, None))
    orc, us, its, mu, mc = _compact_oracle(eng, conf, [s[0] for s in steps], [s[1] for s in steps])
    for t, (u, c, y) in enumerate(steps):
        if scheme == "original":
            want = orc.step_original(mu(u), mc(c), y)
            got = eng.step_original(eng.ids(u[:b]), eng.ids(c)).item()
        else:
            want = orc.step_neg_shared(mu(u), mc(c))
            got = eng.step_neg_shared(eng.ids(u), eng.ids(c)).item()
        assert abs(got - want) <= RTOL_LOSS * max(1.0, abs(want)), (t, got, want)
        np.testing.assert_allclose(eng.U[torch.as_tensor(us, device=eng.U.device)].cpu().numpy(), orc.U, rtol=1e-5, atol=2e-6)
        np.testing.assert_allclose(eng.V[torch.as_tensor(its, device=eng.V.device)].cpu().numpy(), orc.V, rtol=1e-5, atol=2e-6)
    # nothing else moved: spot-check untouched rows against a fresh engine of the same seed
    del orc
    ref = Engine(users, items, d, b, k, loss, conf.loss_gamma, conf.neg_loss_weight, conf.u_reg, conf.learn_rate,
                 conf.normalize_user, conf.normalize_item, seed=11)
    probe = np.setdiff1d(rs.randint(0, users, 4096), us)
    assert torch.equal(eng.U[torch.as_tensor(probe, device=eng.U.device)], ref.U[torch.as_tensor(probe, device=eng.U.device)])
    probe = np.setdiff1d(rs.randint(0, items, 4096), its)
    assert torch.equal(eng.V[torch.as_tensor(probe, device=eng.V.device)], ref.V[torch.as_tensor(probe, device=eng.V.device)])


@pytest.mark.parametrize("loss", LOSSES)
@pytest.mark.parametrize("scheme", ["original", "neg_shared", "group_neg_shared"])
def test_d128_all_schemes_and_losses(scheme, loss):
    """d = 128 (configs[3]'s row width: the full 16-lane x 2 x float4 row) in the scheme x loss matrix"""
    b, k, d = 64, 3, 128
    conf, eng, orc = _mk(loss, b, k if scheme == "original" else 0, d, 70, 90, "mf", seed=9)
    rs = np.random.RandomState(13)
    for step in range(2):
        uid, cid = rs.randint(0, 70, b), rs.randint(0, 90, b)
        if scheme == "original":
            u, c, y = so.assemble_original(np.stack([uid, cid, np.ones(b, dtype=int)], 1), rs.randint(0, 90, b * k), loss)
            want = orc.step_original(u, c, y)
            got = eng.step_original(eng.ids(uid), eng.ids(c)).item()
        elif scheme == "neg_shared":
            want = orc.step_neg_shared(uid, cid)
            got = eng.step_neg_shared(eng.ids(uid), eng.ids(cid)).item()
        else:
            cid = np.repeat(cid[:b // 4], 4)
            want = orc.step_group_neg_shared(uid, cid)
            got = eng.step_group_neg_shared(eng.ids(uid), eng.ids(cid)).item()
        assert abs(got - want) <= RTOL_LOSS * max(1.0, abs(want)), (step, got, want)
        _check_tables(eng, orc)


@pytest.mark.parametrize("scheme", ["original", "neg_shared", "group_neg_shared"])
def test_sparse_adam_as_written_matches_oracle(scheme):
    """sparse_adam_duplicates = 'as_written' (utils/optimizer.py:108-134 on duplicate indices: moments from the last
    occurrence, the step subtracted once per occurrence) vs the oracle, with heavily repeated users and items."""
    b, k, d, nu, ni = 64, 3, 24, 30, 40
    loss = "skip-gram"
    conf = so.OracleConf(batch_size_p=b, num_negatives=k if scheme == "original" else 0, loss=loss, u_reg=1e-3,
                         learn_rate=0.01, model_choice="mf", optimizer="sparse_adam")
    conf.sparse_adam_duplicates = "as_written"
    eng = Engine(nu, ni, d, b, conf.num_negatives, loss, conf.loss_gamma, conf.neg_loss_weight, conf.u_reg, conf.learn_rate,
                 conf.normalize_user, conf.normalize_item, seed=3, init_scale=0.5)
    eng.set_optimizer("sparse_adam")
    eng.set_sparse_adam_duplicates("as_written")
    orc = so.StepOracle(eng.U.cpu().numpy(), eng.V.cpu().numpy(), conf, dtype=np.float32)
    sum_orc = so.StepOracle(eng.U.cpu().numpy(), eng.V.cpu().numpy(),
                            so.OracleConf(batch_size_p=b, num_negatives=conf.num_negatives, loss=loss, u_reg=1e-3, learn_rate=0.01,
                                          model_choice="mf", optimizer="sparse_adam"), dtype=np.float32)
    rs = np.random.RandomState(5)
    for step in range(4):
        uid, cid = rs.randint(0, nu, b), rs.randint(0, ni, b)
        if scheme == "group_neg_shared":
            cid = np.repeat(cid[::4], 4)
        if scheme == "original":
            u, c, y = so.assemble_original(np.stack([uid, cid, np.ones(b, dtype=np.int64)], 1), rs.randint(0, ni, b * k), loss)
            want = orc.step_original(u, c, y)
            sum_orc.step_original(u, c, y)
            got = eng.step_rowwise(eng.ids(u), eng.ids(c), eng.ids(y)).item()   # every row its own lookup position
            with pytest.raises(ValueError):
                eng.step_original(eng.ids(uid), eng.ids(c))
        elif scheme == "neg_shared":
            want = orc.step_neg_shared(uid, cid)
            sum_orc.step_neg_shared(uid, cid)
            got = eng.step_neg_shared(eng.ids(uid), eng.ids(cid)).item()
        else:
            want = orc.step_group_neg_shared(uid, cid)
            sum_orc.step_group_neg_shared(uid, cid)
            got = eng.step_group_neg_shared(eng.ids(uid), eng.ids(cid)).item()
        assert abs(got - want) <= 1e-5 * max(1, abs(want)), (step, got, want)
        np.testing.assert_allclose(eng.U.cpu().numpy(), orc.U, rtol=2e-4, atol=2e-5)
        np.testing.assert_allclose(eng.V.cpu().numpy(), orc.V, rtol=2e-4, atol=2e-5)
        np.testing.assert_allclose(eng.opt_state["mU"].cpu().numpy(), orc.slot_u.m, rtol=2e-4, atol=1e-7)
        np.testing.assert_allclose(eng.opt_state["vV"].cpu().numpy(), orc.slot_v.v, rtol=2e-4, atol=1e-9)
    assert np.abs(orc.U - sum_orc.U).max() > 1e-3      # the two semantics really differ on this data


@pytest.mark.timeout(300)
@pytest.mark.parametrize("seed", range(int(os.environ.get("NNCF_SOAK_SEEDS", "16"))))
def test_random_shapes_all_schemes_and_losses_match(seed):
    """Random shapes through the three schemes' default kernel paths (no path forced) -- every loss, 'mf' and 'content', b = 1 .. 512
    (odd sizes under `original`), k = 1 .. 10, d = 4 .. 128, catalogues from two rows (everything a duplicate) to thousands -- two steps each
    against the float64 oracle.  NNCF_SOAK_SEEDS=<n> widens the sweep (tools/soak.sh)."""
    rs = np.random.RandomState(9000 + seed)
    scheme = str(rs.choice(["original", "neg_shared", "group_neg_shared"]))
    loss = str(rs.choice(LOSSES))
    model = str(rs.choice(["mf", "content"]))
    d = int(rs.choice([4, 20, 50, 64, 100, 128]))
    nu = int(rs.choice([1, 2, 7, 40, 400, 3000]))
    ni = int(rs.choice([2, 5, 33, 300, 2000]))
    if scheme == "original":
        b, k = int(rs.choice([1, 3, 17, 64, 100, 256, 512])), int(rs.choice([1, 3, 5, 10]))
    else:
        b, k = int(rs.choice([32, 64, 96, 160, 256, 512])), 0   # the shared-negative schemes take multiples of 32 (anything else: ValueError)
    gamma = {"max-margin": 0.1, "log-loss": 10.}.get(loss)
    conf = so.OracleConf(batch_size_p=b, num_negatives=k, loss=loss, loss_gamma=gamma, u_reg=1e-3, learn_rate=0.05, model_choice=model)
    eng = Engine(nu, ni, d, b, k, loss, conf.loss_gamma, conf.neg_loss_weight, conf.u_reg, conf.learn_rate,
                 conf.normalize_user, conf.normalize_item, seed=seed, init_scale=0.5)
    orc = so.StepOracle(eng.U.cpu().numpy(), eng.V.cpu().numpy(), conf, dtype=np.float64)
    for step in range(2):
        pu, pi = rs.randint(0, nu, b), rs.randint(0, ni, b)
        if scheme == "original":
            uid, cid, y = so.assemble_original(np.stack([pu, pi, np.ones(b, dtype=int)], 1), rs.randint(0, ni, b * k), loss)
            want = orc.step_original(uid, cid, y)
            got = eng.step_original(eng.ids(uid[:b]), eng.ids(cid)).item()
        elif scheme == "neg_shared":
            if len(np.unique(pi)) < 2:
                continue
            want = orc.step_neg_shared(pu, pi)
            got = eng.step_neg_shared(eng.ids(pu), eng.ids(pi)).item()
        else:
            pi = np.repeat(pi[:b // 4], 4)
            if len(np.unique(pi)) < 2:
                continue
            want = orc.step_group_neg_shared(pu, pi)
            got = eng.step_group_neg_shared(eng.ids(pu), eng.ids(pi)).item()
        assert abs(got - want) <= RTOL_LOSS * max(1.0, abs(want)), (step, scheme, loss, model, b, k, d, nu, ni, got, want)
        # absolute term: the backward of x / |x| multiplies fp32's rounding by 1 / |x| (|x| ~ 0.5 sqrt(d / 3) at init)
        amp = max(1.0, 1.0 / (0.5 * np.sqrt(d / 3.0)))
        amp *= max(1.0, b * (1 + k) / (512.0 * min(nu, ni)))   # ... and grows with the terms that meet in one row (5,632 links onto 2 rows)
        np.testing.assert_allclose(eng.U.cpu().numpy(), orc.U, rtol=2e-5, atol=5e-6 * amp, err_msg=str((scheme, loss, model, b, k, d, nu, ni)))
        np.testing.assert_allclose(eng.V.cpu().numpy(), orc.V, rtol=2e-5, atol=5e-6 * amp, err_msg=str((scheme, loss, model, b, k, d, nu, ni)))


@pytest.mark.timeout(300)
@pytest.mark.parametrize("seed", range(int(os.environ.get("NNCF_SOAK_SEEDS", "16"))))
def test_random_shapes_with_interaction_bias_and_adam_match(seed):
    """The sweep above with the options that ride on the step kernels: interaction_bias = 'both' (interaction_dot.py:62-71), Keras
    Adam and the lazy-sparse Adam (utils/optimizer.py:108-147) -- three steps each (the moments matter from the second on).  Tolerance:
    the fixed one plus four times the float32 oracle's own distance from the float64 one (Adam divides by sqrt(v): where v ~ g^2 a
    rounding-sized gradient difference is a finite step)."""
    rs = np.random.RandomState(11000 + seed)
    scheme = str(rs.choice(["original", "neg_shared", "group_neg_shared"]))
    loss = str(rs.choice(LOSSES))
    bias = "both" if rs.randint(2) else None
    opt = str(rs.choice(["sgd", "adam", "sparse_adam"]))
    if bias and opt != "sgd" and scheme != "original":
        opt = "sgd"   # the shared-negative schemes take Adam or an interaction bias
    d = int(rs.choice([4, 20, 50, 64, 100, 128]))
    nu = int(rs.choice([1, 2, 7, 40, 400, 3000]))
    ni = int(rs.choice([2, 5, 33, 300, 2000]))
    if scheme == "original":
        b, k = int(rs.choice([1, 3, 17, 64, 100, 256, 512])), int(rs.choice([1, 3, 5, 10]))
    else:
        b, k = int(rs.choice([32, 64, 96, 160, 256, 512])), 0
    gamma = {"max-margin": 0.1, "log-loss": 10.}.get(loss)
    kw = dict(interaction_bias=bias) if bias else {}
    conf = so.OracleConf(batch_size_p=b, num_negatives=k, loss=loss, loss_gamma=gamma, u_reg=1e-3, learn_rate=0.05 if opt == "sgd" else 0.01,
                         model_choice="mf", optimizer=opt, **kw)
    eng = Engine(nu, ni, d, b, k, loss, conf.loss_gamma, conf.neg_loss_weight, conf.u_reg, conf.learn_rate,
                 conf.normalize_user, conf.normalize_item, seed=seed, init_scale=0.5, **kw)
    if opt != "sgd":
        eng.set_optimizer(opt)
    o64 = so.StepOracle(eng.U.cpu().numpy(), eng.V.cpu().numpy(), conf, dtype=np.float64)
    o32 = so.StepOracle(eng.U.cpu().numpy(), eng.V.cpu().numpy(), conf, dtype=np.float32)
    if bias:
        eng.bias["ub"].copy_(torch.as_tensor(rs.randn(nu).astype(np.float32) * 0.2))
        eng.bias["cb"].copy_(torch.as_tensor(rs.randn(ni).astype(np.float32) * 0.2))
        for o in (o64, o32):
            o.ub[:] = eng.bias["ub"].cpu().numpy()
            o.cb[:] = eng.bias["cb"].cpu().numpy()
    info = (scheme, loss, bias, opt, b, k, d, nu, ni)
    for step in range(3):
        pu, pi = rs.randint(0, nu, b), rs.randint(0, ni, b)
        if scheme == "original":
            uid, cid, y = so.assemble_original(np.stack([pu, pi, np.ones(b, dtype=int)], 1), rs.randint(0, ni, b * k), loss)
            want, w32 = o64.step_original(uid, cid, y), o32.step_original(uid, cid, y)
            got = eng.step_original(eng.ids(uid[:b]), eng.ids(cid)).item()
        elif scheme == "neg_shared":
            if len(np.unique(pi)) < 2:
                continue
            want, w32 = o64.step_neg_shared(pu, pi), o32.step_neg_shared(pu, pi)
            got = eng.step_neg_shared(eng.ids(pu), eng.ids(pi)).item()
        else:
            pi = np.repeat(pi[:b // 4], 4)
            if len(np.unique(pi)) < 2:
                continue
            want, w32 = o64.step_group_neg_shared(pu, pi), o32.step_group_neg_shared(pu, pi)
            got = eng.step_group_neg_shared(eng.ids(pu), eng.ids(pi)).item()
        assert abs(got - want) <= RTOL_LOSS * max(1.0, abs(want)) + 4 * abs(w32 - want), (step, info, got, want)
        far = lambda a32, a64: 4 * float(np.abs(np.asarray(a32, dtype=np.float64) - a64).max())   # noqa: E731
        base = 5e-6 * max(1.0, 1.0 / (0.5 * np.sqrt(d / 3.0))) * max(1.0, b * (1 + k) / (512.0 * min(nu, ni)))
        for got_t, w64, w32_t in ((eng.U, o64.U, o32.U), (eng.V, o64.V, o32.V)):
            g = got_t.cpu().numpy()
            bad = np.abs(g - w64) > base + far(w32_t, w64) + 2e-5 * np.abs(w64)
            if opt != "sgd":
                # Adam's step is lr g / (sqrt(v) + 1e-8): an element whose gradient is itself rounding noise (|g| ~ 1e-8) moves by a
                # fraction of lr that no two summation orders agree on -- 1 element of 2e5 in 3 of 1,500 sweep cases; a hinge at its
                # kink (max-margin) does the same to one row: allow a row's worth, each within lr (+ the float32 oracle's own distance)
                assert bad.sum() <= max(d, int(1e-5 * bad.size)) and np.abs(g - w64).max() <= conf.learn_rate + far(w32_t, w64), \
                    ((step,) + info, int(bad.sum()), float(np.abs(g - w64).max()))
            else:
                assert not bad.any(), ((step,) + info, int(bad.sum()), float(np.abs(g - w64).max()))
        if bias:
            np.testing.assert_allclose(eng.bias["ub"].cpu().numpy(), o64.ub, rtol=2e-5, atol=base + far(o32.ub, o64.ub), err_msg=str((step,) + info))
            np.testing.assert_allclose(eng.bias["cb"].cpu().numpy(), o64.cb, rtol=2e-5, atol=base + far(o32.cb, o64.cb), err_msg=str((step,) + info))
