"""GPU: pairwise losses (log-loss, max-margin) of the shared-negative schemes through the strip form
(csrc/step_pair.hip: anchors -> wide strips -> commit; the path of batch_size_p > 2,048 where the slab-based
tile + apply pair does not exist) against the numpy oracle, and against the tile + apply pair at sizes both run --
duplicates on both sides, list overflow, ragged strips and stages, loss-only mode, Adam, the epoch runner,
b = 4,096 / 8,192.  Reference: utils/objectives.py:90-98,175,194-201."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import torch  # noqa: E402

from oracle import step_oracle as so  # noqa: E402
from nncf_amd.engine import Engine  # noqa: E402

GAMMA = {"max-margin": 0.1, "log-loss": 10.}


def _engine(path, *args, **kw):
    old = os.environ.get("NNCF_SHARED_PATH")
    os.environ["NNCF_SHARED_PATH"] = path
    try:
        return Engine(*args, **kw)
    finally:
        if old is None:
            del os.environ["NNCF_SHARED_PATH"]
        else:
            os.environ["NNCF_SHARED_PATH"] = old


def _mk(loss, b, d, nu, ni, model, path, seed=0, lr=0.05, u_reg=1e-3):
    conf = so.OracleConf(batch_size_p=b, num_negatives=0, loss=loss, loss_gamma=GAMMA[loss], u_reg=u_reg, learn_rate=lr,
                         model_choice=model)
    eng = _engine(path, nu, ni, d, b, 0, loss, conf.loss_gamma, conf.neg_loss_weight, conf.u_reg, conf.learn_rate,
                  conf.normalize_user, conf.normalize_item, seed=seed, init_scale=0.5)
    orc = so.StepOracle(eng.U.cpu().numpy(), eng.V.cpu().numpy(), conf, dtype=np.float64)
    return conf, eng, orc


def _batch(rs, b, nu, ni, group):
    uid = rs.randint(0, nu, b)
    uid[:6] = uid[0]                      # one user six times
    if group:
        cid = np.repeat(np.minimum((ni * rs.power(0.4, b // 4)).astype(np.int64), ni - 1), 4)
    else:
        cid = np.minimum((ni * rs.power(0.4, b)).astype(np.int64), ni - 1)   # popular items repeat
    return uid, cid


@pytest.fixture
def force_row_tiles():
    old = os.environ.get("NNCF_FUSED_MT")

    def set_(mt):
        os.environ["NNCF_FUSED_MT"] = str(mt)
    yield set_
    if old is None:
        os.environ.pop("NNCF_FUSED_MT", None)
    else:
        os.environ["NNCF_FUSED_MT"] = old


@pytest.mark.parametrize("mt", [2, 4])
@pytest.mark.parametrize("loss", ["log-loss", "max-margin"])
@pytest.mark.parametrize("scheme", ["neg_shared", "group_neg_shared"])
@pytest.mark.parametrize("model", ["mf", "content"])
@pytest.mark.parametrize("shape", [(64, 20, 50, 300), (512, 100, 700, 600), (288, 128, 3000, 90), (32, 4, 10, 12)])
def test_pair_strips_match_oracle_and_tile_path(force_row_tiles, mt, loss, scheme, model, shape):
    """ragged last strips (288 = 4 x 64 + 32), ragged stages, d = 4 / 20 / 100 / 128, duplicates on both sides;
    'content' normalises the item rows too"""
    b, d, nu, ni = shape
    group = scheme == "group_neg_shared"
    force_row_tiles(mt)
    conf, eng, orc = _mk(loss, b, d, nu, ni, model, "pairwide", seed=2)
    _, eng_t, _ = _mk(loss, b, d, nu, ni, model, "tile", seed=2)
    rs = np.random.RandomState(3)
    for step in range(3):
        uid, cid = _batch(rs, b, nu, ni, group)
        if group:
            want = orc.step_group_neg_shared(uid, cid)
            got = eng.step_group_neg_shared(eng.ids(uid), eng.ids(cid)).item()
            ref = eng_t.step_group_neg_shared(eng_t.ids(uid), eng_t.ids(cid)).item()
        else:
            want = orc.step_neg_shared(uid, cid)
            got = eng.step_neg_shared(eng.ids(uid), eng.ids(cid)).item()
            ref = eng_t.step_neg_shared(eng_t.ids(uid), eng_t.ids(cid)).item()
        assert abs(got - want) <= 1e-5 * max(1.0, abs(want)), (step, got, want)
        assert abs(got - ref) <= 1e-5 * max(1.0, abs(ref)), (step, got, ref)
        np.testing.assert_allclose(eng.U.cpu().numpy(), orc.U, rtol=1e-5, atol=5e-6)
        np.testing.assert_allclose(eng.V.cpu().numpy(), orc.V, rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(eng_t.U.cpu().numpy(), orc.U, rtol=1e-5, atol=5e-6)
        np.testing.assert_allclose(eng_t.V.cpu().numpy(), orc.V, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("mt", [2, 4])
@pytest.mark.parametrize("loss", ["log-loss", "max-margin"])
@pytest.mark.parametrize("scheme", ["neg_shared", "group_neg_shared"])
def test_pair_strips_list_overflow_and_loss_only(force_row_tiles, mt, loss, scheme):
    """one user in 300 of 512 positions (group: its list of (positive column, anchor) does not fit LDS) / four items
    in 128 positions each (neg_shared: the item lists do not fit); both sides normalised; update = 0 leaves the tables
    alone and returns the same loss"""
    force_row_tiles(mt)
    b, d, nu, ni = 512, 24, 100, 400
    group = scheme == "group_neg_shared"
    conf = so.OracleConf(batch_size_p=b, num_negatives=0, loss=loss, loss_gamma=GAMMA[loss], u_reg=1e-3, learn_rate=0.05,
                         model_choice="basic_embedding", emb_normalization=True)
    eng = _engine("pairwide", nu, ni, d, b, 0, loss, conf.loss_gamma, conf.neg_loss_weight, conf.u_reg, conf.learn_rate,
                  True, True, seed=5, init_scale=0.5)
    orc = so.StepOracle(eng.U.cpu().numpy(), eng.V.cpu().numpy(), conf, dtype=np.float64)
    rs = np.random.RandomState(1)
    uid = rs.randint(0, nu, b)
    if group:
        uid[rs.choice(b, 300, replace=False)] = 17
        cid = np.repeat(rs.choice(ni, b // 4, replace=False), 4)
    else:
        uid[rs.choice(b, 200, replace=False)] = 17
        cid = rs.choice(rs.choice(ni, 4, replace=False), b)
    f = eng.step_group_neg_shared if group else eng.step_neg_shared
    fo = orc.step_group_neg_shared if group else orc.step_neg_shared
    U0, V0 = eng.U.clone(), eng.V.clone()
    l0 = f(eng.ids(uid), eng.ids(cid), update=False).item()
    assert torch.equal(eng.U, U0) and torch.equal(eng.V, V0)
    want = fo(uid, cid)
    got = f(eng.ids(uid), eng.ids(cid)).item()
    assert abs(got - want) <= 1e-5 * max(1.0, abs(want)) and abs(l0 - want) <= 1e-5 * max(1.0, abs(want)), (got, l0, want)
    np.testing.assert_allclose(eng.U.cpu().numpy(), orc.U, rtol=2e-5, atol=4e-6)
    np.testing.assert_allclose(eng.V.cpu().numpy(), orc.V, rtol=2e-5, atol=4e-6)


@pytest.mark.parametrize("scheme", ["neg_shared", "group_neg_shared"])
@pytest.mark.parametrize("b,d,loss", [(4096, 128, "log-loss"), (8192, 64, "max-margin"), (8192, 128, "log-loss")])
def test_pairwise_large_batch(scheme, b, d, loss):
    """pairwise losses above b = 2,048 (no tile + apply workspace there): one step against the
    float64 oracle, default path selection (no environment switch)"""
    nu, ni = 3 * b, 2 * b
    group = scheme == "group_neg_shared"
    conf = so.OracleConf(batch_size_p=b, num_negatives=0, loss=loss, loss_gamma=GAMMA[loss], u_reg=1e-3, learn_rate=0.05,
                         model_choice="mf")
    eng = Engine(nu, ni, d, b, 0, loss, conf.loss_gamma, conf.neg_loss_weight, conf.u_reg, conf.learn_rate,
                 conf.normalize_user, conf.normalize_item, seed=6, init_scale=0.5)
    orc = so.StepOracle(eng.U.cpu().numpy(), eng.V.cpu().numpy(), conf, dtype=np.float64)
    rs = np.random.RandomState(2)
    uid = rs.randint(0, nu, b)
    if group:
        cid = np.repeat(np.minimum((ni * rs.power(0.6, b // 4)).astype(np.int64), ni - 1), 4)
        want = orc.step_group_neg_shared(uid, cid)
        got = eng.step_group_neg_shared(eng.ids(uid), eng.ids(cid)).item()
    else:
        cid = np.minimum((ni * rs.power(0.6, b)).astype(np.int64), ni - 1)
        want = orc.step_neg_shared(uid, cid)
        got = eng.step_neg_shared(eng.ids(uid), eng.ids(cid)).item()
    assert abs(got - want) <= 1e-5 * max(1.0, abs(want)), (got, want)
    np.testing.assert_allclose(eng.U.cpu().numpy(), orc.U, rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(eng.V.cpu().numpy(), orc.V, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("kind", ["adam", "sparse_adam"])
@pytest.mark.parametrize("scheme", ["neg_shared", "group_neg_shared"])
def test_pair_strips_adam_equals_tile_path(force_row_tiles, kind, scheme):
    """Keras (dense) Adam and the lazy-sparse Adam ('sum' duplicates) through the commit kernel: the same rows as the
    tile + apply pair produces, three steps"""
    force_row_tiles(2)
    b, d, nu, ni = 128, 40, 300, 200
    group = scheme == "group_neg_shared"
    engs = []
    for path in ("pairwide", "tile"):
        e = _engine(path, nu, ni, d, b, 0, "log-loss", 10., 128., 1e-3, 0.01, True, False, seed=3, init_scale=0.5)
        e.set_optimizer(kind)
        engs.append(e)
    rs = np.random.RandomState(4)
    for step in range(3):
        uid, cid = _batch(rs, b, nu, ni, group)
        ls = []
        for e in engs:
            f = e.step_group_neg_shared if group else e.step_neg_shared
            ls.append(f(e.ids(uid), e.ids(cid)).item())
        assert abs(ls[0] - ls[1]) <= 1e-5 * max(1.0, abs(ls[1])), (step, ls)
        np.testing.assert_allclose(engs[0].U.cpu().numpy(), engs[1].U.cpu().numpy(), rtol=2e-4, atol=2e-5)
        np.testing.assert_allclose(engs[0].V.cpu().numpy(), engs[1].V.cpu().numpy(), rtol=2e-4, atol=2e-5)


def test_pair_strips_epoch_runner_equals_step_by_step(force_row_tiles):
    force_row_tiles(2)
    b, d, nu, ni, steps = 128, 40, 900, 800, 70     # 70 > one plan chunk of 64
    _, e1, _ = _mk("log-loss", b, d, nu, ni, "mf", "pairwide", seed=4)
    _, e2, _ = _mk("log-loss", b, d, nu, ni, "mf", "pairwide", seed=4)
    rs = np.random.RandomState(8)
    uid, cid = rs.randint(0, nu, steps * b), rs.randint(0, ni, steps * b)
    ua, ca = e1.ids(uid), e1.ids(cid)
    l1 = e1.epoch_neg_shared(ua, ca, steps).cpu().numpy()
    l2 = np.array([e2.step_neg_shared(ua[t * b:(t + 1) * b], ca[t * b:(t + 1) * b]).item() for t in range(steps)])
    assert np.array_equal(l1, l2.astype(np.float32))
    assert torch.equal(e1.U, e2.U) and torch.equal(e1.V, e2.V)


@pytest.mark.timeout(300)
@pytest.mark.parametrize("seed", range(int(os.environ.get("NNCF_SOAK_SEEDS", "16"))))
def test_pair_strips_random_shapes_terminate_and_match(force_row_tiles, seed):
    """Random small shapes through the strip form of the pairwise losses -- tiny catalogues (most strips empty), one-user
    batches, both schemes and losses, 'mf' and 'content', 1 / 2 / 4 row tiles, row scales from 1e-3 to 30 (the large-batch
    kernels split their operands into two fp16 planes: devutil.h split_h) -- two steps each against the float64 oracle.
    NNCF_SOAK_SEEDS=<n> widens the sweep (tools/soak.sh)."""
    rs = np.random.RandomState(7000 + seed)
    b = int(rs.choice([32, 64, 96, 128, 160, 256, 512]))
    d = int(rs.choice([4, 20, 64, 100, 128]))
    nu = int(rs.choice([1, 2, 7, 40, 400, 3000]))
    ni = int(rs.choice([2, 5, 33, 300, 2000]))
    group = bool(rs.randint(2))
    loss = str(rs.choice(["log-loss", "max-margin"]))
    model = str(rs.choice(["mf", "content"]))
    scale = float(rs.choice([1e-3, 0.1, 1.0, 1.0, 30.0]))
    if model == "content":
        scale = max(scale, 0.1)   # the backward of x / |x| multiplies fp32's rounding by 1 / |x|: rows of 1e-3 leave nothing to compare
    force_row_tiles(int(rs.choice([1, 2, 4])))
    conf, eng, orc = _mk(loss, b, d, nu, ni, model, "pairwide", seed=seed)
    if scale != 1.0:
        eng.U.mul_(scale)
        eng.V.mul_(scale)
        orc.U *= scale
        orc.V *= scale
    for step in range(2):
        uid = rs.randint(0, nu, b)
        cid = np.repeat(rs.randint(0, ni, b // 4), 4) if group else rs.randint(0, ni, b)
        if len(np.unique(cid)) < 2:          # the pairwise losses compare a positive with the OTHER items' scores
            continue
        if group:
            want = orc.step_group_neg_shared(uid, cid)
            got = eng.step_group_neg_shared(eng.ids(uid), eng.ids(cid)).item()
        else:
            want = orc.step_neg_shared(uid, cid)
            got = eng.step_neg_shared(eng.ids(uid), eng.ids(cid)).item()
        assert abs(got - want) <= 1e-5 * max(1.0, abs(want)), (step, got, want)
        # absolute term: 1e-5 of the rows' scale, times 1 / |x| of the smallest normalised rows (|x| ~ 0.5 scale sqrt(d / 3) at init)
        sc = max(1.0, scale) * max(1.0, 1.0 / (0.5 * scale * np.sqrt(d / 3.0)))
        np.testing.assert_allclose(eng.U.cpu().numpy(), orc.U, rtol=2e-5, atol=1e-5 * sc)
        np.testing.assert_allclose(eng.V.cpu().numpy(), orc.V, rtol=2e-5, atol=1e-5 * sc)


@pytest.mark.timeout(600)
@pytest.mark.parametrize("seed", range(int(os.environ.get("NNCF_SOAK_LARGE", "4"))))
def test_large_batch_random_shapes_match(seed):
    """The large-batch regime on its default paths (b > 2,048: fused_wide / pair_wide, products on the f16 pipe): random b (ragged
    strips and stages included), d, loss (all four), scheme, 'mf' / 'content', row scale, and catalogues from 5 rows (every row a
    heavy duplicate: multiplicities in the hundreds) to 3 b -- one step against the float64 oracle.  NNCF_SOAK_LARGE widens it."""
    rs = np.random.RandomState(15000 + seed)
    b = int(rs.choice([2080, 2560, 4096, 4128, 5120, 8192]))
    d = int(rs.choice([16, 32, 64, 100, 128]))
    loss = str(rs.choice(["skip-gram", "mse", "log-loss", "max-margin"]))
    group = bool(rs.randint(2))
    model = str(rs.choice(["mf", "content"]))
    nu = int(rs.choice([5, 300, 3 * b]))
    ni = int(rs.choice([5, 300, 2 * b]))
    scale = float(rs.choice([0.1, 1.0, 1.0, 4.0])) if loss != "mse" else 1.0
    gamma = GAMMA.get(loss, 10.)
    conf = so.OracleConf(batch_size_p=b, num_negatives=0, loss=loss, loss_gamma=gamma, u_reg=1e-3, learn_rate=0.05, model_choice=model,
                         neg_loss_weight=8. if loss == "mse" else 128.)
    eng = Engine(nu, ni, d, b, 0, loss, conf.loss_gamma, conf.neg_loss_weight, conf.u_reg, conf.learn_rate,
                 conf.normalize_user, conf.normalize_item, seed=seed, init_scale=0.5)
    eng.U.mul_(scale)
    eng.V.mul_(scale)
    orc = so.StepOracle(eng.U.cpu().numpy(), eng.V.cpu().numpy(), conf, dtype=np.float64)
    uid = rs.randint(0, nu, b)
    cid = np.repeat(rs.randint(0, ni, b // 4), 4) if group else rs.randint(0, ni, b)
    info = (b, d, loss, group, model, nu, ni, scale)
    if group:
        want = orc.step_group_neg_shared(uid, cid)
        got = eng.step_group_neg_shared(eng.ids(uid), eng.ids(cid)).item()
    else:
        want = orc.step_neg_shared(uid, cid)
        got = eng.step_neg_shared(eng.ids(uid), eng.ids(cid)).item()
    assert abs(got - want) <= 1e-5 * max(1.0, abs(want)), (info, got, want)
    # absolute term: 1e-5 of the rows' scale; x / |x|'s backward multiplies fp32's rounding by 1 / |x|; b / min(nu, ni) terms meet in a row
    amp = max(1.0, scale) * max(1.0, 1.0 / (0.5 * scale * np.sqrt(d / 3.0))) * max(1.0, b / (512.0 * min(nu, ni)))
    np.testing.assert_allclose(eng.U.cpu().numpy(), orc.U, rtol=2e-5, atol=1e-5 * amp, err_msg=str(info))
    np.testing.assert_allclose(eng.V.cpu().numpy(), orc.V, rtol=2e-5, atol=1e-5 * amp, err_msg=str(info))


@pytest.mark.parametrize("case", ["gamma_times_multiplicity", "large_rows"])
def test_pair_strips_leave_the_fp16_planes_outside_their_domain(force_row_tiles, monkeypatch, case):
    """the f16-plane strips store dL/dS = (c - m) gamma sigmoid unscaled -- multiplicity x loss_gamma beyond 65,504
    (a one-user batch with a large gamma) turned the tables NaN behind a finite loss; and un-normalised rows beyond fp16's range
    never fitted the planes.  Round 5: pair_anchor_kernel takes the range of the batch's rows and the largest multiplicity, and
    the step runs on the fp32-pipe twin when the planes would not be exact -- it equals the float64 oracle either way."""
    monkeypatch.delenv("NNCF_WIDE_F16SPLIT", raising=False)
    force_row_tiles(2)
    b, d, nu, ni = 256, 20, 40, 300
    rs = np.random.RandomState(9)
    if case == "gamma_times_multiplicity":
        GAMMA["log-loss"] = 300.          # 256 positions of ONE user x gamma = 76,800
        try:
            conf, eng, orc = _mk("log-loss", b, d, nu, ni, "mf", "pairwide", seed=2, lr=0.01)
        finally:
            GAMMA["log-loss"] = 10.
        uid = np.zeros(b, dtype=np.int64)
        cid = rs.randint(0, ni, b)
    else:   # un-normalised rows (emb_normalization off), one of them of magnitude 10^6
        conf = so.OracleConf(batch_size_p=b, num_negatives=0, loss="max-margin", loss_gamma=GAMMA["max-margin"], u_reg=1e-3,
                             learn_rate=0.01, model_choice="mf", emb_normalization=False)
        assert not conf.normalize_user and not conf.normalize_item
        eng = _engine("pairwide", nu, ni, d, b, 0, "max-margin", conf.loss_gamma, conf.neg_loss_weight, conf.u_reg, conf.learn_rate,
                      False, False, seed=2, init_scale=0.5)
        uid, cid = _batch(rs, b, nu, ni, False)
        eng.U[uid[0], :] = 1e6
        orc = so.StepOracle(eng.U.cpu().numpy(), eng.V.cpu().numpy(), conf, dtype=np.float64)
    want = orc.step_neg_shared(uid, cid)
    got = eng.step_neg_shared(eng.ids(uid), eng.ids(cid)).item()
    assert np.isfinite(got) and abs(got - want) <= 1e-5 * max(1.0, abs(want)), (got, want)
    tu, tv = np.unique(uid), np.unique(cid)
    U, V = eng.U.cpu().numpy(), eng.V.cpu().numpy()
    assert np.isfinite(U).all() and np.isfinite(V).all()
    np.testing.assert_allclose(U[tu], orc.U[tu], rtol=2e-5, atol=5e-6)
    np.testing.assert_allclose(V[tv], orc.V[tv], rtol=2e-5, atol=5e-6)
