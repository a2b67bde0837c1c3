"""GPU: the one-launch shared-negative step (csrc/step_fused.hip: U-pass + V-pass strips, pointwise
losses) against the numpy oracle, and against the two-launch tile + apply path it replaces
(NNCF_SHARED_PATH=tile) -- duplicates on both sides, ragged stage counts, loss-only mode, key-list
overflow."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import torch  # noqa: E402

from oracle import step_oracle as so  # noqa: E402
from nncf_amd.engine import Engine  # noqa: E402


def _mk(loss, b, d, nu, ni, model, path, seed=0, lr=0.05, u_reg=1e-3):
    conf = so.OracleConf(batch_size_p=b, num_negatives=0, loss=loss, u_reg=u_reg, learn_rate=lr, model_choice=model,
                         neg_loss_weight=8. if loss == "mse" else 128.)
    old = os.environ.get("NNCF_SHARED_PATH")
    os.environ["NNCF_SHARED_PATH"] = path
    try:
        eng = Engine(nu, ni, d, b, 0, loss, conf.loss_gamma, conf.neg_loss_weight, conf.u_reg, conf.learn_rate,
                     conf.normalize_user, conf.normalize_item, seed=seed, init_scale=0.5)
    finally:
        if old is None:
            del os.environ["NNCF_SHARED_PATH"]
        else:
            os.environ["NNCF_SHARED_PATH"] = old
    # float64 oracle: with an item in ~40 of 512 positions a float32 reference is itself 5e-6 off
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


@pytest.mark.parametrize("loss", ["skip-gram", "mse"])
@pytest.mark.parametrize("scheme", ["neg_shared", "group_neg_shared"])
@pytest.mark.parametrize("shape", [(64, 20, 50, 300), (512, 100, 700, 600), (256, 128, 3000, 90), (32, 4, 10, 12)])
def test_fused_step_matches_oracle_and_tile_path(loss, scheme, shape):
    b, d, nu, ni = shape
    group = scheme == "group_neg_shared"
    conf, eng, orc = _mk(loss, b, d, nu, ni, "mf", "fused", seed=2)
    _, eng_t, _ = _mk(loss, b, d, nu, ni, "mf", "tile", seed=2)
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


@pytest.mark.parametrize("scheme", ["neg_shared", "group_neg_shared"])
def test_fused_content_normalisation_and_loss_only(scheme):
    """item-side L2 normalisation (content models) + update = 0 leaves the tables alone"""
    b, d, nu, ni = 64, 24, 40, 80
    group = scheme == "group_neg_shared"
    conf = so.OracleConf(batch_size_p=b, num_negatives=0, loss="skip-gram", u_reg=1e-3, learn_rate=0.05,
                         model_choice="basic_embedding", emb_normalization=True)
    eng = Engine(nu, ni, d, b, 0, "skip-gram", conf.loss_gamma, conf.neg_loss_weight, conf.u_reg, conf.learn_rate, True, True,
                 seed=5, init_scale=0.5)
    orc = so.StepOracle(eng.U.cpu().numpy(), eng.V.cpu().numpy(), conf, dtype=np.float32)
    rs = np.random.RandomState(9)
    uid, cid = _batch(rs, b, nu, ni, group)
    U0, V0 = eng.U.clone(), eng.V.clone()
    f = eng.step_group_neg_shared if group else eng.step_neg_shared
    fo = orc.step_group_neg_shared if group else orc.step_neg_shared
    l0 = f(eng.ids(uid), eng.ids(cid), update=False).item()
    assert torch.equal(eng.U, U0) and torch.equal(eng.V, V0)
    want = fo(uid, cid)
    got = f(eng.ids(uid), eng.ids(cid)).item()
    assert abs(got - want) <= 1e-5 * max(1.0, abs(want)) and abs(l0 - want) <= 1e-5 * max(1.0, abs(want))
    np.testing.assert_allclose(eng.U.cpu().numpy(), orc.U, rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(eng.V.cpu().numpy(), orc.V, rtol=1e-5, atol=2e-6)


def test_fused_group_key_list_overflow():
    """group_neg_shared, one user in 300 of 512 positions: its positive-column list does not fit the
    workgroup's LDS key buffer and is read from the plan instead"""
    b, d, nu, ni = 512, 20, 100, 400
    conf, eng, orc = _mk("skip-gram", b, d, nu, ni, "mf", "fused", seed=7)
    rs = np.random.RandomState(1)
    uid = rs.randint(0, nu, b)
    uid[rs.choice(b, 300, replace=False)] = 17
    cid = np.repeat(rs.choice(ni, b // 4, replace=False), 4)
    want = orc.step_group_neg_shared(uid, cid)
    got = eng.step_group_neg_shared(eng.ids(uid), eng.ids(cid)).item()
    assert abs(got - want) <= 1e-5 * max(1.0, abs(want)), (got, want)
    np.testing.assert_allclose(eng.U.cpu().numpy(), orc.U, rtol=2e-5, atol=4e-6)
    np.testing.assert_allclose(eng.V.cpu().numpy(), orc.V, rtol=1e-5, atol=2e-6)


def test_fused_epoch_runner_equals_step_by_step():
    b, d, nu, ni, steps = 128, 40, 900, 800, 70     # 70 > one plan chunk of 64
    _, e1, _ = _mk("skip-gram", b, d, nu, ni, "mf", "fused", seed=4)
    _, e2, _ = _mk("skip-gram", b, d, nu, ni, "mf", "fused", seed=4)
    rs = np.random.RandomState(8)
    uid, cid = rs.randint(0, nu, steps * b), rs.randint(0, ni, steps * b)
    ua, ca = e1.ids(uid), e1.ids(cid)
    l1 = e1.epoch_neg_shared(ua, ca, steps).cpu().numpy()
    l2 = np.array([e2.step_neg_shared(ua[t * b:(t + 1) * b], ca[t * b:(t + 1) * b]).item() for t in range(steps)])
    assert np.array_equal(l1, l2.astype(np.float32))
    assert torch.equal(e1.U, e2.U) and torch.equal(e1.V, e2.V)


@pytest.mark.parametrize("b,d", [(4096, 128), (8192, 64)])
def test_fused_large_batch(b, d):
    """config 4's large-batch regime (b >= 8K per GPU): 2 x b/16 strip workgroups, no split, no
    O(b^2) workspace; one step against the float64 oracle (the tile + apply path does not exist above b = 2048)."""
    nu, ni = 3 * b, 2 * b
    conf, eng, orc = _mk("skip-gram", b, d, nu, ni, "mf", "fused", seed=6, lr=0.05)
    rs = np.random.RandomState(2)
    uid = rs.randint(0, nu, b)
    cid = np.minimum((ni * rs.power(0.6, b)).astype(np.int64), ni - 1)
    want = orc.step_neg_shared(uid, cid)
    got = eng.step_neg_shared(eng.ids(uid), eng.ids(cid)).item()
    assert abs(got - want) <= 1e-5 * max(1.0, abs(want)), (got, want)
    np.testing.assert_allclose(eng.U.cpu().numpy(), orc.U, rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(eng.V.cpu().numpy(), orc.V, rtol=1e-5, atol=1e-5)
    # (pairwise losses at these sizes: tests/test_pair_gpu.py::test_pairwise_large_batch)


@pytest.mark.parametrize("scale", [1.0, 300.0, 1e-4])
def test_wide_kernel_on_the_fp16_pipe_is_as_close_to_float64_as_on_the_fp32_pipe(monkeypatch, scale):
    """Round 4: the large-batch kernels form their products on the fp16 matrix pipe -- x = hi + lo / 2048, two fp16 planes, three
    MFMAs per 32 k-values, the cross terms in their own accumulator (devutil.h split_h; NNCF_WIDE_F16SPLIT=0: the fp32 pipe).
    Same step, both pipes, against the float64 oracle: the loss to 1e-5, the tables to the same tolerance, and the split's error
    no more than twice the fp32 pipe's (measured: smaller) -- with rows of magnitude 1, 300 (fp16 would overflow at 65,504: the
    dot products reach 1e6, in fp32 accumulators) and 1e-4 (below fp16's normal range: the planes keep an absolute precision)."""
    b, d = 4096, 128
    nu, ni = 2 * b, b
    errs = {}
    for pipe in ("1", "0"):
        monkeypatch.setenv("NNCF_WIDE_F16SPLIT", pipe)
        conf, eng, orc = _mk("skip-gram", b, d, nu, ni, "mf", "fused", seed=6, lr=0.05)
        if scale != 1.0:
            eng.U.mul_(scale)
            eng.V.mul_(1.0 / scale if scale > 1 else scale)     # scores stay O(1) at scale 300; both sides tiny at 1e-4
            orc = so.StepOracle(eng.U.cpu().numpy(), eng.V.cpu().numpy(), conf, dtype=np.float64)
        U0 = eng.U.cpu().numpy().copy()
        rs = np.random.RandomState(2)
        uid = rs.randint(0, nu, b)
        cid = np.minimum((ni * rs.power(0.6, b)).astype(np.int64), ni - 1)
        want = orc.step_neg_shared(uid, cid)
        got = eng.step_neg_shared(eng.ids(uid), eng.ids(cid)).item()
        assert abs(got - want) <= 1e-5 * max(1.0, abs(want)), (pipe, got, want)
        step_u = np.abs(orc.U - U0).max()
        errs[pipe] = (np.abs(eng.U.cpu().numpy() - orc.U).max() / step_u, np.abs(eng.V.cpu().numpy() - orc.V).max())
        np.testing.assert_allclose(eng.U.cpu().numpy(), orc.U, rtol=1e-5, atol=1e-5 * max(1.0, scale))
        np.testing.assert_allclose(eng.V.cpu().numpy(), orc.V, rtol=1e-5, atol=1e-5)
    assert errs["1"][0] <= 2 * errs["0"][0] + 1e-7, errs      # error of the step relative to the largest step, fp16 split vs fp32 pipe


def test_wide_kernel_leaves_the_fp16_planes_outside_their_domain(monkeypatch):
    """Rows of magnitude 10^6 do not fit fp16's hi plane.  Round 4's default returned a NaN loss there (loud, but a
    deviation: the reference's fp32 keeps going); round 5 decides per step ON THE DEVICE (wide_domain_kernel over the
    batch's rows -> fused_wide_kernel<H16> or its fp32-pipe twin): the step of such a batch equals the float64 oracle,
    and the step of an ordinary batch right after it is back on the f16 planes.  NNCF_WIDE_F16SPLIT=force (the planes
    without the guard) still shows what the guard prevents."""
    b, d = 4096, 32
    rs = np.random.RandomState(2)
    uid = rs.randint(0, 2 * b, b)
    uid[0] = 0
    cid = rs.randint(0, b, b)
    for mode, finite in ((None, True), ("force", False)):
        if mode is None:
            monkeypatch.delenv("NNCF_WIDE_F16SPLIT", raising=False)
        else:
            monkeypatch.setenv("NNCF_WIDE_F16SPLIT", mode)
        conf, eng, orc = _mk("skip-gram", b, d, 2 * b, b, "mf", "fused", seed=6, lr=0.05)
        eng.U[0, :] = 1e6 * torch.sign(eng.U[0, :])      # one user row of magnitude 10^6
        eng.V[cid[5], :3] = 2e5                         # and an item row beyond 65,504
        orc.U[0, :] = eng.U[0, :].cpu().numpy()
        orc.V[cid[5], :3] = 2e5
        want = orc.step_neg_shared(uid, cid)
        got = eng.step_neg_shared(eng.ids(uid), eng.ids(cid)).item()
        if not finite:
            assert got != got or abs(got) == float("inf")
            continue
        assert abs(got - want) <= 1e-5 * max(1.0, abs(want)), (got, want)
        tu, tv = np.unique(uid), np.unique(cid)
        np.testing.assert_allclose(eng.U.cpu().numpy()[tu], orc.U[tu], rtol=2e-5, atol=5e-6)
        np.testing.assert_allclose(eng.V.cpu().numpy()[tv], orc.V[tv], rtol=2e-5, atol=5e-6)
        # an ordinary batch on the same engine: users / items away from the two large rows
        uid2 = rs.randint(1, 2 * b, b)
        cid2 = np.where(cid == cid[5], (cid[5] + 1) % b, cid)
        want2 = orc.step_neg_shared(uid2, cid2)
        got2 = eng.step_neg_shared(eng.ids(uid2), eng.ids(cid2)).item()
        assert abs(got2 - want2) <= 1e-5 * max(1.0, abs(want2)), (got2, want2)


def test_wide_kernel_mse_gradient_beyond_fp16_takes_the_fp32_pipe(monkeypatch):
    """mse's dL/dS = 2 w (s - y) grows with the score (found by the random-shape sweep, tools/soak.sh: a diverging mse run, rows of
    ~100 after one step): past fp16's range the dL/dS plane would hold inf.  The device-side bound 2 w c_max (d |u|max |v|max + 1)
    sends such a step to the fp32 pipe: loss and tables equal the oracle's in the default mode as with NNCF_WIDE_F16SPLIT=0; the
    unguarded planes (=force) return NaN."""
    b, d, nu, ni = 128, 20, 1, 2
    rs = np.random.RandomState(4)
    uid, cid = rs.randint(0, nu, b), rs.randint(0, ni, b)
    monkeypatch.setenv("NNCF_FUSED_MT", "2")
    for split, finite in ((None, True), ("0", True), ("force", False)):
        if split is None:
            monkeypatch.delenv("NNCF_WIDE_F16SPLIT", raising=False)
        else:
            monkeypatch.setenv("NNCF_WIDE_F16SPLIT", split)
        conf, eng, orc = _mk("mse", b, d, nu, ni, "mf", "fused", seed=3)
        eng.U.mul_(200.)
        eng.V.mul_(200.)
        orc.U *= 200.
        orc.V *= 200.
        want = orc.step_neg_shared(uid, cid)
        got = eng.step_neg_shared(eng.ids(uid), eng.ids(cid)).item()
        if finite:
            assert abs(got - want) <= 1e-5 * abs(want), (got, want)
            np.testing.assert_allclose(eng.U.cpu().numpy(), orc.U, rtol=1e-4)
        else:
            assert got != got, got


@pytest.fixture
def force_row_tiles():
    """NNCF_FUSED_MT (read per launch by fused_row_tiles): run the large-batch kernel at test sizes"""
    old = os.environ.get("NNCF_FUSED_MT")

    def set_(mt):
        os.environ["NNCF_FUSED_MT"] = str(mt)
    yield set_
    if old is None:
        os.environ.pop("NNCF_FUSED_MT", None)
    else:
        os.environ["NNCF_FUSED_MT"] = old


@pytest.mark.parametrize("mt", [2, 4])
@pytest.mark.parametrize("loss", ["skip-gram", "mse"])
@pytest.mark.parametrize("scheme", ["neg_shared", "group_neg_shared"])
@pytest.mark.parametrize("shape", [(64, 20, 50, 300), (512, 100, 700, 600), (288, 128, 3000, 90)])
def test_wide_kernel_matches_oracle(force_row_tiles, mt, loss, scheme, shape):
    """fused_wide_kernel (16 MT own rows per workgroup; the path of b >= 4096) forced at small sizes: ragged last
    strips (288 = 4 x 64 + 32), ragged stages, d = 20 / 100 / 128 (generic, 7 and 8 column groups), duplicates."""
    b, d, nu, ni = shape
    group = scheme == "group_neg_shared"
    force_row_tiles(mt)
    conf, eng, orc = _mk(loss, b, d, nu, ni, "mf", "fused", seed=2)
    rs = np.random.RandomState(3)
    for step in range(3):
        uid, cid = _batch(rs, b, nu, ni, group)
        if group:
            want = orc.step_group_neg_shared(uid, cid)
            got = eng.step_group_neg_shared(eng.ids(uid), eng.ids(cid)).item()
        else:
            want = orc.step_neg_shared(uid, cid)
            got = eng.step_neg_shared(eng.ids(uid), eng.ids(cid)).item()
        assert abs(got - want) <= 1e-5 * max(1.0, abs(want)), (step, got, want)
        np.testing.assert_allclose(eng.U.cpu().numpy(), orc.U, rtol=1e-5, atol=5e-6)
        np.testing.assert_allclose(eng.V.cpu().numpy(), orc.V, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("mt", [2, 4])
def test_wide_kernel_normalised_and_key_overflow(force_row_tiles, mt):
    """both sides L2-normalised; group scheme with one user in 300 of 512 positions (key list read from the plan
    when it does not fit); loss-only launches leave the tables alone; epoch runner == step by step"""
    force_row_tiles(mt)
    b, d, nu, ni = 512, 24, 100, 400
    conf = so.OracleConf(batch_size_p=b, num_negatives=0, loss="skip-gram", u_reg=1e-3, learn_rate=0.05,
                         model_choice="basic_embedding", emb_normalization=True)
    eng = Engine(nu, ni, d, b, 0, "skip-gram", conf.loss_gamma, conf.neg_loss_weight, conf.u_reg, conf.learn_rate, True, True,
                 seed=5, init_scale=0.5)
    orc = so.StepOracle(eng.U.cpu().numpy(), eng.V.cpu().numpy(), conf, dtype=np.float64)
    rs = np.random.RandomState(1)
    uid = rs.randint(0, nu, b)
    uid[rs.choice(b, 300, replace=False)] = 17
    cid = np.repeat(rs.choice(ni, b // 4, replace=False), 4)
    U0, V0 = eng.U.clone(), eng.V.clone()
    l0 = eng.step_group_neg_shared(eng.ids(uid), eng.ids(cid), update=False).item()
    assert torch.equal(eng.U, U0) and torch.equal(eng.V, V0)
    want = orc.step_group_neg_shared(uid, cid)
    got = eng.step_group_neg_shared(eng.ids(uid), eng.ids(cid)).item()
    assert abs(got - want) <= 1e-5 * max(1.0, abs(want)) and abs(l0 - want) <= 1e-5 * max(1.0, abs(want))
    np.testing.assert_allclose(eng.U.cpu().numpy(), orc.U, rtol=2e-5, atol=4e-6)
    np.testing.assert_allclose(eng.V.cpu().numpy(), orc.V, rtol=1e-5, atol=2e-6)

    steps = 70
    _, e1, _ = _mk("skip-gram", 128, 40, 900, 800, "mf", "fused", seed=4)
    _, e2, _ = _mk("skip-gram", 128, 40, 900, 800, "mf", "fused", seed=4)
    u2, c2 = rs.randint(0, 900, steps * 128), rs.randint(0, 800, steps * 128)
    ua, ca = e1.ids(u2), e1.ids(c2)
    l1 = e1.epoch_neg_shared(ua, ca, steps).cpu().numpy()
    l2 = np.array([e2.step_neg_shared(ua[t * 128:(t + 1) * 128], ca[t * 128:(t + 1) * 128]).item() for t in range(steps)])
    assert np.array_equal(l1, l2.astype(np.float32))
    assert torch.equal(e1.U, e2.U) and torch.equal(e1.V, e2.V)


@pytest.mark.parametrize("b,d", [(512, 64), (2048, 32), (4096, 32), (8192, 16)])
def test_fused_rows_are_read_before_any_is_stepped(b, d):
    """The one-launch step updates table rows IN PLACE while other workgroups of the same launch stream the same
    tables; a strip may only store after every strip has finished reading (FusedArgs::done).  A large SGD rate makes
    a row read after its update visible far above the tolerance; three steps so that launch parity alternates."""
    nu, ni = b // 2, b // 2          # every row is read by many strips and owned by one
    conf, eng, orc = _mk("mse", b, d, nu, ni, "mf", "fused", seed=11, lr=4.0, u_reg=0.)
    rs = np.random.RandomState(5)
    for step in range(3):
        uid, cid = rs.randint(0, nu, b), rs.randint(0, ni, b)
        want = orc.step_neg_shared(uid, cid)
        got = eng.step_neg_shared(eng.ids(uid), eng.ids(cid)).item()
        assert abs(got - want) <= 1e-5 * max(1.0, abs(want)), (step, got, want)
        np.testing.assert_allclose(eng.U.cpu().numpy(), orc.U, rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(eng.V.cpu().numpy(), orc.V, rtol=1e-5, atol=1e-5)


@pytest.fixture
def force_deferred():
    old = os.environ.get("NNCF_FUSED_DEFER")
    os.environ["NNCF_FUSED_DEFER"] = "1"
    yield
    if old is None:
        os.environ.pop("NNCF_FUSED_DEFER", None)
    else:
        os.environ["NNCF_FUSED_DEFER"] = old


@pytest.mark.parametrize("mt", [1, 4])
@pytest.mark.parametrize("scheme", ["neg_shared", "group_neg_shared"])
@pytest.mark.parametrize("optimizer", ["sgd", "sparse_adam"])
def test_deferred_step_stage_and_commit(force_deferred, force_row_tiles, mt, scheme, optimizer):
    """The two-launch form a grid beyond the device's resident capacity takes (b = 16384): gradient rows staged by
    the strip / wide kernel, tables stepped by fused_commit_kernel -- forced here at b = 512 and compared with the
    oracle and, bitwise, with the in-place one-launch form."""
    force_row_tiles(mt)
    b, d, nu, ni = 512, 36, 300, 500
    group = scheme == "group_neg_shared"
    conf = so.OracleConf(batch_size_p=b, num_negatives=0, loss="skip-gram", u_reg=1e-3, learn_rate=0.05, model_choice="mf",
                         optimizer=optimizer)

    def mk():
        e = Engine(nu, ni, d, b, 0, "skip-gram", conf.loss_gamma, conf.neg_loss_weight, conf.u_reg, conf.learn_rate,
                   False, False, seed=3, init_scale=0.5)
        if optimizer != "sgd":
            e.set_optimizer(optimizer)
        return e
    eng = mk()
    orc = so.StepOracle(eng.U.cpu().numpy(), eng.V.cpu().numpy(), conf, dtype=np.float64)
    ref = mk()
    rs = np.random.RandomState(3)
    name = "step_group_neg_shared" if group else "step_neg_shared"
    for step in range(3):
        uid, cid = _batch(rs, b, nu, ni, group)
        want = getattr(orc, name)(uid, cid)
        got = getattr(eng, name)(eng.ids(uid), eng.ids(cid)).item()
        os.environ["NNCF_FUSED_DEFER"] = "0"
        same = getattr(ref, name)(ref.ids(uid), ref.ids(cid)).item()
        os.environ["NNCF_FUSED_DEFER"] = "1"
        assert got == same
        assert torch.equal(eng.U, ref.U) and torch.equal(eng.V, ref.V)
        assert abs(got - want) <= 1e-5 * max(1.0, abs(want)), (step, got, want)
        # Adam divides by sqrt(v): a tiny gradient difference is amplified where v ~ g^2 (test_step_gpu.py)
        tol = dict(rtol=2e-5, atol=1e-5) if optimizer == "sgd" else dict(rtol=2e-4, atol=1e-4)
        np.testing.assert_allclose(eng.U.cpu().numpy(), orc.U, **tol)
        np.testing.assert_allclose(eng.V.cpu().numpy(), orc.V, **tol)


def test_fused_batch_16384():
    """b = 16384 (config 4's upper per-GPU batch): the 16384-entry plan (sort keys in LDS, per-entry
    arrays in global memory), 2 x 256 wide workgroups -- more than the device holds at once, so the step stages
    its gradient rows and a commit launch steps the tables.  float32 oracle (a float64 one would hold 2 GB per b x b
    temporary), large SGD rate: rows read after an update would show."""
    b, d = 16384, 16
    nu, ni = b // 2, b // 2
    conf = so.OracleConf(batch_size_p=b, num_negatives=0, loss="mse", u_reg=0., learn_rate=4.0, model_choice="mf",
                         neg_loss_weight=8.)
    eng = Engine(nu, ni, d, b, 0, "mse", conf.loss_gamma, conf.neg_loss_weight, conf.u_reg, conf.learn_rate, False, False,
                 seed=11, init_scale=0.5)
    orc = so.StepOracle(eng.U.cpu().numpy(), eng.V.cpu().numpy(), conf, dtype=np.float32)
    rs = np.random.RandomState(5)
    for step in range(2):
        uid, cid = rs.randint(0, nu, b), rs.randint(0, ni, b)
        want = orc.step_neg_shared(uid, cid)
        got = eng.step_neg_shared(eng.ids(uid), eng.ids(cid)).item()
        assert abs(got - want) <= 1e-5 * max(1.0, abs(want)), (step, got, want)
        np.testing.assert_allclose(eng.U.cpu().numpy(), orc.U, rtol=1e-4, atol=2e-5)
        np.testing.assert_allclose(eng.V.cpu().numpy(), orc.V, rtol=1e-4, atol=2e-5)


@pytest.mark.timeout(900)
def test_fused_batch_32768():
    """b = 32768 (inside config [3]'s "b >~ 16K per GPU" regime): the 32768-entry plan (sort keys in global
    memory, one 16384-key LDS tile), 2 x 512 wide workgroups, staged rows + commit launch.  One step against the float32
    oracle (4 GB per b x b temporary); many duplicate users / items (tables of b / 4 rows)."""
    b, d = 32768, 16
    nu, ni = b // 4, b // 4
    conf = so.OracleConf(batch_size_p=b, num_negatives=0, loss="mse", u_reg=0., learn_rate=4.0, model_choice="mf",
                         neg_loss_weight=8.)
    eng = Engine(nu, ni, d, b, 0, "mse", conf.loss_gamma, conf.neg_loss_weight, conf.u_reg, conf.learn_rate, False, False,
                 seed=11, init_scale=0.5)
    orc = so.StepOracle(eng.U.cpu().numpy(), eng.V.cpu().numpy(), conf, dtype=np.float32)
    rs = np.random.RandomState(5)
    uid, cid = rs.randint(0, nu, b), rs.randint(0, ni, b)
    want = orc.step_neg_shared(uid, cid)
    got = eng.step_neg_shared(eng.ids(uid), eng.ids(cid)).item()
    # the loss is a mean of 2^30 float32 terms on BOTH sides (the oracle's own float32 sum is good to ~1e-5 here)
    assert abs(got - want) <= 5e-5 * max(1.0, abs(want)), (got, want)
    np.testing.assert_allclose(eng.U.cpu().numpy(), orc.U, rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(eng.V.cpu().numpy(), orc.V, rtol=1e-4, atol=2e-5)


@pytest.mark.timeout(300)
@pytest.mark.parametrize("seed", range(int(os.environ.get("NNCF_SOAK_SEEDS", "24"))))
def test_fused_random_shapes_terminate_and_match(force_row_tiles, seed):
    """Random small shapes -- tiny catalogues (few unique rows: most strips empty), one-item / one-user batches, both
    schemes, 1 / 2 / 4 row tiles -- through two steps each: the in-launch hand-shakes count arrivals from the plans'
    segment counts, and a miscount would hang the launch; every case must finish and match the oracle."""
    rs = np.random.RandomState(1000 + seed)
    b = int(rs.choice([32, 64, 96, 128, 160, 256, 512]))
    d = int(rs.choice([4, 20, 64, 100, 128]))
    nu = int(rs.choice([1, 2, 7, 40, 400, 3000]))
    ni = int(rs.choice([1, 2, 5, 33, 300, 2000]))
    group = bool(rs.randint(2))
    loss = str(rs.choice(["skip-gram", "mse"]))
    force_row_tiles(int(rs.choice([1, 2, 4])))
    conf, eng, orc = _mk(loss, b, d, nu, ni, "mf", "fused", seed=seed)
    scale = 1.0
    if seed >= 24:   # the wider sweep (NNCF_SOAK_SEEDS): row scales too -- the large-batch kernels split their operands into fp16 planes
        scale = float(rs.choice([1e-3, 0.1, 1.0, 8.0]))
        if loss == "mse":
            scale = min(scale, 1.0)   # mse above that diverges on the tiny catalogues (rows of 1e6 after two steps: nothing to compare;
                                      # on the fp16 pipe a NaN loss: test_wide_kernel_fp16_pipe_mse_gradient_overflow_is_loud)
        eng.U.mul_(scale)
        eng.V.mul_(scale)
        orc.U *= scale
        orc.V *= scale
    for step in range(2):
        uid = rs.randint(0, nu, b)
        if group:
            cid = np.repeat(rs.randint(0, ni, b // 4), 4)
            if len(np.unique(cid)) < 2:      # the reference's group loss divides by (n_u - 1): needs two unique items
                continue
            want = orc.step_group_neg_shared(uid, cid)
            got = eng.step_group_neg_shared(eng.ids(uid), eng.ids(cid)).item()
        else:
            cid = rs.randint(0, ni, b)
            want = orc.step_neg_shared(uid, cid)
            got = eng.step_neg_shared(eng.ids(uid), eng.ids(cid)).item()
        assert abs(got - want) <= 1e-5 * max(1.0, abs(want)), (step, got, want)
        np.testing.assert_allclose(eng.U.cpu().numpy(), orc.U, rtol=2e-5, atol=2e-5 * max(1.0, scale))
        np.testing.assert_allclose(eng.V.cpu().numpy(), orc.V, rtol=2e-5, atol=2e-5 * max(1.0, scale))


def test_in_launch_wait_is_bounded_and_fails_loudly(monkeypatch):
    """The read-before-write hand-shake of the in-place step spins on other workgroups of the same launch.  The spin is
    bounded: with the bound set to ~nothing (NNCF_FUSED_SPIN_MS) a waiting workgroup gives up, stores nothing, raises the
    model's error flag -- nncf_model_check reports it and every later loss reads NaN -- instead of hanging the GPU
    (a co-tenant process / stream can keep part of the grid from becoming resident)."""
    monkeypatch.setenv("NNCF_FUSED_SPIN_MS", "0.00001")     # one tick of the 100 MHz clock
    b, d, nu, ni = 2048, 100, 5000, 600
    conf, eng, orc = _mk("skip-gram", b, d, nu, ni, "mf", "fused", seed=2)
    rs = np.random.RandomState(3)
    losses = []
    for _ in range(4):
        # group scheme, b = 2,048 (no split), 16 unique items: the ~128 user strips have ONE stage of columns each, the single
        # item strip 16 stages of rows -- the user strips reach their stores tens of microseconds before the item strip
        # announces its last read: they must wait
        uid, cid = rs.randint(0, nu, b), np.repeat(rs.choice(ni, 16, replace=False), b // 16)
        losses.append(eng.step_group_neg_shared(eng.ids(uid), eng.ids(cid)).item())
    with pytest.raises(RuntimeError, match="timed out"):
        eng.check()
    assert np.isnan(losses[-1])
    # the same workload with the default bound never trips it; and a shared device takes the stage + commit form,
    # which has no in-launch wait at all: bitwise the same tables
    monkeypatch.delenv("NNCF_FUSED_SPIN_MS")
    b, nu = 512, 700
    _, eng_a, _ = _mk("skip-gram", b, d, nu, ni, "mf", "fused", seed=2)
    _, eng_b, _ = _mk("skip-gram", b, d, nu, ni, "mf", "fused", seed=2)
    eng_b.set_shared_device(True)
    rs = np.random.RandomState(3)
    for _ in range(4):
        uid, cid = _batch(rs, b, nu, ni, False)
        la = eng_a.step_neg_shared(eng_a.ids(uid), eng_a.ids(cid)).item()
        lb = eng_b.step_neg_shared(eng_b.ids(uid), eng_b.ids(cid)).item()
        assert la == lb and np.isfinite(la)
    eng_a.check()
    eng_b.check()
    assert torch.equal(eng_a.U, eng_b.U) and torch.equal(eng_a.V, eng_b.V)
