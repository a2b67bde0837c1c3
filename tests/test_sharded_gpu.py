"""GPU: HipBackend of the sharded path (export-gradient step, row gather, deterministic
scatter-apply) at world_size 1 against the plain single-GPU step and the oracle."""
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
