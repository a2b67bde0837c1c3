"""CPU: the torch-CPU baseline step (oracle/step_torch.py, bench.py's cpu_baseline leg) against the
numpy oracle on identical inputs: id tables and the mean-pool text tower, 3 schemes x 4 losses."""
import numpy as np
import pytest

from oracle import step_oracle as so
from oracle.step_torch import TorchStep

LOSSES = ["skip-gram", "mse", "log-loss", "max-margin"]


def _setup(loss, scheme, model, seed=0):
    rs = np.random.RandomState(seed)
    b, k, d, nu, ni, L, W = 16, 3, 8, 30, 25, 6, 40
    conf = so.OracleConf(batch_size_p=b, num_negatives=k if scheme == "original" else 0, loss=loss,
                         loss_gamma={"max-margin": 0.1, "log-loss": 10.}.get(loss), neg_loss_weight=16., u_reg=1e-3,
                         learn_rate=0.05, model_choice=model)
    U = rs.uniform(-.5, .5, (nu, d)).astype(np.float32)
    V = rs.uniform(-.5, .5, (ni, d)).astype(np.float32)
    tower = None
    if model != "mf":
        tower = dict(C=rs.randint(0, W, (ni, L)), W=rs.uniform(-.5, .5, (W, d)).astype(np.float32),
                     Wd=rs.uniform(-.5, .5, (d, d)).astype(np.float32), bd=rs.uniform(-.1, .1, d).astype(np.float32),
                     gamma=rs.uniform(.5, 1.5, d).astype(np.float32), beta=rs.uniform(.1, .5, d).astype(np.float32))
    return rs, conf, U, V, tower, (b, k, nu, ni, W)


@pytest.mark.parametrize("loss", LOSSES)
@pytest.mark.parametrize("scheme", ["original", "neg_shared", "group_neg_shared"])
@pytest.mark.parametrize("model", ["mf", "basic_embedding"])
def test_torch_step_matches_numpy_oracle(loss, scheme, model):
    rs, conf, U, V, tower, (b, k, nu, ni, W) = _setup(loss, scheme, model)
    to = None
    if tower is not None:
        to = so.TowerOracle(tower["C"], tower["W"], tower["Wd"], tower["bd"], tower["gamma"], tower["beta"], conf.learn_rate)
    o = so.StepOracle(U, None if tower is not None else V, conf, dtype=np.float32, tower=to)
    t = TorchStep(U, None if tower is not None else V, conf, tower=tower, threads=1)
    for step in range(3):
        uid = rs.randint(0, nu, b)
        cid = np.repeat(rs.randint(0, ni, b // 4), 4) if scheme == "group_neg_shared" else rs.randint(0, ni, b)
        scale = ((rs.rand(W) < 0.7) / 0.7).astype(np.float32) if tower is not None else None
        o.row_scale = t.row_scale = scale
        if scheme == "original":
            uu, cc, yy = so.assemble_original(np.stack([uid, cid, np.ones(b, dtype=int)], 1), rs.randint(0, ni, b * k), loss)
            want, got = o.step_original(uu, cc, yy), t.step_original(uu, cc, yy)
        elif scheme == "neg_shared":
            want, got = o.step_neg_shared(uid, cid), t.step_neg_shared(uid, cid)
        else:
            want, got = o.step_group_neg_shared(uid, cid), t.step_group_neg_shared(uid, cid)
        assert abs(got - want) <= 2e-5 * max(1.0, abs(want)), (step, got, want)
        np.testing.assert_allclose(t.U.numpy(), o.U, rtol=2e-4, atol=2e-6)
        if tower is None:
            np.testing.assert_allclose(t.V.numpy(), o.V, rtol=2e-4, atol=2e-6)
        else:
            for name in ("W", "Wd", "bd", "gamma", "beta"):
                np.testing.assert_allclose(t.tw[name].numpy(), getattr(to, name), rtol=5e-4, atol=5e-6, err_msg=name)
