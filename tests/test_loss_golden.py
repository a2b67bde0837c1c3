"""CPU: the numpy oracle (oracle/step_oracle.py) against tests/golden/loss_golden.npz -- numbers the
REFERENCE's own loss / interaction / regulariser text produced when tools/gen_golden_losses.py
exec'd it here (utils/objectives.py:15-220, interaction_dot.py:82-107, utilities.py:122-135).
Rows a11, a15-a20 of SURVEY.md section 8: 4 families x 4 losses x {toy, content-normalised,
bias, b=32, b=512 d=100 (mf and content)}.
float64 oracle vs the float64 reference run: formulas to rounding error (1e-10);
float32 oracle vs the float32 reference run: the 1e-5 budget of north_star."""
import os

import numpy as np
import pytest

from golden_inputs import case_inputs, parse_meta
from oracle import step_oracle as so

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "loss_golden.npz"))
CASES = [parse_meta(m) for m in GOLD["meta"]]
IDS = ["%d-%s-%s-%s" % (i, c["family"], c["loss"], c["tag"]) for i, c in enumerate(CASES)]


def oracle_step(c, i, dtype):
    """One lr = 1 SGD step of the oracle on the case's inputs: (oracle, loss, loss_fn(pred) -> (l, G))."""
    p = "c%d_" % i
    if c["full"]:
        U, V = GOLD[p + "U"], GOLD[p + "V"]
        _, _, _, _, ub, cb = case_inputs(c["seed"], c["family"], c["b"], c["k"], c["d"], c["nu"], c["ni"], c["scale"])
    else:
        U, V, _, _, ub, cb = case_inputs(c["seed"], c["family"], c["b"], c["k"], c["d"], c["nu"], c["ni"], c["scale"])
    uid, cid = GOLD[p + "uid"].astype(np.int64), GOLD[p + "cid"].astype(np.int64)
    conf = so.OracleConf(batch_size_p=c["b"], num_negatives=c["k"], loss=c["loss"], loss_gamma=c["gamma"],
                         neg_loss_weight=c["lam"], u_reg=c["u_reg"], learn_rate=1.0, emb_normalization=c["normalize"],
                         model_choice="mf" if c["mf"] else "basic_embedding", interaction_bias=c["bias"])
    o = so.StepOracle(U, V, conf, dtype=dtype)
    if c["bias"]:
        o.ub[:] = GOLD[p + "ub"]
        o.cb[:] = GOLD[p + "cb"]
    fam = c["family"]
    if fam == "original":
        y = np.ones(len(uid))
        y[c["b"]:] = -1 if c["loss"] == "skip-gram" else 0
        loss = o.step_original(uid, cid, y)
    elif fam == "neg_shared":
        loss = o.step_neg_shared(uid, cid)
    elif fam == "group_neg_shared":
        loss = o.step_group_neg_shared(uid, cid)
    else:
        loss = o.step_sampled_neg_shared(uid, cid)
    return o, loss, conf


def loss_fn(c, conf, pred, i):
    fam = c["family"]
    if fam == "original":
        y = np.ones(pred.shape[0], dtype=pred.dtype)
        y[c["b"]:] = -1 if c["loss"] == "skip-gram" else 0
        l, g = so.original_loss(pred.reshape(-1), y, conf)
        return l, g.reshape(-1, 1)
    if fam == "neg_shared":
        return so.neg_shared_loss(pred, conf)
    if fam == "group_neg_shared":
        return so.group_neg_shared_loss(pred, GOLD["c%d_cid_x" % i].astype(np.int64), conf)
    return so.sampled_neg_shared_loss(pred, conf)


@pytest.mark.parametrize("i", range(len(CASES)), ids=IDS)
def test_oracle_matches_reference_text_float64(i):
    c, p = CASES[i], "c%d_" % i
    o, loss, conf = oracle_step(c, i, np.float64)
    want_loss = GOLD[p + "loss64"][0]
    assert abs(loss - want_loss) <= 1e-11 * max(1.0, abs(want_loss)), (loss, want_loss)
    dU, dV = o.last["dU"], o.last["dV"]
    if c["family"] == "group_neg_shared":
        assert np.array_equal(o.cid_u, GOLD[p + "cid_u"]) and np.array_equal(o.cid_x, GOLD[p + "cid_x"])
    if c["full"]:
        np.testing.assert_allclose(o.scores.reshape(GOLD[p + "pred64"].shape), GOLD[p + "pred64"], rtol=1e-11, atol=1e-13)
        l, G = loss_fn(c, conf, GOLD[p + "pred64"], i)
        np.testing.assert_allclose(G, GOLD[p + "dpred64"], rtol=1e-9, atol=1e-13)
        assert abs(l - GOLD[p + "loss64"][1]) <= 1e-11 * max(1.0, abs(l))
        np.testing.assert_allclose(dU, GOLD[p + "dU64"], rtol=1e-9, atol=1e-13)
        np.testing.assert_allclose(dV, GOLD[p + "dV64"], rtol=1e-9, atol=1e-13)
    else:
        R = GOLD["proj"][:c["d"]]
        np.testing.assert_allclose(dU[:8], GOLD[p + "dU_head64"], rtol=1e-9, atol=1e-14)
        np.testing.assert_allclose(dV[:8], GOLD[p + "dV_head64"], rtol=1e-9, atol=1e-14)
        np.testing.assert_allclose(dU @ R, GOLD[p + "dU_proj64"], rtol=1e-8, atol=1e-12)
        np.testing.assert_allclose(dV @ R, GOLD[p + "dV_proj64"], rtol=1e-8, atol=1e-12)
    if c["bias"]:   # lr = 1 SGD: bias_after = bias_before - gradient
        np.testing.assert_allclose(GOLD[p + "ub"] - o.ub, GOLD[p + "dub64"], rtol=1e-9, atol=1e-13)
        np.testing.assert_allclose(GOLD[p + "cb"] - o.cb, GOLD[p + "dcb64"], rtol=1e-9, atol=1e-13)


@pytest.mark.parametrize("i", range(len(CASES)), ids=IDS)
def test_oracle_float32_within_parity_budget(i):
    """The fp32 oracle (what the GPU tests compare against) vs the reference text run in fp32."""
    c, p = CASES[i], "c%d_" % i
    o, loss, conf = oracle_step(c, i, np.float32)
    want = GOLD[p + "loss32"][0]
    assert abs(loss - want) <= 1e-5 * max(1.0, abs(want)), (loss, want)
    key = "%s64" if c["full"] else "%s_head64"
    for name, got in (("dU", o.last["dU"]), ("dV", o.last["dV"])):
        ref = GOLD[p + key % name]
        got = got if c["full"] else got[:8]
        scale = max(np.abs(ref).max(), 1e-30)
        assert np.abs(got - ref).max() <= 2e-5 * scale, (name, np.abs(got - ref).max(), scale)
