"""GPU: the HIP step kernels (through the C ABI) against tests/golden/loss_golden.npz -- numbers
produced by the REFERENCE's own loss / interaction / regulariser text (tools/gen_golden_losses.py;
utils/objectives.py:15-220, interaction_dot.py:82-107, utilities.py:122-135) -- with no oracle in
between.  One sparse-SGD step with a power-of-two learning rate; loss within 1e-5 relative
(north_star), d(loss)/d(table row) = (row_before - row_after) / lr within 2e-5 of the largest
gradient entry (fp32 kernels with hardware exp2 / log2 / rcp vs the float64 run of the formulas)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import torch  # noqa: E402

from golden_inputs import case_inputs, parse_meta  # noqa: E402
from nncf_amd.engine import Engine  # noqa: E402

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "loss_golden.npz"))
CASES = [parse_meta(m) for m in GOLD["meta"]]
SHARED = ("neg_shared", "group_neg_shared")
RUNNABLE = [i for i, c in enumerate(CASES) if not (c["family"] in SHARED and c["b"] % 32)]
IDS = ["%d-%s-%s-%s" % (i, CASES[i]["family"], CASES[i]["loss"], CASES[i]["tag"]) for i in RUNNABLE]


def _wide_cases():
    """pointwise shared-negative cases without bias: the ones the one-launch step runs -- again through its large-batch
    kernel (fused_wide_kernel, 2 and 4 row tiles per workgroup), forced at the goldens' sizes"""
    return [i for i in RUNNABLE if CASES[i]["family"] in SHARED and CASES[i]["loss"] in ("skip-gram", "mse")
            and not CASES[i]["bias"] and CASES[i]["b"] % 16 == 0]


@pytest.mark.parametrize("mt", [2, 4])
@pytest.mark.parametrize("i", _wide_cases(), ids=["%d-%s-%s-%s" % (i, CASES[i]["family"], CASES[i]["loss"], CASES[i]["tag"])
                                                  for i in _wide_cases()])
def test_wide_kernel_matches_reference_text(i, mt, monkeypatch):
    monkeypatch.setenv("NNCF_FUSED_MT", str(mt))
    test_hip_step_matches_reference_text(i)


@pytest.mark.parametrize("i", RUNNABLE, ids=IDS)
def test_hip_step_matches_reference_text(i):
    c, p = CASES[i], "c%d_" % i
    U, V, _, _, ub, cb = case_inputs(c["seed"], c["family"], c["b"], c["k"], c["d"], c["nu"], c["ni"], c["scale"])
    uid, cid = GOLD[p + "uid"].astype(np.int64), GOLD[p + "cid"].astype(np.int64)
    b, k, fam = c["b"], c["k"], c["family"]
    head = GOLD[p + ("dU64" if c["full"] else "dU_head64")]
    gmax = max(np.abs(head).max(), 1e-12)
    lr = float(2.0 ** np.round(np.log2(np.abs(U).max() / gmax)))      # lr * gradient ~ table scale; exact scaling
    dev = torch.device("cuda", 0)
    Ut = torch.as_tensor(U, dtype=torch.float32).to(dev).contiguous()
    Vt = torch.as_tensor(V, dtype=torch.float32).to(dev).contiguous()
    U0, V0 = Ut.cpu().numpy().astype(np.float64), Vt.cpu().numpy().astype(np.float64)
    # model_framework.py:62-63 normalises U_emb for every head but sampled_neg_shared (U_emb_front, :64-65);
    # :109-111 normalises the item side inside the content-model branch only
    norm_u = c["normalize"] and fam != "sampled_neg_shared"
    norm_v = c["normalize"] and not c["mf"]
    eng = Engine(c["nu"], c["ni"], c["d"], b, k if fam in ("original", "sampled_neg_shared") else 0, c["loss"], c["gamma"],
                 c["lam"], c["u_reg"], lr, norm_u, norm_v, tables=(Ut, Vt), interaction_bias=c["bias"])
    if c["bias"]:
        eng.bias["ub"].copy_(torch.as_tensor(ub, dtype=torch.float32))
        eng.bias["cb"].copy_(torch.as_tensor(cb, dtype=torch.float32))
        ub0, cb0 = eng.bias["ub"].cpu().numpy().astype(np.float64), eng.bias["cb"].cpu().numpy().astype(np.float64)
    if fam == "original":
        loss = eng.step_original(eng.ids(uid[:b]), eng.ids(cid)).item()
    elif fam == "neg_shared":
        loss = eng.step_neg_shared(eng.ids(uid), eng.ids(cid)).item()
    elif fam == "group_neg_shared":
        out, cu, cx, nu = eng.step_group_neg_shared(eng.ids(uid), eng.ids(cid), want_unique=True)
        loss = out.item()
        n_u = int(nu.item())
        assert np.array_equal(cu.cpu().numpy()[:n_u], GOLD[p + "cid_u"]) and np.array_equal(cx.cpu().numpy(), GOLD[p + "cid_x"])
    else:   # the k shared negatives replicated per positive (what SampledNegSharedModel feeds the per-positive kernels)
        loss = eng.step_original(eng.ids(uid[:b]), eng.ids(np.concatenate([cid[:b], np.tile(cid[b:], b)]))).item()
    want = GOLD[p + "loss64"][0]
    assert abs(loss - want) <= 1e-5 * max(1.0, abs(want)), (loss, want)
    dU = (U0 - eng.U.cpu().numpy().astype(np.float64)) / lr
    dV = (V0 - eng.V.cpu().numpy().astype(np.float64)) / lr
    for name, got in (("dU", dU), ("dV", dV)):
        if c["full"]:
            ref = GOLD[p + name + "64"]
            scale = max(np.abs(ref).max(), 1e-30)
            assert np.abs(got - ref).max() <= 2e-5 * scale, (name, np.abs(got - ref).max(), scale)
        else:
            ref = GOLD[p + name + "_head64"]
            scale = max(np.abs(ref).max(), 1e-30)
            assert np.abs(got[:8] - ref).max() <= 2e-5 * scale, (name, np.abs(got[:8] - ref).max(), scale)
            proj, want_p = got @ GOLD["proj"][:c["d"]], GOLD[p + name + "_proj64"]
            assert np.abs(proj - want_p).max() <= 4e-5 * max(np.abs(want_p).max(), 1e-30), name
    if c["bias"]:
        for name, before, t in (("dub", ub0, eng.bias["ub"]), ("dcb", cb0, eng.bias["cb"])):
            ref = GOLD[p + name + "64"]
            got = (before - t.cpu().numpy().astype(np.float64)) / lr
            # pairwise `original`: a user's bias cancels in every (positive - negative) pair: exactly 0 (reference: 1e-17)
            assert np.abs(got - ref).max() <= 2e-5 * max(np.abs(ref).max(), 1e-3 * gmax), name
