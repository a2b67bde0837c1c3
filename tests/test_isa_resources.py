"""CPU: register / scratch audit of the shipped code objects (tools/isa_resources.py).

VERDICT r4: several hot kernels spilled VGPRs to scratch without the design saying so; in a chain-bound kernel every
scratch reload is one more dependent trip.  This test reads the gfx950 code objects out of nncf_amd/libnncf_hip.so (no
recompilation, no GPU) and fails when a kernel of a DEFAULT path spills a vector register.  Kernels that are reachable
only through a developer switch are listed in OPT_IN with the switch that selects them."""
import os
import re
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import isa_resources  # noqa: E402

# kernel-name patterns that no default path launches, and why
OPT_IN = [
    (r"rocprim::", "rocPRIM's radix sort (device shuffle): library code, its private segment is the sort's own"),
]


@pytest.fixture(scope="module")
def rows():
    return isa_resources.audit()


def test_every_kernel_of_the_library_is_listed(rows):
    names = " ".join(r["name"] for r in rows)
    for k in ("sample_kernel", "fused_strip_kernel", "fused_wide_kernel", "pair_wide_kernel", "shared_tile_kernel", "cnn_conv_fwd_kernel",
              "cnn_conv_bwd_kernel", "tower_pool_kernel", "tower_word_kernel", "tower_bwd_kernel", "eval_topk_kernel", "plan"):
        assert k in names, k
    assert len(rows) > 150


def test_no_default_path_kernel_spills_vector_registers(rows):
    bad = []
    for r in rows:
        if any(re.search(p, r["name"]) for p, _ in OPT_IN):
            continue
        if r.get("vgpr_spill_count", 0) or r.get("scratch_loads", 0) or r.get("scratch_stores", 0):
            bad.append("%s: %d VGPRs spilled, %d scratch loads / %d stores, %d B private" % (
                r["name"], r.get("vgpr_spill_count", 0), r.get("scratch_loads", 0), r.get("scratch_stores", 0),
                r.get("private_segment_fixed_size", 0)))
    assert not bad, "kernels that spill:\n" + "\n".join(bad)


# Scalar-register spills: an SGPR spilled by the compiler lives in a lane of a VGPR (v_writelane_b32 / v_readlane_b32) --
# no scratch traffic, but in a chain-bound kernel every restore is a VALU instruction on the dependent path.  They are LISTED per
# kernel family (profiles/r06_isa_resources.txt, column sspill) and BUDGETED here: the worst instantiation of every family that
# spills today, plus a margin of 8; a family not listed must not spill more than 8.  A change that pushes a kernel past its
# budget fails the CPU suite and has to say why (raise the budget with the measurement beside it).
SGPR_SPILL_BUDGET = {
    "cnn_conv_bwd_kernel": 135,     # 127: six tile pointers + tap offsets per filter group (round 5 formed them per tile to save VGPRs)
    "fused_strip_kernel": 129,      # 121 (group_neg_shared); the headline instantiation <skip-gram, neg_shared> 92
    "plan_steps_kernel": 98,        # 90: once per 64 steps
    "fused_wide_kernel": 58,        # 50
    "plan_kernel": 56,              # 48
    "eval_fused_kernel": 56,        # 48 (the 8-wave form; the default 16-wave form 15): round 6's list cut (keys and compaction masks) on top of 106 SGPRs; the restores sit in the
                                    #     interaction_bias branch of the stage loop and in the cuts, not on the default path of a stage
    "shared_apply_kernel": 46,      # 38
    "cnn_pool_dense_kernel": 44,    # 36
    "tower_word_kernel": 39,        # 31
    "cnn_conv_fwd_kernel": 37,      # 29
    "tower_bwd_kernel": 24,         # 16
    "tower_pool_kernel": 21,        # 13
}


def _family(name):
    m = re.match(r"(?:nncf::)?(?:\(anonymous namespace\)::)?([A-Za-z_0-9]+)", name)
    return m.group(1) if m else name


def test_scalar_register_spills_stay_within_their_budgets(rows):
    worst = {}
    for r in rows:
        if any(re.search(p, r["name"]) for p, _ in OPT_IN):
            continue
        f = _family(r["name"])
        worst[f] = max(worst.get(f, 0), r.get("sgpr_spill_count", 0))
    over = ["%s: %d spilled SGPRs (budget %d)" % (f, n, SGPR_SPILL_BUDGET.get(f, 8)) for f, n in sorted(worst.items())
            if n > SGPR_SPILL_BUDGET.get(f, 8)]
    assert not over, "scalar-register spills past their budget:\n" + "\n".join(over)
    # the list itself is not stale: every budgeted family still exists and still spills (else the entry should go)
    for f in SGPR_SPILL_BUDGET:
        assert f in worst, "budgeted kernel family %s is gone" % f
