"""GPU: bench.py end to end at the quick functional size (--small): the ONE JSON line the driver parses carries the
contract's fields, the roofline object and the cpu_baseline object, for the headline model and for the id-table
model; the step count is honoured exactly."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*extra):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--small", "--gpus", "1", "--steps", "8", "--warmup", "3",
           "--cpu-seconds", "1", "--recall-epochs", "1"] + list(extra)
    out = subprocess.run(cmd, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert out.returncode == 0, out.stderr.decode()[-2000:]
    lines = [ln for ln in out.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line on stdout"
    return json.loads(lines[0])


@pytest.mark.parametrize("extra", [(), ("--model", "mf"), ("--scheme", "group_neg_shared")])
def test_bench_line_contract(extra):
    j = _run(*extra)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in j, k
    assert j["metric"].startswith("positive-links/sec") and j["unit"] == "positive-links/s"
    assert (j["n_gpus"], j["steps"], j["warmup"]) == (1, 8, 3) and j["higher_is_better"] is True
    assert j["scaling"] == "weak" and j["vs_baseline"] is None and j["dtype"] == "f32" and "synthetic" in j["data"]
    assert "workload" in j["config"] and "model" not in j["config"]
    assert j["value"] > 0 and abs(j["value"] - 512 * 1e3 / j["ms_per_step"]) <= 1e-6 * j["value"]
    r = j["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) <= 1e-9 and 0 < r["frac"] < 1
    c = j["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("port", "reference") and c["value"] > 0 and c["cores"] >= 1
    w = j["windows"]
    assert w["count"] >= 5 and w["timed_seconds_total"] >= 0.2 and w["steps_each"] == 8     # several windows, the median reported
    assert abs(j["ms_per_step"] - w["ms_per_step_median"]) <= 1e-9
    assert r["kernel_ms_source"].startswith("HIP event bracket of this run")                 # frac is what THIS run measured
    if "bracket_cost_ms" in r:     # calibrated: every kernel of the step was bracketed and the brackets add up to the step
        assert 0 <= r["bracket_cost_ms"] <= 1.05 * r["empty_bracket_ms"] + 1e-9
        assert abs(r["kernel_ms"] - (r["event_bracket_ms"] - r["bracket_cost_ms"])) <= 1e-9
    else:                          # (the 8-step --small run is host-bound: the brackets need not add up)
        assert "bracket_calibration" in r and abs(r["kernel_ms"] - (r["event_bracket_ms"] - r["empty_bracket_ms"])) <= 1e-9


def _run_full(*extra):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "390", "--warmup", "39", "--cpu-seconds", "0",
           "--recall-epochs", "0"] + list(extra)
    out = subprocess.run(cmd, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert out.returncode == 0, out.stderr.decode()[-2000:]
    lines = [ln for ln in out.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    return json.loads(lines[0])


def test_headline_kernel_times_sum_to_the_step():
    """at the BASELINE size the live per-kernel figures of the headline add up to the driver-timed step
    (sum kernel_ms <= ms_per_step <= sum kernel_ms + 5 %): no bracket correction that flatters the kernels."""
    j = _run_full()
    ks = [k for k in [j["roofline"]] + j["roofline_other_kernels"] if k.get("launches")]
    assert {"fused_strip_kernel", "tower_pool_kernel", "tower_bwd_kernel", "tower_word_kernel"} <= {k["kernel"] for k in ks}
    tot = sum(k["launches_per_step"] * k["kernel_ms"] for k in ks)
    assert tot <= j["ms_per_step"] * 1.0005 and j["ms_per_step"] <= tot * 1.05, (tot, j["ms_per_step"])
    assert all("bracket_cost_ms" in k for k in ks)


def test_config2_tile_work_is_the_measured_unique_items():
    """group_neg_shared's tile kernel is priced at 6 b n_u d with the run's measured n_u, not at N = b."""
    j = _run_full("--scheme", "group_neg_shared")
    tile = [k for k in [j["roofline"]] + j["roofline_other_kernels"] if k["kernel"] == "shared_tile_kernel"][0]
    n_u = j["mean_unique_items_per_step"]
    assert 60 < n_u < 300
    assert abs(tile["algorithmic_per_launch"] - 6.0 * 512 * n_u * 100) <= 1e-6 * tile["algorithmic_per_launch"]


@pytest.mark.parametrize("extra", [("--model", "mf"), ("--model", "mf", "--scaling", "strong"), ()])
def test_multi_rank_lines_carry_roofline_and_cpu_baseline(extra, monkeypatch):
    """The N > 1 code paths (item-sharded weak / strong, data-parallel tower) run here as ONE rank over a real process group
    (NNCF_FORCE_SHARDED=1): their line has the same objects as the N = 1 line -- roofline of the same kernel, rank 0's
    cpu_baseline, the timed windows."""
    monkeypatch.setenv("NNCF_FORCE_SHARDED", "1")
    monkeypatch.setenv("MASTER_PORT", "29533")
    j = _run(*extra)
    assert j["n_gpus"] == 1 and j["scaling"] == ("strong" if "strong" in extra else "weak")
    r, c, w = j["roofline"], j["cpu_baseline"], j["windows"]
    assert r["kernel"] == "fused_strip_kernel" and r["bound"] == "mfma" and 0 < r["frac"] < 1
    assert abs(r["frac"] - r["achieved"] / r["peak"]) <= 1e-9
    assert c["kind"] == "port" and c["value"] > 0 and c["cores"] >= 1
    assert w["count"] >= 5 and w["timed_seconds_total"] >= 0.2
    assert j["value"] > 0 and abs(j["value"] - 512 * 1e3 / j["ms_per_step"]) <= 1e-6 * j["value"]
    if "mf" in extra:
        assert j["config"]["wire_bytes_per_rank_per_step"] == 0        # one rank: nothing travels
