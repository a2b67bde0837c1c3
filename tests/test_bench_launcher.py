"""CPU: `python bench.py --gpus N` outside torchrun starts its own N ranks.  Runs the
launcher for real with NNCF_BENCH_DRYRUN=1 (gloo, the exchange pattern of the item-sharded step with a
no-arithmetic backend): rank 0's single JSON line comes back with n_gpus = N, and a failing rank makes
the parent exit non-zero."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra_env, *argv):
    env = dict(os.environ, NNCF_BENCH_DRYRUN="1", **extra_env)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    env.pop("LOCAL_RANK", None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(argv), stdout=subprocess.PIPE,
                          stderr=subprocess.PIPE, universal_newlines=True, env=env, cwd="/tmp", timeout=300)


def test_gpus_2_spawns_two_ranks_and_relays_one_line():
    p = _run({}, "--gpus", "2", "--small", "--steps", "4", "--warmup", "2")
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, p.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 4 and line["warmup"] == 2 and line["dry_run"] is True
    assert line["config"]["global_batch"] == 2 * 512 and line["scaling"] == "weak"


def test_failing_rank_fails_the_launcher():
    p = _run({"NNCF_BENCH_DRYRUN_FAIL": "1"}, "--gpus", "2", "--small", "--steps", "2", "--warmup", "1")
    assert p.returncode != 0
    assert not [l for l in p.stdout.splitlines() if l.strip().startswith("{")]


def test_world_size_mismatch_is_refused():
    env = dict(os.environ, NNCF_BENCH_DRYRUN="1", WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, universal_newlines=True, env=env, cwd="/tmp", timeout=120)
    assert p.returncode != 0 and "WORLD_SIZE" in p.stderr
