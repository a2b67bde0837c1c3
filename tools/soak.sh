#!/bin/bash
# Wider random-shape sweep of the step kernels against the float64 oracle (GPU box): the seeded tests of tests/test_fused_gpu.py,
# tests/test_pair_gpu.py, tests/test_tower_gpu.py, tests/test_step_gpu.py tests/test_sharded_gpu.py, tests/test_schemes_gpu.py and tests/test_trainer_gpu.py (top-k evaluator) with NNCF_SOAK_SEEDS seeds instead of their 24 / 16 / 10 + 12 / 16 + 16 / 12 / 12 / 12.
# (the large-batch sweep of tests/test_pair_gpu.py, ~0.4 s per case, takes seeds / 5, the sampler's bit-exact sweep of tests/test_sampler_gpu.py seeds / 4)
# usage: bash tools/soak.sh [seeds]      -> gpurun_out/soak.log
N=${1:-300}
mkdir -p gpurun_out
NNCF_SOAK_SEEDS=$N NNCF_SOAK_LARGE=$((N / 5)) NNCF_SOAK_SAMPLER=$((N / 4)) python -m pytest tests/test_fused_gpu.py tests/test_pair_gpu.py tests/test_tower_gpu.py tests/test_step_gpu.py tests/test_sharded_gpu.py tests/test_sampler_gpu.py tests/test_schemes_gpu.py tests/test_trainer_gpu.py tests/test_eval_gpu.py -m gpu -q -k "random" -p no:cacheprovider 2>&1 | tail -40 > gpurun_out/soak.log
tail -40 gpurun_out/soak.log
