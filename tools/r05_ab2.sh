#!/bin/bash
B="--cpu-seconds 0 --recall-epochs 0"
python -m pytest tests/test_eval_gpu.py -x -q 2>&1 | tail -6
python -m pytest tests/test_trainer_gpu.py tests/test_tower_gpu.py tests/test_fused_gpu.py -x -q -n 4 2>&1 | tail -4
python tools/eval_bench.py 2>/dev/null | tail -1 > gpurun_out/r05_eval_bench_v1.json; cut -c1-700 gpurun_out/r05_eval_bench_v1.json
python3 bench.py $B --scheme original --steps 780 2>/dev/null | tail -1 | cut -c1-240
bash tools/quick_prof.sh orig --scheme original --steps 390 --warmup 39 2>&1 | grep -E "tower_word|tower_pool" | cut -c1-150
python3 bench.py $B --large --batch 8192 --scheme group_neg_shared --loss skip-gram --steps 60 --warmup 10 2>/dev/null | tail -1 | cut -c1-240
