#!/bin/bash
# round-5 checkpoint 1: parity of the touched kernels + the benches they move
B="--cpu-seconds 0 --recall-epochs 0"
python -m pytest tests/test_fused_gpu.py tests/test_pair_gpu.py tests/test_sampler_gpu.py -x -q 2>&1 | tail -4
python -m pytest tests/test_tower_gpu.py tests/test_step_gpu.py tests/test_schemes_gpu.py -x -q -n 4 2>&1 | tail -4
python tools/sampler_bench.py 2>&1 | tail -1 > gpurun_out/r05_sampler_bench_v3.json
NNCF_SAMPLER_LOCKSTEP=0 NNCF_SAMPLER_LDS=0 python tools/sampler_bench.py 2>&1 | tail -1 > gpurun_out/r05_sampler_bench_v3_global_seq.json
NNCF_SAMPLER_LDS=0 python tools/sampler_bench.py 2>&1 | tail -1 > gpurun_out/r05_sampler_bench_v3_global_lock.json
for f in gpurun_out/r05_sampler_bench_v3*.json; do echo $f; python3 -c "
import json,sys
d=json.loads(open('$f').read())
for k,v in d.items():
    if isinstance(v,dict) and 'kernel_us' in v: print(' ', k, round(v['kernel_us'],2),'us', '%.3g'%v['samples_per_s_kernel'])
"; done
python3 bench.py $B --steps 1170 2>/dev/null | tail -1 | cut -c1-300
python3 bench.py $B --scheme original --steps 780 2>/dev/null | tail -1 | cut -c1-300
python3 bench.py $B --model cnn_embedding --scheme group_neg_shared --steps 780 2>/dev/null | tail -1 | cut -c1-300
python3 bench.py $B --large --batch 8192 --steps 60 --warmup 10 2>/dev/null | tail -1 | cut -c1-300
python3 bench.py $B --large --batch 8192 --loss mse --steps 60 --warmup 10 2>/dev/null | tail -1 | cut -c1-300
python3 bench.py $B --large --batch 8192 --scheme group_neg_shared --loss skip-gram --steps 60 --warmup 10 2>/dev/null | tail -1 | cut -c1-300
python3 bench.py $B --large --batch 8192 --scheme group_neg_shared --loss mse --steps 60 --warmup 10 2>/dev/null | tail -1 | cut -c1-300
bash tools/quick_prof.sh orig --scheme original --steps 390 --warmup 39 2>&1 | grep -E "tower_word|tower_pool|original" | cut -c1-150
bash tools/quick_prof.sh head --steps 780 --warmup 390 2>&1 | grep -E "fused_strip|tower_" | cut -c1-150
