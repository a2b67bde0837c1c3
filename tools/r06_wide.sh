#!/bin/bash
# round-6 large-batch kernel check: parity tests of the wide kernels, bench lines at b = 8,192, phase stamps
set -u
OUT=gpurun_out
B="--recall-epochs 0 --cpu-seconds 0"
python3 -m pytest tests/test_fused_gpu.py tests/test_pair_gpu.py -x -q 2>&1 | tail -4 > $OUT/wide_tests.txt
python3 bench.py --large --batch 8192 --steps 100 --warmup 10 $B 2>/dev/null | tail -1 > $OUT/wide_b8192.json
python3 bench.py --large --batch 4096 --steps 100 --warmup 10 $B 2>/dev/null | tail -1 > $OUT/wide_b4096.json
python3 bench.py --large --batch 8192 --loss mse --steps 100 --warmup 10 $B 2>/dev/null | tail -1 > $OUT/wide_b8192_mse.json
python3 bench.py --large --batch 8192 --scheme group_neg_shared --loss skip-gram --steps 100 --warmup 10 $B 2>/dev/null | tail -1 > $OUT/wide_b8192_group.json
python3 bench.py --large --batch 8192 --loss log-loss --steps 100 --warmup 10 $B 2>/dev/null | tail -1 > $OUT/pair_b8192_logloss.json
python3 tools/fused_timing.py skip-gram 8192 128 2>/dev/null | tail -19 > $OUT/wide_phase_stamps.txt
cat $OUT/wide_tests.txt
for f in wide_b8192 wide_b4096 wide_b8192_mse wide_b8192_group pair_b8192_logloss; do python3 -c "
import json,sys; j=json.load(open('$OUT/$f.json')); r=j['roofline']; print('$f', round(j['ms_per_step']*1e3,1), 'us/step', r['kernel'], round(r['kernel_ms']*1e3,1), 'us', r.get('f16_domain','')[:40])"; done
cat $OUT/wide_phase_stamps.txt
