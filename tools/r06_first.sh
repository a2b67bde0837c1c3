#!/bin/bash
# round-6 first GPU call: one-epoch parity at the BASELINE shape (three cases) + baseline bench + fused_strip stamps
set -u
OUT=gpurun_out
python3 tools/convergence_parity.py --epochs 1 --optimizer sgd --out $OUT/parity_e1_sgd.json > $OUT/parity_e1_sgd.log 2>&1
python3 tools/convergence_parity.py --epochs 1 --optimizer sgd --scheme group_neg_shared --out $OUT/parity_e1_group.json > $OUT/parity_e1_group.log 2>&1
python3 tools/convergence_parity.py --epochs 1 --optimizer adam --out $OUT/parity_e1_adam.json > $OUT/parity_e1_adam.log 2>&1
python3 bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > $OUT/bench_driver.json
NNCF_DEBUG_TIMING=1 python3 bench.py --steps 390 --warmup 39 --cpu-seconds 0 --recall-epochs 0 --dump-stamps $OUT/strip_be.npy > /dev/null 2>&1
python3 tools/fused_timing.py --load $OUT/strip_be.npy --split 8 32 > $OUT/strip_be_stamps.txt 2>&1
NNCF_DEBUG_TIMING=1 python3 bench.py --model mf --steps 390 --warmup 39 --cpu-seconds 0 --recall-epochs 0 --dump-stamps $OUT/strip_mf.npy > /dev/null 2>&1
python3 tools/fused_timing.py --load $OUT/strip_mf.npy --split 8 32 > $OUT/strip_mf_stamps.txt 2>&1
tail -5 $OUT/parity_e1_sgd.log $OUT/parity_e1_group.log $OUT/parity_e1_adam.log
cat $OUT/bench_driver.json | cut -c1-600
