#!/bin/bash
# The large-batch bench lines of tools/profile_round.sh alone (same commands, same file names): bash tools/profile_large.sh r05
TAG=${1:-r05}
OUT=$PWD/gpurun_out
B="--recall-epochs 0"
python3 bench.py --large --batch 8192 --steps 100 --warmup 10 $B 2>/dev/null | tail -1 > $OUT/${TAG}_bench_large_mf_neg_shared_b8192.json
python3 bench.py --large --batch 4096 --steps 100 --warmup 10 $B 2>/dev/null | tail -1 > $OUT/${TAG}_bench_large_mf_neg_shared_b4096.json
python3 bench.py --large --batch 16384 --steps 50 --warmup 5 $B 2>/dev/null | tail -1 > $OUT/${TAG}_bench_large_mf_neg_shared_b16384.json
for l in log-loss max-margin; do for sc in neg_shared group_neg_shared; do
python3 bench.py --large --batch 8192 --loss $l --scheme $sc --steps 100 --warmup 10 --cpu-seconds 0 $B 2>/dev/null | tail -1 > $OUT/${TAG}_bench_large_mf_${sc}_${l}_b8192.json
done; done
python3 bench.py --large --batch 4096 --loss log-loss --steps 100 --warmup 10 --cpu-seconds 0 $B 2>/dev/null | tail -1 > $OUT/${TAG}_bench_large_mf_neg_shared_log-loss_b4096.json
python3 bench.py --large --batch 16384 --loss log-loss --steps 50 --warmup 5 --cpu-seconds 0 $B 2>/dev/null | tail -1 > $OUT/${TAG}_bench_large_mf_neg_shared_log-loss_b16384.json
python3 bench.py --large --batch 32768 --steps 20 --warmup 3 --cpu-seconds 0 $B 2>/dev/null | tail -1 > $OUT/${TAG}_bench_large_mf_neg_shared_b32768.json
python3 bench.py --large --batch 32768 --loss log-loss --steps 20 --warmup 3 --cpu-seconds 0 $B 2>/dev/null | tail -1 > $OUT/${TAG}_bench_large_mf_neg_shared_log-loss_b32768.json
for l in skip-gram mse; do
python3 bench.py --large --batch 8192 --scheme group_neg_shared --loss $l --steps 100 --warmup 10 --cpu-seconds 0 $B 2>/dev/null | tail -1 > $OUT/${TAG}_bench_large_mf_group_neg_shared_${l}_b8192.json
done
python3 bench.py --large --batch 8192 --loss mse --steps 100 --warmup 10 --cpu-seconds 0 $B 2>/dev/null | tail -1 > $OUT/${TAG}_bench_large_mf_neg_shared_mse_b8192.json
NNCF_WIDE_F16SPLIT=force python3 bench.py --large --batch 8192 --steps 100 --warmup 10 --cpu-seconds 0 $B 2>/dev/null | tail -1 > $OUT/${TAG}_bench_large_mf_neg_shared_b8192_unguarded.json
NNCF_FORCE_SHARDED=1 python3 bench.py --large --batch 8192 --steps 100 --warmup 10 --cpu-seconds 0 $B 2>/dev/null | tail -1 > $OUT/${TAG}_bench_large_mf_sharded_world1_b8192.json
ls -la $OUT/${TAG}_bench_large_mf_*
