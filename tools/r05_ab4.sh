#!/bin/bash
B="--cpu-seconds 0 --recall-epochs 0"
python -m pytest tests/test_pair_gpu.py tests/test_sharded_gpu.py tests/test_sharded_two_ranks_gpu.py tests/test_bench_gpu.py -x -q -n 4 2>&1 | tail -5
for nat in 1 0; do echo "DP world1 native=$nat"; NNCF_DP_NATIVE=$nat NNCF_FORCE_SHARDED=1 python3 bench.py --steps 1170 $B 2>/dev/null | tail -1 | cut -c1-200; done
for nat in 1 0; do echo "DP world1 RCCL real native=$nat"; NNCF_DP_NATIVE=$nat NNCF_FORCE_SHARDED=1 NNCF_SHARD_NOALIAS=1 python3 bench.py --steps 1170 $B 2>/dev/null | tail -1 | cut -c1-200; done
