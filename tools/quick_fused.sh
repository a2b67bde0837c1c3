export TMPDIR=/tmp
NNCF_TOWER_BWD=fused python -m pytest tests/test_tower_gpu.py -q -k "every_kernel_path or bitwise" 2>&1 | tail -5 > gpurun_out/r02_gputest_d.log
NNCF_TOWER_BWD=fused python bench.py --steps 780 --warmup 390 --cpu-seconds 0 --recall-epochs 0 2>/dev/null | tail -1 > gpurun_out/r02_q_basic_embedding_fusedbwd.json
