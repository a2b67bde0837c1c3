export TMPDIR=/tmp
NNCF_TOWER_BWD=fused python -m pytest tests/test_tower_gpu.py tests/test_sharded_gpu.py -q 2>&1 | tail -8 > gpurun_out/r02_gputest_d.log
NNCF_TOWER_BWD=fused python bench.py --steps 780 --warmup 390 --cpu-seconds 0 --recall-epochs 0 2>/dev/null | tail -1 > gpurun_out/r02_q_basic_embedding_fusedbwd.json
R=$PWD
(cd /tmp && NNCF_TOWER_BWD=fused rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_x -o p -- python3 $R/bench.py --steps 780 --warmup 390 --cpu-seconds 0 --recall-epochs 0 --profile-launches 10 > $R/gpurun_out/prof_x.log 2>&1)
python3 tools/prof_summary.py gpurun_out/prof_x/p_results.db gpurun_out/r02_q_basic_embedding_fusedbwd_trace.txt
rm -rf gpurun_out/prof_x
