export TMPDIR=/tmp
python -m pytest tests/test_fused_gpu.py tests/test_tower_gpu.py -q 2>&1 | tail -6 > gpurun_out/r02_fused_b.log
for m in basic_embedding; do for p in fused; do NNCF_SHARED_PATH=$p python bench.py --model $m --steps 780 --warmup 390 --cpu-seconds 0 --recall-epochs 0 2>/dev/null | tail -1 > gpurun_out/r02_q_${m}_${p}.json; done; done
R=$PWD
for m in basic_embedding; do
 (cd /tmp && rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$m -o p -- python3 $R/bench.py --model $m --steps 780 --warmup 390 --cpu-seconds 0 --recall-epochs 0 --profile-launches 10 > $R/gpurun_out/prof_$m.log 2>&1)
 python3 tools/prof_summary.py gpurun_out/prof_$m/p_results.db gpurun_out/r02_q_${m}_trace.txt
 rm -rf gpurun_out/prof_$m
done
