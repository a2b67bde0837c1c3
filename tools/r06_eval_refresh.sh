#!/bin/bash
# the evaluator's records after a kernel change: tests, bench JSON (d = 100, 128), kernel trace, A/B against round 5's kernel and the
# 8-wave form, phase stamps
set -u
R=$PWD; OUT=$R/gpurun_out; export TMPDIR=/tmp
timeout 900 python3 -m pytest tests/test_eval_gpu.py tests/test_trainer_gpu.py tests/test_epoch_parity_gpu.py -m gpu -q 2>&1 | tail -3 > $OUT/r06_eval_tests.txt
cat $OUT/r06_eval_tests.txt
python3 tools/eval_bench.py 2>/dev/null | tail -1 > $OUT/r06_eval_bench.json
python3 tools/eval_bench.py 128 2>/dev/null | tail -1 > $OUT/r06_eval_bench_d128.json
(cd /tmp && rocprofv3 --kernel-trace --stats -d $OUT/prof_ev -o p -- python3 $R/tools/eval_bench.py > /dev/null 2>&1); python3 tools/prof_summary.py $OUT/prof_ev/p_results.db $OUT/r06_eval_kernel_trace.txt; rm -rf $OUT/prof_ev
bash tools/ab_eval.sh "evalold" > $OUT/r06_eval_ab.txt 2>&1
for rep in 1 2; do NNCF_EVAL_WAVES=8 python3 tools/eval_bench.py 2>/dev/null | tail -1 > /tmp/e.json; python3 -c "
import json; j=json.load(open('/tmp/e.json')); print('main NNCF_EVAL_WAVES=8', 'fused pass %.4f ms' % j['fused']['kernel_ms_per_pass'], 'frac %.3f' % j['roofline']['frac'])" >> $OUT/r06_eval_ab.txt; done
cp nncf_amd/libnncf_hip.so /tmp/main.so; cp ab/libnncf_hip_evalstamps.so nncf_amd/libnncf_hip.so
python3 tools/eval_stamps.py 2>/dev/null | tail -20 > $OUT/r06_eval_stamps_after16.txt
cp /tmp/main.so nncf_amd/libnncf_hip.so
python3 tools/isa_resources.py --out $OUT/r06_isa_resources.txt > /dev/null
cat $OUT/r06_eval_ab.txt; head -c 600 $OUT/r06_eval_bench.json; echo; head -c 300 $OUT/r06_eval_bench_d128.json; echo; cat $OUT/r06_eval_stamps_after16.txt
