#!/bin/bash
# round-6 second refresh (after the `original` group merge and the evaluator work): everything tools/r06_final.sh collects, plus the
# traces / bench lines of the kernels that changed, the evaluator's bench + trace + phase stamps, the register audit
set -u
TAG=r06
R=$PWD
OUT=$R/gpurun_out
export TMPDIR=/tmp
B="--recall-epochs 0"
bash tools/r06_final.sh > $OUT/r06_final_log.txt 2>&1
tail -8 $OUT/r06_final_log.txt
run_prof () {
  local name=$1; shift
  local extra=$1; shift
  (cd /tmp && rocprofv3 --kernel-trace $extra -d $OUT/prof_$name -o p -- python3 $R/bench.py "$@" $B > $OUT/prof_$name.log 2>&1)
  python3 $R/tools/prof_summary.py $OUT/prof_$name/p_results.db $OUT/${TAG}_$name.txt
  rm -rf $OUT/prof_$name
}
run_prof basic_embedding_original_kernel_trace "--stats" --scheme original --steps 390 --warmup 39 --cpu-seconds 0 --profile-launches 10
run_prof basic_embedding_group_neg_shared_kernel_trace "--stats" --scheme group_neg_shared --steps 780 --warmup 390 --cpu-seconds 0 --profile-launches 10
run_prof cnn_embedding_original_kernel_trace "--stats" --model cnn_embedding --scheme original --steps 100 --warmup 20 --cpu-seconds 0 --profile-launches 10
run_prof cnn_embedding_group_neg_shared_kernel_trace "--stats" --model cnn_embedding --scheme group_neg_shared --steps 390 --warmup 39 --cpu-seconds 0 --profile-launches 10
python3 bench.py --scheme original --cpu-seconds 6 --steps 780 2>/dev/null | tail -1 > $OUT/${TAG}_bench_basic_embedding_original.json
python3 bench.py --scheme group_neg_shared --cpu-seconds 6 --steps 1170 2>/dev/null | tail -1 > $OUT/${TAG}_bench_basic_embedding_group_neg_shared.json
python3 bench.py --model cnn_embedding --scheme original --cpu-seconds 0 --steps 200 --warmup 20 $B 2>/dev/null | tail -1 > $OUT/${TAG}_bench_cnn_embedding_original.json
python3 bench.py --model cnn_embedding --scheme group_neg_shared --cpu-seconds 8 --steps 780 2>/dev/null | tail -1 > $OUT/${TAG}_bench_cnn_embedding_group_neg_shared.json
python3 tools/eval_bench.py 2>/dev/null | tail -1 > $OUT/${TAG}_eval_bench.json
python3 tools/eval_bench.py 128 2>/dev/null | tail -1 > $OUT/${TAG}_eval_bench_d128.json
(cd /tmp && rocprofv3 --kernel-trace --stats -d $OUT/prof_ev -o p -- python3 $R/tools/eval_bench.py > /dev/null 2>&1); python3 tools/prof_summary.py $OUT/prof_ev/p_results.db $OUT/${TAG}_eval_kernel_trace.txt; rm -rf $OUT/prof_ev
python3 tools/isa_resources.py --out $OUT/${TAG}_isa_resources.txt > /dev/null
cp nncf_amd/libnncf_hip.so /tmp/main.so; cp ab/libnncf_hip_evalstamps.so nncf_amd/libnncf_hip.so
python3 tools/eval_stamps.py 2>/dev/null | tail -20 > $OUT/${TAG}_eval_stamps_after.txt
cp /tmp/main.so nncf_amd/libnncf_hip.so
bash tools/ab_eval.sh "evalold" > $OUT/${TAG}_eval_ab.txt 2>&1
for f in basic_embedding_original basic_embedding_group_neg_shared cnn_embedding_original cnn_embedding_group_neg_shared; do python3 -c "
import json; j=json.load(open('$OUT/${TAG}_bench_$f.json')); print('$f', round(j['ms_per_step']*1e3,2), j['value'])"; done
cat $OUT/${TAG}_eval_bench.json | head -c 400; echo; cat $OUT/${TAG}_eval_ab.txt; head -12 $OUT/${TAG}_basic_embedding_original_kernel_trace.txt
