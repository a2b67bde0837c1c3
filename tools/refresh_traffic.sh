#!/bin/bash
# profiles/traffic.json only (after a change of csrc/ that does not change what the kernels do -- a comment): the kernel traces and
# FETCH / WRITE passes it is made of; on the GPU box, from the repo root.  Copy gpurun_out/<tag>_traffic.json to profiles/traffic.json.
set -u
TAG=${1:-r06}
R=$PWD; OUT=$R/gpurun_out; export TMPDIR=/tmp
B="--recall-epochs 0"
run_prof () {
  local name=$1; shift
  local extra=$1; shift
  (cd /tmp && rocprofv3 --kernel-trace $extra -d $OUT/prof_$name -o p -- python3 $R/bench.py "$@" $B > $OUT/prof_$name.log 2>&1)
  python3 $R/tools/prof_summary.py $OUT/prof_$name/p_results.db $OUT/${TAG}_$name.txt
  rm -rf $OUT/prof_$name
}
run_prof basic_embedding_neg_shared_kernel_trace "--stats" --steps 780 --warmup 390 --cpu-seconds 0 --profile-launches 10
run_prof basic_embedding_neg_shared_pmc_fetch "--pmc FETCH_SIZE" --steps 390 --warmup 39 --cpu-seconds 0 --profile-launches 10
run_prof basic_embedding_neg_shared_pmc_write "--pmc WRITE_SIZE" --steps 390 --warmup 39 --cpu-seconds 0 --profile-launches 10
run_prof mf_neg_shared_kernel_trace "--stats" --model mf --steps 1170 --warmup 390 --cpu-seconds 0 --profile-launches 10
run_prof mf_neg_shared_pmc_fetch "--pmc FETCH_SIZE" --model mf --steps 390 --warmup 39 --cpu-seconds 0 --profile-launches 10
run_prof mf_neg_shared_pmc_write "--pmc WRITE_SIZE" --model mf --steps 390 --warmup 39 --cpu-seconds 0 --profile-launches 10
for c in FETCH_SIZE WRITE_SIZE; do
  n=$(echo $c | tr 'A-Z' 'a-z' | sed 's/_size//')
  (cd /tmp && rocprofv3 --kernel-trace --pmc $c -d $OUT/prof_evp -o p -- python3 $R/tools/eval_bench.py > /dev/null 2>&1)
  python3 $R/tools/prof_summary.py $OUT/prof_evp/p_results.db $OUT/${TAG}_eval_pmc_$n.txt; rm -rf $OUT/prof_evp
done
python3 tools/make_traffic.py $OUT $TAG $OUT/${TAG}_traffic.json > /dev/null
cp $OUT/${TAG}_traffic.json profiles/traffic.json
python3 bench.py 2>/dev/null | tail -1 > $OUT/${TAG}_bench_basic_embedding_neg_shared.json
python3 -c "
import json; j=json.load(open('$OUT/${TAG}_bench_basic_embedding_neg_shared.json')); print(j['ms_per_step'], j['roofline']['traffic'], j['roofline'].get('rocprof_kernel_ms'), j['evaluator']['traffic'])"
