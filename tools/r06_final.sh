#!/bin/bash
# round-6 final refresh on the shipped tree: the passes profiles/traffic.json is made of (its source hash must be the tree's),
# the headline bench lines, the full GPU suite
set -u
TAG=r06
R=$PWD
OUT=$R/gpurun_out
export TMPDIR=/tmp
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
python3 bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > $OUT/${TAG}_bench_driver_style_steps20.json
python3 bench.py --model mf --cpu-seconds 6 2>/dev/null | tail -1 > $OUT/${TAG}_bench_mf_neg_shared.json
python3 bench.py --large --batch 8192 --steps 100 --warmup 10 $B 2>/dev/null | tail -1 > $OUT/${TAG}_bench_large_mf_neg_shared_b8192.json
python3 -m pytest tests -m gpu -q 2>&1 | tail -3 > $OUT/${TAG}_gpu_suite.txt
cat $OUT/${TAG}_gpu_suite.txt
python3 -c "
import json
for f in ('basic_embedding_neg_shared','driver_style_steps20','mf_neg_shared','large_mf_neg_shared_b8192'):
    j=json.load(open('$OUT/${TAG}_bench_%s.json'%f)); r=j['roofline']; print(f, round(j['ms_per_step']*1e3,2), r['kernel'], round(r['kernel_ms']*1e3,2), round(r['frac'],4), r.get('traffic'))
"
