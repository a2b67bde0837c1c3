#!/bin/bash
# The subset of tools/profile_round.sh that profiles/traffic.json depends on (kernel trace + the two PMC passes of the
# headline and of the mf workload) and the bench lines that read it -- for a re-run after a source change that does not
# touch the other workloads' kernels.   usage: bash tools/profile_traffic.sh r02
set -u
TAG=${1:-r02}
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
python3 tools/make_traffic.py $OUT $TAG $OUT/${TAG}_traffic.json > /dev/null
cp $OUT/${TAG}_traffic.json profiles/traffic.json
python3 bench.py 2>/dev/null | tail -1 > $OUT/${TAG}_bench_basic_embedding_neg_shared.json
python3 bench.py --model mf --cpu-seconds 6 2>/dev/null | tail -1 > $OUT/${TAG}_bench_mf_neg_shared.json
python3 bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > $OUT/${TAG}_bench_driver_style_steps20.json
ls -la $OUT/${TAG}_traffic.json
