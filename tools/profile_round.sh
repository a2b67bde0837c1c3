#!/bin/bash
# Collects the round's evidence on the GPU box: bench JSON lines + rocprofv3 summaries.
# usage (from the repo root, on the GPU box): bash tools/profile_round.sh r01
set -u
TAG=${1:-r01}
R=$PWD
OUT=$R/gpurun_out
export TMPDIR=/tmp
run_prof () {  # name, extra rocprof args, bench args...
  local name=$1; shift
  local extra=$1; shift
  (cd /tmp && rocprofv3 --kernel-trace $extra -d $OUT/prof_$name -o p -- python3 $R/bench.py "$@" > $OUT/prof_$name.log 2>&1)
  python3 $R/tools/prof_summary.py $OUT/prof_$name/p_results.db $OUT/${TAG}_$name.txt
  rm -rf $OUT/prof_$name
}
python3 bench.py 2>/dev/null | tail -1 > $OUT/${TAG}_bench_basic_embedding_neg_shared.json
python3 bench.py --scheme group_neg_shared --cpu-seconds 6 --steps 1170 2>/dev/null | tail -1 > $OUT/${TAG}_bench_basic_embedding_group_neg_shared.json
python3 bench.py --scheme original --cpu-seconds 6 --steps 780 2>/dev/null | tail -1 > $OUT/${TAG}_bench_basic_embedding_original.json
python3 bench.py --optimizer adam --cpu-seconds 0 --steps 780 2>/dev/null | tail -1 > $OUT/${TAG}_bench_basic_embedding_neg_shared_adam.json
python3 bench.py --model cnn_embedding --scheme group_neg_shared --cpu-seconds 8 --steps 780 2>/dev/null | tail -1 > $OUT/${TAG}_bench_cnn_embedding_group_neg_shared.json
python3 bench.py --model cnn_embedding --scheme group_neg_shared --optimizer adam --cpu-seconds 0 --steps 780 2>/dev/null | tail -1 > $OUT/${TAG}_bench_cnn_embedding_group_neg_shared_adam.json
python3 bench.py --model mf --cpu-seconds 6 2>/dev/null | tail -1 > $OUT/${TAG}_bench_mf_neg_shared.json
python3 bench.py --model mf --scheme original --cpu-seconds 6 2>/dev/null | tail -1 > $OUT/${TAG}_bench_mf_original.json
python3 bench.py --model mf --scheme group_neg_shared --cpu-seconds 6 2>/dev/null | tail -1 > $OUT/${TAG}_bench_mf_group_neg_shared.json
python3 bench.py --model mf --large --steps 1170 2>/dev/null | tail -1 > $OUT/${TAG}_bench_large_mf_neg_shared.json
python3 bench.py --model mf --large --scheme original --steps 1170 2>/dev/null | tail -1 > $OUT/${TAG}_bench_large_mf_original.json
run_prof basic_embedding_neg_shared_kernel_trace "--stats" --steps 780 --warmup 390 --cpu-seconds 0 --profile-launches 10
run_prof basic_embedding_neg_shared_pmc_fetch "--pmc FETCH_SIZE" --steps 390 --warmup 39 --cpu-seconds 0 --profile-launches 10
run_prof basic_embedding_neg_shared_pmc_write "--pmc WRITE_SIZE" --steps 390 --warmup 39 --cpu-seconds 0 --profile-launches 10
run_prof basic_embedding_group_neg_shared_kernel_trace "--stats" --scheme group_neg_shared --steps 780 --warmup 390 --cpu-seconds 0 --profile-launches 10
run_prof basic_embedding_original_kernel_trace "--stats" --scheme original --steps 390 --warmup 39 --cpu-seconds 0 --profile-launches 10
run_prof cnn_embedding_group_neg_shared_kernel_trace "--stats" --model cnn_embedding --scheme group_neg_shared --steps 390 --warmup 39 --cpu-seconds 0 --profile-launches 10
run_prof mf_neg_shared_kernel_trace "--stats" --model mf --steps 1170 --warmup 390 --cpu-seconds 0 --profile-launches 10
run_prof mf_neg_shared_pmc_fetch "--pmc FETCH_SIZE" --model mf --steps 390 --warmup 39 --cpu-seconds 0 --profile-launches 10
run_prof mf_neg_shared_pmc_write "--pmc WRITE_SIZE" --model mf --steps 390 --warmup 39 --cpu-seconds 0 --profile-launches 10
run_prof mf_original_kernel_trace "--stats" --model mf --scheme original --steps 1170 --warmup 390 --cpu-seconds 0 --profile-launches 10
run_prof mf_group_neg_shared_kernel_trace "--stats" --model mf --scheme group_neg_shared --steps 1170 --warmup 390 --cpu-seconds 0 --profile-launches 10
bash tools/pmc_mfma.sh $TAG > /dev/null 2>&1
python3 tools/make_traffic.py $OUT $TAG $OUT/${TAG}_traffic.json > /dev/null
ls -la $OUT/${TAG}_*
