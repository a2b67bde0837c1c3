#!/bin/bash
# Collects the round's evidence on the GPU box: bench JSON lines + rocprofv3 summaries.
# usage (from the repo root, on the GPU box): bash tools/profile_round.sh r04
set -u
TAG=${1:-r06}
R=$PWD
OUT=$R/gpurun_out
export TMPDIR=/tmp
B="--recall-epochs 0"
run_prof () {  # name, extra rocprof args, bench args...
  local name=$1; shift
  local extra=$1; shift
  (cd /tmp && rocprofv3 --kernel-trace $extra -d $OUT/prof_$name -o p -- python3 $R/bench.py "$@" $B > $OUT/prof_$name.log 2>&1)
  python3 $R/tools/prof_summary.py $OUT/prof_$name/p_results.db $OUT/${TAG}_$name.txt
  rm -rf $OUT/prof_$name
}
# kernel traces + PMC passes first: make_traffic.py turns them into profiles/traffic.json, which the bench lines below read
run_prof basic_embedding_neg_shared_kernel_trace "--stats" --steps 780 --warmup 390 --cpu-seconds 0 --profile-launches 10
run_prof basic_embedding_neg_shared_pmc_fetch "--pmc FETCH_SIZE" --steps 390 --warmup 39 --cpu-seconds 0 --profile-launches 10
run_prof basic_embedding_neg_shared_pmc_write "--pmc WRITE_SIZE" --steps 390 --warmup 39 --cpu-seconds 0 --profile-launches 10
run_prof basic_embedding_group_neg_shared_kernel_trace "--stats" --scheme group_neg_shared --steps 780 --warmup 390 --cpu-seconds 0 --profile-launches 10
run_prof basic_embedding_original_kernel_trace "--stats" --scheme original --steps 390 --warmup 39 --cpu-seconds 0 --profile-launches 10
run_prof cnn_embedding_group_neg_shared_kernel_trace "--stats" --model cnn_embedding --scheme group_neg_shared --steps 390 --warmup 39 --cpu-seconds 0 --profile-launches 10
run_prof cnn_embedding_group_neg_shared_pmc_mfma "--pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE" --model cnn_embedding --scheme group_neg_shared --steps 195 --warmup 39 --cpu-seconds 0 --profile-launches 10
run_prof cnn_embedding_original_kernel_trace "--stats" --model cnn_embedding --scheme original --steps 100 --warmup 20 --cpu-seconds 0 --profile-launches 10
run_prof mf_neg_shared_kernel_trace "--stats" --model mf --steps 1170 --warmup 390 --cpu-seconds 0 --profile-launches 10
run_prof mf_neg_shared_pmc_fetch "--pmc FETCH_SIZE" --model mf --steps 390 --warmup 39 --cpu-seconds 0 --profile-launches 10
run_prof mf_neg_shared_pmc_write "--pmc WRITE_SIZE" --model mf --steps 390 --warmup 39 --cpu-seconds 0 --profile-launches 10
run_prof mf_neg_shared_pmc_mfma "--pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE" --model mf --steps 390 --warmup 39 --cpu-seconds 0 --profile-launches 10
run_prof mf_original_kernel_trace "--stats" --model mf --scheme original --steps 1170 --warmup 390 --cpu-seconds 0 --profile-launches 10
run_prof mf_group_neg_shared_kernel_trace "--stats" --model mf --scheme group_neg_shared --steps 1170 --warmup 390 --cpu-seconds 0 --profile-launches 10
run_prof large_mf_b8192_kernel_trace "--stats" --large --batch 8192 --steps 40 --warmup 8 --cpu-seconds 0 --profile-launches 10
run_prof large_mf_b16384_kernel_trace "--stats" --large --batch 16384 --steps 20 --warmup 4 --cpu-seconds 0 --profile-launches 5
run_prof large_mf_b8192_logloss_kernel_trace "--stats" --large --batch 8192 --loss log-loss --steps 40 --warmup 8 --cpu-seconds 0 --profile-launches 10
run_prof large_mf_b8192_logloss_pmc_mfma "--pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" --large --batch 8192 --loss log-loss --steps 20 --warmup 4 --cpu-seconds 0 --profile-launches 10
run_prof large_mf_b8192_pmc_mfma "--pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" --large --batch 8192 --steps 20 --warmup 4 --cpu-seconds 0 --profile-launches 10
(cd /tmp && NNCF_FORCE_SHARDED=1 rocprofv3 --kernel-trace --stats -d $OUT/prof_sh -o p -- python3 $R/bench.py --model mf --steps 780 --warmup 390 --cpu-seconds 0 $B > $OUT/prof_sh.log 2>&1)
python3 $R/tools/prof_summary.py $OUT/prof_sh/p_results.db $OUT/${TAG}_mf_sharded_world1_kernel_trace.txt; rm -rf $OUT/prof_sh
for c in FETCH_SIZE WRITE_SIZE; do   # the evaluator's traffic (traffic.json: evaluator/...)
  n=$(echo $c | tr 'A-Z' 'a-z' | sed 's/_size//')
  (cd /tmp && rocprofv3 --kernel-trace --pmc $c -d $OUT/prof_evp -o p -- python3 $R/tools/eval_bench.py > /dev/null 2>&1)
  python3 $R/tools/prof_summary.py $OUT/prof_evp/p_results.db $OUT/${TAG}_eval_pmc_$n.txt; rm -rf $OUT/prof_evp
done
python3 tools/make_traffic.py $OUT $TAG $OUT/${TAG}_traffic.json > /dev/null
cp $OUT/${TAG}_traffic.json profiles/traffic.json
# bench lines (the headline first: full default run incl. cpu_baseline and the trained-Recall side run)
python3 bench.py 2>/dev/null | tail -1 > $OUT/${TAG}_bench_basic_embedding_neg_shared.json
python3 bench.py --scheme group_neg_shared --cpu-seconds 6 --steps 1170 2>/dev/null | tail -1 > $OUT/${TAG}_bench_basic_embedding_group_neg_shared.json
python3 bench.py --scheme original --cpu-seconds 6 --steps 780 2>/dev/null | tail -1 > $OUT/${TAG}_bench_basic_embedding_original.json
python3 bench.py --optimizer adam --cpu-seconds 0 --steps 780 $B 2>/dev/null | tail -1 > $OUT/${TAG}_bench_basic_embedding_neg_shared_adam.json
NNCF_SHARED_PATH=tile python3 bench.py --cpu-seconds 0 --steps 780 $B 2>/dev/null | tail -1 > $OUT/${TAG}_bench_basic_embedding_neg_shared_tilepath.json
python3 bench.py --model cnn_embedding --scheme group_neg_shared --cpu-seconds 8 --steps 780 2>/dev/null | tail -1 > $OUT/${TAG}_bench_cnn_embedding_group_neg_shared.json
python3 bench.py --model mf --cpu-seconds 6 2>/dev/null | tail -1 > $OUT/${TAG}_bench_mf_neg_shared.json
NNCF_SHARED_PATH=tile python3 bench.py --model mf --cpu-seconds 0 $B 2>/dev/null | tail -1 > $OUT/${TAG}_bench_mf_neg_shared_tilepath.json
python3 bench.py --model mf --scheme original --cpu-seconds 6 $B 2>/dev/null | tail -1 > $OUT/${TAG}_bench_mf_original.json
python3 bench.py --model mf --scheme group_neg_shared --cpu-seconds 6 $B 2>/dev/null | tail -1 > $OUT/${TAG}_bench_mf_group_neg_shared.json
python3 bench.py --large --steps 1170 $B 2>/dev/null | tail -1 > $OUT/${TAG}_bench_large_mf_neg_shared.json
python3 bench.py --large --device-links 1e9 --steps 1170 --cpu-seconds 0 $B 2>/dev/null | tail -1 > $OUT/${TAG}_bench_large_mf_neg_shared_devicelinks1e9.json
python3 bench.py --large --scheme original --steps 1170 $B 2>/dev/null | tail -1 > $OUT/${TAG}_bench_large_mf_original.json
python3 bench.py --large --batch 8192 --steps 100 --warmup 10 $B 2>/dev/null | tail -1 > $OUT/${TAG}_bench_large_mf_neg_shared_b8192.json
# the same with the large-batch kernel's products on the fp32 matrix pipe (round 4 moved them to the f16 pipe: devutil.h split_h)
NNCF_WIDE_F16SPLIT=0 python3 bench.py --large --batch 8192 --steps 100 --warmup 10 --cpu-seconds 0 $B 2>/dev/null | tail -1 > $OUT/${TAG}_bench_large_mf_neg_shared_b8192_fp32pipe.json
NNCF_WIDE_F16SPLIT=0 python3 bench.py --large --batch 8192 --loss log-loss --steps 100 --warmup 10 --cpu-seconds 0 $B 2>/dev/null | tail -1 > $OUT/${TAG}_bench_large_mf_neg_shared_log-loss_b8192_fp32pipe.json
python3 bench.py --large --batch 4096 --steps 100 --warmup 10 $B 2>/dev/null | tail -1 > $OUT/${TAG}_bench_large_mf_neg_shared_b4096.json
python3 bench.py --large --batch 16384 --steps 50 --warmup 5 $B 2>/dev/null | tail -1 > $OUT/${TAG}_bench_large_mf_neg_shared_b16384.json
for l in log-loss max-margin; do for sc in neg_shared group_neg_shared; do
python3 bench.py --large --batch 8192 --loss $l --scheme $sc --steps 100 --warmup 10 --cpu-seconds 0 $B 2>/dev/null | tail -1 > $OUT/${TAG}_bench_large_mf_${sc}_${l}_b8192.json
done; done
python3 bench.py --large --batch 4096 --loss log-loss --steps 100 --warmup 10 --cpu-seconds 0 $B 2>/dev/null | tail -1 > $OUT/${TAG}_bench_large_mf_neg_shared_log-loss_b4096.json
python3 bench.py --large --batch 16384 --loss log-loss --steps 50 --warmup 5 --cpu-seconds 0 $B 2>/dev/null | tail -1 > $OUT/${TAG}_bench_large_mf_neg_shared_log-loss_b16384.json
# the 32768-entry plan (sort keys in global memory): config [3]'s "b >~ 16K per GPU" regime
python3 bench.py --large --batch 32768 --steps 20 --warmup 3 --cpu-seconds 0 $B 2>/dev/null | tail -1 > $OUT/${TAG}_bench_large_mf_neg_shared_b32768.json
python3 bench.py --large --batch 32768 --loss log-loss --steps 20 --warmup 3 --cpu-seconds 0 $B 2>/dev/null | tail -1 > $OUT/${TAG}_bench_large_mf_neg_shared_log-loss_b32768.json
python3 bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > $OUT/${TAG}_bench_driver_style_steps20.json
python3 bench.py --model cnn_embedding --scheme group_neg_shared --dim 50 --filter-lens 1,3,5 --conv-dropout 0.3 --dropout 0 --cpu-seconds 6 --steps 780 2>/dev/null | tail -1 > $OUT/${TAG}_bench_cnn_embedding_default_preset_group_neg_shared.json
NNCF_TILE_STATS=launch python3 bench.py --scheme group_neg_shared --cpu-seconds 0 --steps 1170 $B 2>/dev/null | tail -1 > $OUT/${TAG}_bench_basic_embedding_group_neg_shared_statslaunch.json
(cd /tmp && rocprofv3 --kernel-trace -d $OUT/prof_tl -o p -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --cpu-seconds 0 $B --profile-launches 0 > /dev/null 2>&1); python3 tools/prof_timeline.py $OUT/prof_tl/p_results.db 130 > $OUT/${TAG}_timeline_steps20.txt; rm -rf $OUT/prof_tl
NNCF_FORCE_SHARDED=1 python3 bench.py --model mf --steps 1170 --cpu-seconds 0 $B 2>/dev/null | tail -1 > $OUT/${TAG}_bench_mf_sharded_world1.json
NNCF_FORCE_SHARDED=1 python3 bench.py --large --steps 1170 --cpu-seconds 0 $B 2>/dev/null | tail -1 > $OUT/${TAG}_bench_large_mf_sharded_world1.json
NNCF_FORCE_SHARDED=1 python3 bench.py --large --batch 8192 --steps 100 --warmup 10 --cpu-seconds 0 $B 2>/dev/null | tail -1 > $OUT/${TAG}_bench_large_mf_sharded_world1_b8192.json
NNCF_FORCE_SHARDED=1 python3 bench.py --model mf --scaling strong --steps 1170 --cpu-seconds 0 $B 2>/dev/null | tail -1 > $OUT/${TAG}_bench_mf_sharded_strong_world1.json
NNCF_FORCE_SHARDED=1 python3 bench.py --steps 1170 --cpu-seconds 6 $B 2>/dev/null | tail -1 > $OUT/${TAG}_bench_basic_embedding_data_parallel_world1.json
# ONE rank over RCCL with the exchanges kept real (NNCF_SHARD_NOALIAS=1; the rank's own share through ncclSend / ncclRecv too): the
# native chunk runner (default transport on an nccl process group) and torch.distributed's collectives
RC="NNCF_FORCE_SHARDED=1 NNCF_SHARD_NOALIAS=1 NNCF_COMM_SELF_VIA_RCCL=1"
env $RC python3 bench.py --model mf --steps 1170 --cpu-seconds 0 $B 2>/dev/null | tail -1 > $OUT/${TAG}_bench_mf_sharded_world1_rccl_native.json
env $RC NNCF_SHARD_TRANSPORT=torch python3 bench.py --model mf --steps 1170 --cpu-seconds 0 $B 2>/dev/null | tail -1 > $OUT/${TAG}_bench_mf_sharded_world1_rccl_torch.json
env $RC python3 bench.py --model mf --scaling strong --steps 1170 --cpu-seconds 0 $B 2>/dev/null | tail -1 > $OUT/${TAG}_bench_mf_sharded_strong_world1_rccl_native.json
env $RC python3 bench.py --large --batch 8192 --steps 100 --warmup 10 --cpu-seconds 0 $B 2>/dev/null | tail -1 > $OUT/${TAG}_bench_large_mf_sharded_world1_b8192_rccl_native.json
env $RC python3 bench.py --steps 1170 --cpu-seconds 0 $B 2>/dev/null | tail -1 > $OUT/${TAG}_bench_basic_embedding_data_parallel_world1_rccl_native.json
NNCF_FORCE_SHARDED=1 python3 bench.py --model mf --scaling strong --loss log-loss --steps 1170 --cpu-seconds 0 $B 2>/dev/null | tail -1 > $OUT/${TAG}_bench_mf_sharded_strong_logloss_world1.json
python3 bench.py --model cnn_embedding --scheme original --cpu-seconds 0 --steps 200 --warmup 20 $B 2>/dev/null | tail -1 > $OUT/${TAG}_bench_cnn_embedding_original.json
python3 bench.py --model cnn_embedding --scheme neg_shared --cpu-seconds 0 --steps 390 $B 2>/dev/null | tail -1 > $OUT/${TAG}_bench_cnn_embedding_neg_shared.json
NNCF_CNN_DBG=64 python3 tools/cnn_timing.py > $OUT/${TAG}_cnn_phase_stamps.txt 2>/dev/null
python3 tools/sampler_bench.py 2>/dev/null | tail -1 > $OUT/${TAG}_sampler_bench.json
# round 5: the one-kernel evaluator (whole@50 pass, fused vs predict + topk), its kernel trace, the register audit of the shipped
# library, the group_neg_shared pointwise large-batch line (the positive-count tile) and the f16-plane guard's A/B
python3 tools/eval_bench.py 2>/dev/null | tail -1 > $OUT/${TAG}_eval_bench.json
python3 tools/eval_bench.py 128 2>/dev/null | tail -1 > $OUT/${TAG}_eval_bench_d128.json
(cd /tmp && rocprofv3 --kernel-trace --stats -d $OUT/prof_ev -o p -- python3 $R/tools/eval_bench.py > /dev/null 2>&1); python3 tools/prof_summary.py $OUT/prof_ev/p_results.db $OUT/${TAG}_eval_kernel_trace.txt; rm -rf $OUT/prof_ev
python3 tools/isa_resources.py --out $OUT/${TAG}_isa_resources.txt > /dev/null
for l in skip-gram mse; do
python3 bench.py --large --batch 8192 --scheme group_neg_shared --loss $l --steps 100 --warmup 10 --cpu-seconds 0 $B 2>/dev/null | tail -1 > $OUT/${TAG}_bench_large_mf_group_neg_shared_${l}_b8192.json
done
python3 bench.py --large --batch 8192 --loss mse --steps 100 --warmup 10 --cpu-seconds 0 $B 2>/dev/null | tail -1 > $OUT/${TAG}_bench_large_mf_neg_shared_mse_b8192.json
python3 tools/fused_timing.py skip-gram 8192 128 2>/dev/null | tail -19 > $OUT/${TAG}_wide_phase_stamps.txt
run_prof large_mf_group_b8192_kernel_trace "--stats" --large --batch 8192 --scheme group_neg_shared --loss skip-gram --steps 40 --warmup 8 --cpu-seconds 0 --profile-launches 10
NNCF_WIDE_F16SPLIT=force python3 bench.py --large --batch 8192 --steps 100 --warmup 10 --cpu-seconds 0 $B 2>/dev/null | tail -1 > $OUT/${TAG}_bench_large_mf_neg_shared_b8192_unguarded.json
NNCF_DP_NATIVE=0 NNCF_FORCE_SHARDED=1 python3 bench.py --steps 1170 --cpu-seconds 0 $B 2>/dev/null | tail -1 > $OUT/${TAG}_bench_basic_embedding_data_parallel_world1_pythonloop.json
for sp in 0 48 64; do NNCF_CNN_SPLIT=$sp bash tools/quick_prof.sh ${TAG}_cnn_split$sp --model cnn_embedding --scheme group_neg_shared --steps 390 --warmup 39 > /dev/null 2>&1; mv $OUT/quick_${TAG}_cnn_split$sp.txt $OUT/${TAG}_cnn_split${sp}_kernel_trace.txt; done
ls -la $OUT/${TAG}_*
