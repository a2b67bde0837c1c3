#!/bin/bash
# MFMA / LDS counters of the cnn_embedding step: bash tools/pmc_mfma.sh r01
set -u
TAG=${1:-r01}
R=$PWD
OUT=$R/gpurun_out
export TMPDIR=/tmp
name=cnn_embedding_group_neg_shared_pmc_mfma
(cd /tmp && rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE -d $OUT/prof_$name -o p -- python3 $R/bench.py --model cnn_embedding --scheme group_neg_shared --steps 200 --warmup 20 --cpu-seconds 0 --profile-launches 10 > $OUT/prof_$name.log 2>&1)
python3 $R/tools/prof_summary.py $OUT/prof_$name/p_results.db $OUT/${TAG}_$name.txt
rm -rf $OUT/prof_$name
grep -E "cnn_conv|shared_tile" $OUT/${TAG}_$name.txt | cut -c1-130
