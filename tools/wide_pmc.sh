#!/bin/bash
# PMC passes of the large-batch step (b = 8,192, d = 128, mf neg_shared), one counter group per pass: bash tools/wide_pmc.sh r06 [bench args]
set -u
TAG=${1:-r06}; shift
R=$PWD
OUT=$R/gpurun_out
export TMPDIR=/tmp
F=$OUT/${TAG}_wide_pmc.txt
: > $F
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS" "SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_ANY SQ_LDS_ADDR_CONFLICT"; do
  (cd /tmp && rocprofv3 --kernel-trace --pmc $grp -d $OUT/prof_wp -o p -- python3 $R/bench.py --large --batch 8192 --steps 20 --warmup 4 --cpu-seconds 0 --recall-epochs 0 --profile-launches 10 "$@" > /dev/null 2>&1)
  echo "## --pmc $grp" >> $F
  python3 $R/tools/prof_summary.py $OUT/prof_wp/p_results.db /tmp/wp.txt > /dev/null 2>&1 && grep -i "fused_wide\|pair_wide\|^kernel\|^#" /tmp/wp.txt >> $F
  rm -rf $OUT/prof_wp
done
cat $F
