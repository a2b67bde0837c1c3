#!/bin/bash
# PMC passes of the one-kernel evaluator (tools/eval_bench.py), one counter group per pass: bash tools/eval_pmc.sh r05
set -u
TAG=${1:-r05}
R=$PWD
OUT=$R/gpurun_out
export TMPDIR=/tmp
: > $OUT/${TAG}_eval_pmc.txt
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_F32" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_INSTS_VMEM_RD"; do
  (cd /tmp && rocprofv3 --kernel-trace --pmc $grp -d $OUT/prof_evp -o p -- python3 $R/tools/eval_bench.py > /dev/null 2>&1)
  echo "## --pmc $grp" >> $OUT/${TAG}_eval_pmc.txt
  python3 $R/tools/prof_summary.py $OUT/prof_evp/p_results.db /tmp/evp.txt > /dev/null 2>&1 && grep -i "eval_fused\|^kernel\|^#" /tmp/evp.txt >> $OUT/${TAG}_eval_pmc.txt
  rm -rf $OUT/prof_evp
done
cat $OUT/${TAG}_eval_pmc.txt
