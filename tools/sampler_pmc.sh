#!/bin/bash
# PMC passes of the device sampler (tools/sampler_bench.py), one counter group per pass (separate --pmc runs, kernel trace only):
#   bash tools/sampler_pmc.sh r05      (on the GPU box, from the repo root)
set -u
TAG=${1:-r05}
R=$PWD
OUT=$R/gpurun_out
export TMPDIR=/tmp
: > $OUT/${TAG}_sampler_pmc.txt
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_INSTS_VALU_INT64"; do
  (cd /tmp && rocprofv3 --kernel-trace --pmc $grp -d $OUT/prof_smp -o p -- python3 $R/tools/sampler_bench.py > /dev/null 2>&1)
  echo "## --pmc $grp" >> $OUT/${TAG}_sampler_pmc.txt
  python3 $R/tools/prof_summary.py $OUT/prof_smp/p_results.db /tmp/smp.txt > /dev/null 2>&1 && grep -i "sample_kernel\|^kernel\|^#" /tmp/smp.txt >> $OUT/${TAG}_sampler_pmc.txt
  rm -rf $OUT/prof_smp
done
cat $OUT/${TAG}_sampler_pmc.txt
