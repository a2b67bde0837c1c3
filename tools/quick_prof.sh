#!/bin/bash
# One kernel-trace of a bench configuration: bash tools/quick_prof.sh <name> [bench args...]
set -u
NAME=$1; shift
R=$PWD
OUT=$R/gpurun_out
export TMPDIR=/tmp
(cd /tmp && rocprofv3 --kernel-trace --stats -d $OUT/prof_$NAME -o p -- python3 $R/bench.py "$@" --cpu-seconds 0 --profile-launches 10 > $OUT/prof_$NAME.log 2>&1)
python3 $R/tools/prof_summary.py $OUT/prof_$NAME/p_results.db $OUT/quick_$NAME.txt
rm -rf $OUT/prof_$NAME
tail -1 $OUT/prof_$NAME.log | cut -c1-400
head -30 $OUT/quick_$NAME.txt
