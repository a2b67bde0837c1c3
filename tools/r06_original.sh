#!/bin/bash
# config [0] (`original`) after a kernel change: the parity tests that cover it, the bench line, the kernel trace
set -u
R=$PWD; OUT=$R/gpurun_out; export TMPDIR=/tmp
timeout 900 python3 -m pytest tests/test_tower_gpu.py tests/test_schemes_gpu.py tests/test_trainer_gpu.py -m gpu -x -q 2>&1 | tail -5 > $OUT/orig_tests.txt
python3 bench.py --scheme original --cpu-seconds 0 --steps 780 --recall-epochs 0 2>/dev/null | tail -1 > $OUT/orig_bench.json
(cd /tmp && rocprofv3 --kernel-trace --stats -d $OUT/prof_orig -o p -- python3 $R/bench.py --scheme original --steps 390 --warmup 39 --cpu-seconds 0 --profile-launches 10 --recall-epochs 0 > $OUT/prof_orig.log 2>&1)
python3 $R/tools/prof_summary.py $OUT/prof_orig/p_results.db $OUT/orig_trace.txt
rm -rf $OUT/prof_orig
cat $OUT/orig_tests.txt; python3 -c "import json;d=json.load(open('$OUT/orig_bench.json'));print(d['ms_per_step'],d['value'])"; head -12 $OUT/orig_trace.txt
