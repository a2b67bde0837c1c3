#!/bin/bash
python -m pytest tests/test_eval_gpu.py -x -q 2>&1 | tail -6
for dbg in 0 1 2 3; do echo "dbg=$dbg"; NNCF_EVAL_DBG=$dbg python tools/eval_bench.py 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('  fused kernel ms/pass', round(d['fused']['kernel_ms_per_pass'],3), 'wall', round(d['fused']['pass_ms_wall'],3), ' two-kernel', round(d['two_kernel']['kernel_ms_per_pass'],3), 'wall', round(d['two_kernel']['pass_ms_wall'],3), 'frac', round(d['roofline']['frac'],3))"; done
