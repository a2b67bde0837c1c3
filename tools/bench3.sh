#!/bin/bash
# quick per-scheme step time + per-kernel brackets (developer tool): bash tools/bench3.sh [extra bench args]
for s in neg_shared original group_neg_shared; do
  python bench.py --scheme $s --steps 390 --warmup 100 --cpu-seconds 0 --recall-epochs 0 "$@" 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$s', round(j['ms_per_step']*1e3,2), [(r['kernel'], round(r['kernel_ms']*1e3,2)) for r in [j['roofline']]+j['roofline_other_kernels']])"
done
