#!/bin/bash
# A/B of library variants (tools/build_variant.sh) on the large-batch step: bash tools/ab_wide.sh "v1 v2 ..." [bench args]
set -u
VARS=$1; shift
cp nncf_amd/libnncf_hip.so /tmp/main.so
for v in main $VARS; do
  if [ $v = main ]; then cp /tmp/main.so nncf_amd/libnncf_hip.so; else cp ab/libnncf_hip_$v.so nncf_amd/libnncf_hip.so; fi
  for rep in 1 2; do
  NNCF_WIDE_F16SPLIT=${F16SPLIT:-1} python3 bench.py --large --batch 8192 --steps 100 --warmup 10 --recall-epochs 0 --cpu-seconds 0 "$@" 2>/dev/null | tail -1 > /tmp/ab.json
  python3 -c "
import json; j=json.load(open('/tmp/ab.json')); r=j['roofline']; print('$v', 'step %.1f us' % (j['ms_per_step']*1e3), r['kernel'], '%.1f us' % (r['kernel_ms']*1e3))"
  done
done
cp /tmp/main.so nncf_amd/libnncf_hip.so
