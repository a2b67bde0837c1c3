#!/bin/bash
# A/B of library variants (tools/build_variant.sh) on the headline step: bash tools/ab_headline.sh "v1 v2 ..." [bench args]
set -u
VARS=$1; shift
cp nncf_amd/libnncf_hip.so /tmp/main.so
for v in main $VARS; do
  if [ $v = main ]; then cp /tmp/main.so nncf_amd/libnncf_hip.so; else cp ab/libnncf_hip_$v.so nncf_amd/libnncf_hip.so; fi
  for rep in 1 2 3; do
  python3 bench.py --steps 1560 --warmup 390 --recall-epochs 0 --cpu-seconds 0 "$@" 2>/dev/null | tail -1 > /tmp/ab.json
  python3 -c "
import json; j=json.load(open('/tmp/ab.json')); r=j['roofline']; print('$v', 'step %.2f us' % (j['ms_per_step']*1e3), r['kernel'], '%.2f us' % (r['kernel_ms']*1e3))"
  done
done
cp /tmp/main.so nncf_amd/libnncf_hip.so
