#!/bin/bash
# A/B of library variants (VARIANT_FILES=eval tools/build_variant.sh) on the one-kernel evaluator: bash tools/ab_eval.sh "v1 v2 ..."
set -u
VARS=$1; shift
cp nncf_amd/libnncf_hip.so /tmp/main.so
for v in main $VARS; do
  if [ $v = main ]; then cp /tmp/main.so nncf_amd/libnncf_hip.so; else cp ab/libnncf_hip_$v.so nncf_amd/libnncf_hip.so; fi
  for rep in 1 2; do
    python3 tools/eval_bench.py 2>/dev/null | tail -1 > /tmp/abe.json
    python3 -c "
import json; j=json.load(open('/tmp/abe.json')); print('$v', 'fused pass %.4f ms (wall %.3f)' % (j['fused']['kernel_ms_per_pass'], j['fused']['pass_ms_wall']), 'frac %.3f' % j['roofline']['frac'])"
  done
done
cp /tmp/main.so nncf_amd/libnncf_hip.so
