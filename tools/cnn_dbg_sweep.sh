#!/bin/bash
# Times the conv kernels with parts of them switched off (NNCF_CNN_DBG bits: 1 no forward MFMA loop,
# 2 no forward staging, 4 no dX part, 8 no dWk part) -- results are wrong on purpose, timings only.
for d in ${@:-0 4 8 12}; do
NNCF_CNN_DBG=$d python bench.py --model cnn_embedding --scheme group_neg_shared --cpu-seconds 0 --steps 300 --warmup 50 2>/dev/null | tail -1 > gpurun_out/dbg_$d.json
python - <<P
import json
d=json.loads(open("gpurun_out/dbg_$d.json").read())
print($d, round(d["ms_per_step"],4), round(d["roofline"]["kernel_ms"],4), [ (k["kernel"],round(k["kernel_ms"],4)) for k in d["roofline_other_kernels"]][:1])
P
done
