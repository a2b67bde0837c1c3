#!/bin/bash
# Times tower_word_kernel at train_scheme = original with parts switched off (NNCF_WORD_DBG bits: 1 skip the word workgroups,
# 2 no dh row gathers, 4 no scan / postings walk, 8 no hot partials) -- results are wrong on purpose, timings only.
for d in ${@:-0 1 2 4 8 14}; do
NNCF_WORD_DBG=$d python bench.py --scheme ${SCHEME:-original} --cpu-seconds 0 --steps 390 --warmup 39 --recall-epochs 0 2>/dev/null | tail -1 > gpurun_out/wdbg_$d.json
python - <<P
import json
d=json.loads(open("gpurun_out/wdbg_$d.json").read())
print($d, round(d["ms_per_step"],4), round(d["roofline"]["kernel_ms"],4), [ (k["kernel"],round(k["kernel_ms"],4)) for k in d["roofline_other_kernels"]][:2])
P
done
