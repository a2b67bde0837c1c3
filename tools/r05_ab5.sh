#!/bin/bash
# ad-hoc A/B on the GPU box: large-batch lines (wide kernel)
B="--recall-epochs 0 --cpu-seconds 0"
for bs in 4096 8192 16384; do
python3 bench.py --large --batch $bs --steps 60 --warmup 6 $B 2>/dev/null | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('b$bs neg_shared skip-gram', round(j['ms_per_step']*1e3,1), 'us', j['roofline'].get('kernel'), round(j['roofline'].get('kernel_ms',0)*1e3,1))"
done
python3 bench.py --large --batch 8192 --scheme group_neg_shared --steps 60 --warmup 6 $B 2>/dev/null | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('b8192 group skip-gram', round(j['ms_per_step']*1e3,1), 'us')"
NNCF_WIDE_F16SPLIT=force python3 bench.py --large --batch 8192 --steps 60 --warmup 6 $B 2>/dev/null | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('b8192 unguarded', round(j['ms_per_step']*1e3,1), 'us')"
NNCF_WIDE_F16SPLIT=0 python3 bench.py --large --batch 8192 --steps 60 --warmup 6 $B 2>/dev/null | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('b8192 fp32 pipe', round(j['ms_per_step']*1e3,1), 'us')"
