#!/bin/bash
B="--recall-epochs 0 --cpu-seconds 0"
for sc in neg_shared group_neg_shared; do
python3 bench.py --large --batch 8192 --loss mse --scheme $sc --steps 60 --warmup 6 $B 2>/dev/null | tail -1 | python3 -c "
import sys,json; j=json.loads(sys.stdin.read()); print('b8192 $sc mse', round(j['ms_per_step']*1e3,1), 'us', j['roofline']['kernel'])"
done
