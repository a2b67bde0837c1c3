set -x
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -6
for i in 1 2 3; do timeout 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-seconds 0 --recall-epochs 0 2>/dev/null | tail -1 | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print('driver-style', j['ms_per_step'], j['value'])"; done
timeout 300 python3 bench.py --steps 1170 --cpu-seconds 0 --recall-epochs 0 2>/dev/null | tail -1 | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print('headline', j['ms_per_step'], j['value'])"
