set -x
timeout 900 python -m pytest tests/test_fused_gpu.py -x -q 2>&1 | tail -15
timeout 600 python -m pytest tests/test_tower_gpu.py tests/test_sharded_gpu.py -x -q 2>&1 | tail -5
timeout 300 python3 bench.py --steps 1170 --cpu-seconds 0 2>/dev/null | tail -1 | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print('headline', j['ms_per_step'], j['roofline'].get('kernel_ms'))"
timeout 300 python3 bench.py --model mf --steps 1170 --cpu-seconds 0 2>/dev/null | tail -1 | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print('mf', j['ms_per_step'], j['roofline'].get('kernel_ms'))"
for b in 8192 4096 2048; do
  timeout 300 python3 bench.py --large --batch $b --steps 60 --warmup 10 --cpu-seconds 0 2>/dev/null | tail -1 | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print('b$b', j['ms_per_step'], j['roofline'])"
done
