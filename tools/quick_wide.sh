set -x
timeout 1200 python -m pytest tests/test_fused_gpu.py -x -q 2>&1 | tail -8
timeout 600 python -m pytest tests/test_sharded_gpu.py tests/test_step_gpu.py -x -q 2>&1 | tail -4
for b in 16384 8192; do
  timeout 300 python3 bench.py --large --batch $b --steps 40 --warmup 8 --cpu-seconds 0 --recall-epochs 0 2>&1 | tail -1 | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print('b$b', j['ms_per_step'], j['value'], j['roofline']['frac'], j['roofline']['kernel'])"
done
