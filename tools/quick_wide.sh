set -x
timeout 1200 python -m pytest tests/test_tower_gpu.py tests/test_fused_gpu.py tests/test_loss_golden_gpu.py -x -q 2>&1 | tail -8
timeout 300 python3 bench.py --steps 1170 --cpu-seconds 0 --recall-epochs 0 2>/dev/null | tail -1 | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print('headline', j['ms_per_step'], j['roofline'].get('kernel_ms'))"
timeout 300 python3 bench.py --scheme group_neg_shared --steps 1170 --cpu-seconds 0 --recall-epochs 0 2>/dev/null | tail -1 | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print('group', j['ms_per_step'], j['roofline'].get('kernel_ms'))"
timeout 300 python3 bench.py --model mf --steps 1170 --cpu-seconds 0 --recall-epochs 0 2>/dev/null | tail -1 | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print('mf', j['ms_per_step'], j['roofline'].get('kernel_ms'))"
NNCF_DEBUG_TIMING=1 python3 bench.py --steps 200 --warmup 50 --cpu-seconds 0 --recall-epochs 0 --dump-stamps gpurun_out/st_tower.npy > /dev/null 2>&1; python tools/fused_timing.py --load gpurun_out/st_tower.npy --split 4 32 > gpurun_out/tt.log 2>&1; grep -A10 "^U-pass\|^V-pass" gpurun_out/tt.log
