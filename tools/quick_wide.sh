set -x
timeout 1500 python -m pytest tests/test_tower_gpu.py tests/test_loss_golden_gpu.py tests/test_step_gpu.py -x -q 2>&1 | tail -8
timeout 300 python3 bench.py --scheme group_neg_shared --steps 1170 --cpu-seconds 0 --recall-epochs 0 2>/dev/null | tail -1 | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print('group log-loss', j['ms_per_step'])"
NNCF_TILE_STATS=launch timeout 300 python3 bench.py --scheme group_neg_shared --steps 1170 --cpu-seconds 0 --recall-epochs 0 2>/dev/null | tail -1 | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print('group log-loss, stats launch', j['ms_per_step'])"
timeout 300 python3 bench.py --model cnn_embedding --scheme group_neg_shared --steps 390 --warmup 39 --cpu-seconds 0 --recall-epochs 0 2>/dev/null | tail -1 | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print('cnn group', j['ms_per_step'])"
timeout 300 python3 bench.py --loss log-loss --steps 1170 --cpu-seconds 0 --recall-epochs 0 2>/dev/null | tail -1 | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print('neg_shared log-loss', j['ms_per_step'])"
