set -x
timeout 1200 python -m pytest tests/test_tower_gpu.py tests/test_fused_gpu.py -x -q 2>&1 | tail -8
timeout 300 python3 bench.py --steps 1170 --cpu-seconds 0 --recall-epochs 0 2>/dev/null | tail -1 | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print('headline', j['ms_per_step'], [(o['kernel'], round(o['kernel_ms']*1e3,2)) for o in [j['roofline']]+j.get('roofline_other_kernels',[])])"
NNCF_TOWER_ROLEA=bwd timeout 300 python3 bench.py --steps 1170 --cpu-seconds 0 --recall-epochs 0 2>/dev/null | tail -1 | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print('headline rolea=bwd', j['ms_per_step'])"
timeout 300 python3 bench.py --scheme group_neg_shared --loss skip-gram --steps 1170 --cpu-seconds 0 --recall-epochs 0 2>/dev/null | tail -1 | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print('group skip-gram', j['ms_per_step'])"
