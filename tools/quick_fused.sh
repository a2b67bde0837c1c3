export TMPDIR=/tmp
python -m pytest tests -m gpu -q 2>&1 | tail -25 > gpurun_out/r02_gputest_c.log
NNCF_FORCE_SHARDED=1 python bench.py --model mf --steps 780 --warmup 390 --cpu-seconds 0 --recall-epochs 0 2>gpurun_out/r02_q_sharded.err | tail -1 > gpurun_out/r02_q_mf_sharded_w1.json
NNCF_FORCE_SHARDED=1 python bench.py --large --steps 780 --warmup 390 --cpu-seconds 0 --recall-epochs 0 2>>gpurun_out/r02_q_sharded.err | tail -1 > gpurun_out/r02_q_large_sharded_w1.json
python bench.py --large --steps 780 --warmup 390 --cpu-seconds 0 --recall-epochs 0 2>/dev/null | tail -1 > gpurun_out/r02_q_large_plain.json
