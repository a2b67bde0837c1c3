for v in 0 2 4 6; do
NNCF_WORD_DBG=$v python bench.py --model cnn_embedding --scheme group_neg_shared --cpu-seconds 0 --steps 390 --recall-epochs 0 2>/dev/null | tail -1 > gpurun_out/q_w$v.json
python -c "
import json; d=json.load(open('gpurun_out/q_w$v.json')); print($v, d['ms_per_step'], [(k['kernel'],round(k['kernel_ms']*1000,1)) for k in [d['roofline']]+d['roofline_other_kernels']])"
done
