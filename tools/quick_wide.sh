for scheme in group_neg_shared neg_shared original; do
echo "== cnn_embedding default preset, $scheme"
timeout 600 python main.py --data_name synthetic_citeulike --model_choice cnn_embedding --conf_choice default --train_scheme $scheme \
     --eval_scheme whole@50 --param_dict "{'max_epoch': 3}" 2>&1 | grep -E "epoch 3|epoch 1|Training time|Error|error|Traceback" | cut -c1-260
done
timeout 600 python3 bench.py --model cnn_embedding --scheme group_neg_shared --dim 50 --filter-lens 1,3,5 --conv-dropout 0.3 --dropout 0 --steps 390 --warmup 39 --cpu-seconds 6 2>&1 | tail -1 | cut -c1-1800
