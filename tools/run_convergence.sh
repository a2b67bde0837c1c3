#!/bin/bash
# 30 epochs (the demo scripts' max_epoch) of the three headline schemes on both text towers: final
# Recall@50 / MAP@50 on train items and on cold-start test items, seconds of training per epoch.
for model in basic_embedding cnn_embedding; do
for scheme in original neg_shared group_neg_shared; do
  echo "== $model $scheme"
  python main.py --data_name synthetic_citeulike --model_choice $model --conf_choice best --train_scheme $scheme \
     --eval_scheme whole@50 --param_dict "{'reset_after_getconf': True, 'optimizer': 'adam', 'max_epoch': 30, 'chop_size': 4}" 2>&1 \
     | grep -E "epoch (10|20|30) |Training time|Error|Traceback" | cut -c1-230
done; done
