#!/bin/bash
# main.py with the demo scripts' loss settings (scripts/demos/run_*.sh): every loss x the three headline
# schemes on cnn_embedding, one epoch each.
for scheme in original neg_shared group_neg_shared; do
for loss in skip-gram log-loss max-margin mse; do
  extra="'neg_loss_weight': 128, 'loss_gamma': 10, 'learn_rate': 0.01"
  [ $loss = mse ] && extra="'neg_loss_weight': 8, 'loss_gamma': 10, 'learn_rate': 0.001"
  [ $loss = max-margin ] && extra="'neg_loss_weight': 128, 'loss_gamma': 0.1, 'learn_rate': 0.01"
  echo "== cnn_embedding $scheme $loss"
  python main.py --data_name synthetic_citeulike --model_choice cnn_embedding --conf_choice best --train_scheme $scheme \
     --eval_scheme whole@50 --param_dict "{'reset_after_getconf': True, 'optimizer': 'adam', 'max_epoch': 1, 'loss': '$loss', 'chop_size': 4, $extra}" 2>&1 \
     | grep -E "epoch 1|Training time|Error|error|Traceback" | cut -c1-200
done; done
