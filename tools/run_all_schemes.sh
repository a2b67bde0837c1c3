#!/bin/bash
# End-to-end check of main.py: every train_scheme x model on the synthetic CiteULike-shaped data,
# two epochs each; prints the last epoch line and the time per epoch.
for model in ${MODELS:-basic_embedding cnn_embedding mf}; do
for scheme in original neg_shared group_neg_shared presample group_sample sampled_neg_shared reverse; do
  loss=skip-gram
  echo "== $model $scheme"
  python main.py --data_name synthetic_citeulike --model_choice $model --conf_choice best --train_scheme $scheme \
     --eval_scheme whole@50 --param_dict "{'reset_after_getconf': True, 'optimizer': 'adam', 'max_epoch': 2}" 2>&1 \
     | grep -E "epoch 2|Training time|Error|error|Traceback" | cut -c1-230
done; done
