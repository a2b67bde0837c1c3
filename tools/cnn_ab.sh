#!/bin/bash
# Developer A/B of the conv pair (config [4]): kernel trace per NNCF_CNN_SPLIT setting.  bash tools/cnn_ab.sh "48 0 64 32"
for sp in ${1:-48 0}; do
  NNCF_CNN_SPLIT=$sp bash tools/quick_prof.sh cnn_split$sp --model cnn_embedding --scheme group_neg_shared --steps 390 --warmup 39 2>&1 | grep -E "metric|cnn_conv|tower_word|cnn_pool" | cut -c1-200
done
