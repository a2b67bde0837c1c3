#!/bin/bash
# round-6: epoch parity test + the 30-epoch convergence record (Keras Adam, the optimiser of bench.py's recall_at_50_trained)
set -u
OUT=gpurun_out
python3 -m pytest tests/test_epoch_parity_gpu.py -x -q 2>&1 | tail -15 > $OUT/epoch_parity_test.txt
python3 tools/convergence_parity.py --epochs 30 --optimizer adam --eval-epochs 1,2,3,5,10,15,20,25,30 --out $OUT/r06_convergence_parity.json > $OUT/conv30.log 2>&1
python3 tools/convergence_parity.py --epochs 30 --optimizer sgd --learn-rate 0.5 --eval-epochs 1,2,5,10,20,30 --out $OUT/r06_convergence_parity_sgd.json > $OUT/conv30_sgd.log 2>&1
cat $OUT/epoch_parity_test.txt; tail -12 $OUT/conv30.log; tail -8 $OUT/conv30_sgd.log
