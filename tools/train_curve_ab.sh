#!/bin/bash
# The training loop end to end under the default arithmetic (plain GEMMs of the plan on the fp16 pipe) and with fp32 MFMA only: train.py on the same seeded
# synthetic batches (64 samples, so the model can fit them), PlaneRecNet_101, batch 8, 600 iterations each; the console log (moving averages per 100 iterations).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for mode in default fp32; do
  echo "== $mode"
  D=/tmp/curve_$mode; rm -rf $D; mkdir -p $D
  if [ $mode = fp32 ]; then export PRN_SPLIT_GEMM=0; else unset PRN_SPLIT_GEMM; fi
  timeout 600 python train.py --config PlaneRecNet_101_config --dataset synthetic --batch_size 8 --save_folder $D/ --num_workers 0 --synthetic_size 64 \
     --max_iter 600 --reproductablity --no_autoscale --no_tensorboard --validation_epoch 100000 --save_interval 100000 --no_interrupt 2>&1 | grep -E "^\[|Begin|NaN|nan|rror|not supported" | cut -c1-200
done
