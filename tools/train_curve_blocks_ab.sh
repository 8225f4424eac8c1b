#!/bin/bash
# The training loop end to end with the backbone blocks as ONE C call each way (default) and as the operator sequence (PRN_BLOCKS=0), and with the blocks' producer ->
# BatchNorm hand-overs off (PRN_BLOCK_HANDOVER=0, bit-identical to the operator sequence per block): train.py on the same seeded synthetic batches (64 samples),
# PlaneRecNet_101, batch 8, 600 iterations each; the console log (moving averages per 100 iterations).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for mode in blocks operators blocks_no_handover; do
  echo "== $mode"
  D=/tmp/curve_$mode; rm -rf $D; mkdir -p $D
  unset PRN_BLOCKS PRN_BLOCK_HANDOVER
  [ $mode = operators ] && export PRN_BLOCKS=0
  [ $mode = blocks_no_handover ] && export PRN_BLOCK_HANDOVER=0
  timeout 600 python train.py --config PlaneRecNet_101_config --dataset synthetic --batch_size 8 --save_folder $D/ --num_workers 0 --synthetic_size 64 \
     --max_iter 600 --reproductablity --no_autoscale --no_tensorboard --validation_epoch 100000 --save_interval 100000 --no_interrupt 2>&1 | grep -E "^\[|Begin|NaN|nan|rror|not supported" | cut -c1-200
done
