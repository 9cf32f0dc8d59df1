#!/bin/bash
# k_rollout_fnn_w: initial skew of the upper four waves (x 512 cycles) against us per step; VS_FNN_SHAPE=w
out=${1:-gpurun_out/s5}
mkdir -p $out
for sk in 0 2 4 6 8 12; do
  echo "skew $sk" >> $out/fnn_w_skew.txt
  VS_FNN_SKEW=$sk VS_FNN_SHAPE=w python scratch/r4_fnn.py >> $out/fnn_w_skew.txt 2>&1
done
