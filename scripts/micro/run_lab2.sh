#!/bin/bash
# lab run 2: vector epilogue (code 0 / 21 / 22 / 122) against the scalar one (same codes under FRCNN_SCALAR_EPILOGUE=1)
cd "$(dirname "$0")/../.."
O=gpurun_out/lab2; mkdir -p $O
B=scripts/micro/_bin
$B/conv_lab time trunk 0,122 > $O/time_trunk_vec.txt 2>&1
FRCNN_SCALAR_EPILOGUE=1 $B/conv_lab time trunk 0 > $O/time_trunk_scalar.txt 2>&1
$B/conv_lab time head 0,21,22 > $O/time_head_vec.txt 2>&1
for L in s4_2c s3_2c s2_2c s2x_2a s5_2c; do
  $B/conv_lab_stamps stamps $L 0 >> $O/stamps.txt 2>&1
done
cat $O/time_trunk_vec.txt $O/time_trunk_scalar.txt $O/time_head_vec.txt
