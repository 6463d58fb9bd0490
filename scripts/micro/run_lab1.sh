#!/bin/bash
# round-2 lab run 1: baseline per-shape table + phase timestamps of representative trunk launches
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/lab1
B=scripts/micro/_bin
$B/conv_lab time trunk 0,22,122 > gpurun_out/lab1/time_trunk.txt 2>&1
$B/conv_lab time head 0,21,22 > gpurun_out/lab1/time_head.txt 2>&1
for L in s4_2c s4_2b s4x_2a s3_2c s3_2b s2_2c s2x_2a s2_2b s5_2c conv1; do
  $B/conv_lab_stamps stamps $L 0 >> gpurun_out/lab1/stamps.txt 2>&1
done
$B/conv_lab_stamps stamps s4_2c 122 >> gpurun_out/lab1/stamps.txt 2>&1
$B/conv_lab_stamps stamps s4_2b 122 >> gpurun_out/lab1/stamps.txt 2>&1
cat gpurun_out/lab1/time_trunk.txt
