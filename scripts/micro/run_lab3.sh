#!/bin/bash
# lab run 3: mid-chunk-barrier main loop (codes 23 / x23) against the late-store loop (22 / x22); regrouped vector epilogue on the big tiles
cd "$(dirname "$0")/../.."
O=gpurun_out/lab3; mkdir -p $O
B=scripts/micro/_bin
$B/conv_lab time trunk 0,23,122,123,223,323 > $O/time_trunk.txt 2>&1
$B/conv_lab time head 0,21,22,23 > $O/time_head.txt 2>&1
for T in 122 123; do $B/conv_lab_stamps stamps s4_2b $T >> $O/stamps.txt 2>&1; done
for T in 22 23; do $B/conv_lab_stamps stamps s4_2c $T >> $O/stamps.txt 2>&1; $B/conv_lab_stamps stamps s5_2c $T >> $O/stamps.txt 2>&1; done
cat $O/time_trunk.txt $O/time_head.txt
