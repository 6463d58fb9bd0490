#!/bin/bash
# lab run 4: variant 2 on the big tile (26) and the 64x128 / 128x64 tiles (24 / 25)
cd "$(dirname "$0")/../.."
O=gpurun_out/lab4; mkdir -p $O
B=scripts/micro/_bin
$B/conv_lab time head 21,26,23,24,25 > $O/time_head.txt 2>&1
$B/conv_lab time trunk 23,24,25,26 > $O/time_trunk.txt 2>&1
cat $O/time_head.txt $O/time_trunk.txt
