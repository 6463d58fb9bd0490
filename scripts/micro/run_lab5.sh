#!/bin/bash
# lab run 5: grouped tile order (FRCNN_GROUP_M) on the head shapes and the trunk
cd "$(dirname "$0")/../.."
O=gpurun_out/lab5; mkdir -p $O
B=scripts/micro/_bin
for G in 0 2 4 8 12 16 32; do
  echo "== FRCNN_GROUP_M=$G" >> $O/time_head.txt
  FRCNN_GROUP_M=$G $B/conv_lab time head 23,26,21 >> $O/time_head.txt 2>&1
done
for G in 0 4 8 16; do
  echo "== FRCNN_GROUP_M=$G" >> $O/time_trunk.txt
  FRCNN_GROUP_M=$G $B/conv_lab time trunk 23 >> $O/time_trunk.txt 2>&1
done
cat $O/time_head.txt; grep -E "==|total" $O/time_trunk.txt
