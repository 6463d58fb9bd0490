#!/bin/bash
GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
# one gpurun call: phase stamps of the tiled bf16 kernel + the ring lab on the 1x1 layers of configs[3]  ->  profiles/round4_lab/
cd $GRAFT_REPO_ROOT
O=gpurun_out/ring_lab; mkdir -p $O
B=scripts/micro/_bin
{
for sh in "117600 512 2048 0 1" "117600 512 2048 0 0" "28576 256 1024 0 1" "28576 1024 256 0 0" "117600 512 2048 45 1"; do $B/bf16_stamps $sh; done
} > $O/bf16_stamps.txt 2>&1
{
for v in v1_32_4 v1_64_4 v2_2 v2_4 v2_2nt v3_2_2 v3_4_2 v4_2; do
  $B/bf16_ring_lab 117600 512 2048 1 $v
  $B/bf16_ring_lab 28576 256 1024 1 $v
  $B/bf16_ring_lab 117600 2048 512 0 $v
done
} > $O/bf16_ring_lab.txt 2>&1
python3 scripts/bf16_one.py 2400 7 7 512 2048 1 1 valid 0 20 1 1 > $O/bf16_one.txt 2>&1
python3 scripts/bf16_one.py 2400 7 7 512 2048 1 1 valid 0 20 1 0 >> $O/bf16_one.txt 2>&1
tail -3 $O/bf16_ring_lab.txt
