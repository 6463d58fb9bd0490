#!/bin/bash
# Why is the ring slower on the sixteen-chunk 512 -> 2048 launches?  Run: the double buffer against the ring on every plane-input launch.
# (A lab build of this script's first version also ran the double buffer with the ring's LDS request -- 128 / 144 KB instead of 96:
#  514-537 us against 516-520, so the larger request is not it; nor is the place of the scale reads, in front of or behind the first
#  two chunks' requests: 552 / 552 against 545 / 552.)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
run() { python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-io --conv-table --no-extra > /tmp/ab.json 2> /tmp/ab.err; grep "^conv" /tmp/ab.err | head -2 | cut -c1-150; }
for rep in 1 2; do
  echo "== double buffer"; FRCNN_H3_RING=0 run
  echo "== ring for every plane-input launch"; FRCNN_H3_RING_MIN_CHUNKS=1 run
  echo "== default (ring from 32 chunks)"; run
done
