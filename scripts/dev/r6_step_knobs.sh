#!/bin/bash
# round 6 dev: the replayed mixed / f32 RPN step under stream priorities and weight-gradient flush sizes (device inputs)
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
run() { python scripts/dev/r6_host_costs.py $1 2>&1 | grep "device inputs" | sed "s/^/$2 /"; }
for dt in bf16 f32; do
  run $dt "base"
  FRCNN_TRAIN_MAIN_PRIO=-1 run $dt "main-high"
  FRCNN_TRAIN_MAIN_PRIO=-1 FRCNN_TRAIN_WGRAD_PRIO=-1 run $dt "main+wgrad-high"
  FRCNN_TRAIN_WGRAD_PRIO=-1 run $dt "wgrad-high"
  for f in 4 6 12 16; do FRCNN_WGRAD_FLUSH=$f run $dt "flush$f"; done
  run $dt "base-again"
done
