#!/bin/bash
# round 6: the captured training step -- parity tests, then step times with the graphs on / off (device inputs, host inputs)
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out/r6_step_graph
mkdir -p $OUT
cd $R
python -m pytest tests/test_train_graph_gpu.py -x -q 2>&1 | tail -15 > $OUT/tests.txt
cat $OUT/tests.txt
for dt in bf16 f32; do
  for g in 1 0; do
    FRCNN_TRAIN_GRAPH=$g python scripts/dev/r6_host_costs.py $dt 2>&1 | grep inputs | sed "s/^/graph=$g /"
  done
done
