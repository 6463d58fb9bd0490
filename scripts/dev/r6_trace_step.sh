#!/bin/bash
# round 6: kernel trace of one training step, graphs on / off    usage: r6_trace_step.sh [extra bench_train args]
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out/r6_trace_step
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for g in 1 0; do
  export FRCNN_TRAIN_GRAPH=$g
  rocprofv3 --kernel-trace --output-format csv -d $OUT/ts_g$g -- python3 $R/scripts/bench_train.py --bf16 --only rpn --steps 30 --warmup 10 "$@" > $OUT/ts_g$g.log 2>&1
  python3 $R/scripts/trace_step.py $OUT/ts_g$g 60 > $OUT/step_rpn_mixed_graph$g.txt 2>&1
  python3 $R/scripts/dev/trace_timeline.py $OUT/ts_g$g > $OUT/timeline_graph$g.txt 2>&1
  head -3 $OUT/step_rpn_mixed_graph$g.txt
done
find $OUT -name "*kernel_trace.csv" -delete
find $OUT -name "*.db" -delete
