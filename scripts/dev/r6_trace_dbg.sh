#!/bin/bash
# round 6 dev: timeline of the replayed mixed RPN step (device inputs) under FRCNN_DBG variants
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out/r6_trace_dbg
mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
export FRCNN_TRAIN_GRAPH=1
for v in "$@"; do
  export FRCNN_DBG=$v
  rocprofv3 --kernel-trace --output-format csv -d $OUT/ts_$v -- python3 $R/scripts/dev/r6_host_costs.py bf16 > $OUT/ts_$v.log 2>&1
  python3 $R/scripts/dev/trace_timeline.py $OUT/ts_$v > $OUT/timeline_$v.txt 2>&1
  grep inputs $OUT/ts_$v.log
done
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -delete
