#!/bin/bash
# round 6: timeline of one mixed RPN step fed with DEVICE inputs, graphs on / off
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out/r6_trace_dev
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
DT=${1:-bf16}
for g in 1 0; do
  export FRCNN_TRAIN_GRAPH=$g
  python3 $R/scripts/dev/r6_host_costs.py $DT 2>&1 | grep -v amdgpu.ids > $OUT/host_g${g}_$DT.txt
  rocprofv3 --kernel-trace --output-format csv -d $OUT/ts_g$g -- python3 $R/scripts/dev/r6_host_costs.py $DT > $OUT/ts_g$g.log 2>&1
  python3 $R/scripts/dev/trace_timeline.py $OUT/ts_g$g > $OUT/timeline_${DT}_graph$g.txt 2>&1
  grep -E "inputs" $OUT/host_g${g}_$DT.txt
done
find $OUT -name "*kernel_trace.csv" -delete
find $OUT -name "*.db" -delete
