#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out/graph_event_probe
mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
for v in graph graph_tick graph_bgraph eager; do
  rocprofv3 --kernel-trace --output-format csv -d $OUT/$v -- python3 $R/scripts/micro/graph_event_probe.py $v > $OUT/$v.log 2>&1
  python3 $R/scripts/micro/graph_event_probe_tl.py $OUT/$v > $OUT/tl_$v.txt 2>&1
done
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -delete
