#!/bin/bash
# round 6: bare step and loop iteration (train_util.train_rpn / train_detector_step2), graphs on / off, f32 and mixed bf16
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out/r6_through_loop
mkdir -p $OUT
cd $R
for g in 1 0; do
  for dt in "--bf16" ""; do
    FRCNN_TRAIN_GRAPH=$g python scripts/bench_train.py $dt --steps 60 --warmup 40 --through-loop > $OUT/bench_graph${g}${dt}.json 2> $OUT/bench_graph${g}${dt}.err
    python - <<PY
import json
d=json.load(open("$OUT/bench_graph${g}${dt}.json"))
for k in ("rpn_step1","det_step2"):
    tl=d[k].get("through_loop",{})
    print("graph=$g $dt", k, "bare", d[k]["ms_per_step"], "sync", d[k].get("ms_per_step_losses_read_every_step"), "loop fast", tl.get("fast_feed",{}).get("ms_per_iteration"), "loop host", tl.get("host_feed",{}).get("ms_per_iteration"))
PY
  done
done
