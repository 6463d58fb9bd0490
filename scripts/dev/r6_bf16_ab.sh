#!/bin/bash
# round 6: A/B on one box -- the bf16 engine on 16x16x32 (the library) against the 32x32x16 build (scripts/micro/_bin/libfrcnn_hip_oldbf16.so)
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
for rep in 1 2; do
for lib in new old; do
  if [ $lib = old ]; then export FRCNN_LIB_PATH=$R/scripts/micro/_bin/libfrcnn_hip_oldbf16.so; else unset FRCNN_LIB_PATH; fi
  python bench.py --config c4 --no-cpu-baseline --no-io --steps 20 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('$lib c4', d['value'], d['roofline']['frac'])"
  python scripts/bench_train.py --bf16 --steps 60 --warmup 40 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('$lib mixed steps', d['rpn_step1']['ms_per_step'], d['det_step2']['ms_per_step'])"
done
done
