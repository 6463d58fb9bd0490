#!/bin/bash
# round 6 dev: per-kernel durations of a mixed RPN step, new bf16 engine against the 32x32x16 build
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out/r6_bf16_ab
mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
for lib in new old; do
  if [ $lib = old ]; then export FRCNN_LIB_PATH=$R/scripts/micro/_bin/libfrcnn_hip_oldbf16.so; else unset FRCNN_LIB_PATH; fi
  rocprofv3 --kernel-trace --output-format csv -d $OUT/ts_$lib -- python3 $R/scripts/bench_train.py --bf16 --only rpn --steps 30 --warmup 10 > $OUT/ts_$lib.log 2>&1
  python3 $R/scripts/trace_step.py $OUT/ts_$lib 40 > $OUT/step_$lib.txt 2>&1
done
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -delete
