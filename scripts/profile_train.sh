#!/bin/bash
# The training part of the round profile on its own: bench lines of the four steps, kernel traces grouped by (kernel, grid), one step cut
# out per stream.   usage: bash scripts/profile_train.sh [tag]
TAG=${1:-round4}
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
python3 $R/scripts/bench_train.py > $OUT/bench_train_f32.json 2> $OUT/bench_train.err
python3 $R/scripts/bench_train.py --bf16 > $OUT/bench_train_mixed_bf16.json 2>> $OUT/bench_train.err
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_train_f32 -- python3 $R/scripts/bench_train.py --steps 10 --warmup 5 > $OUT/trace_train_f32.log 2>&1
python3 $R/scripts/trace_by_grid.py $OUT/trace_train_f32 60 > $OUT/train_f32_trace_by_grid.txt 2>&1
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_train_mixed -- python3 $R/scripts/bench_train.py --bf16 --steps 10 --warmup 5 > $OUT/trace_train_mixed.log 2>&1
python3 $R/scripts/trace_by_grid.py $OUT/trace_train_mixed 60 > $OUT/train_mixed_bf16_trace_by_grid.txt 2>&1
for m in rpn det; do
  rocprofv3 --kernel-trace --output-format csv -d $OUT/ts_f32_$m -- python3 $R/scripts/bench_train.py --only $m --steps 30 --warmup 10 > $OUT/ts.log 2>&1
  python3 $R/scripts/trace_step.py $OUT/ts_f32_$m 40 > $OUT/train_step_${m}_f32.txt 2>&1
  rocprofv3 --kernel-trace --output-format csv -d $OUT/ts_mixed_$m -- python3 $R/scripts/bench_train.py --bf16 --only $m --steps 30 --warmup 10 > $OUT/ts.log 2>&1
  python3 $R/scripts/trace_step.py $OUT/ts_mixed_$m 40 > $OUT/train_step_${m}_mixed_bf16.txt 2>&1
done
find $OUT -name "*kernel_trace.csv" -delete
find $OUT -name "*.db" -delete
cut -c1-400 $OUT/bench_train_f32.json; cut -c1-400 $OUT/bench_train_mixed_bf16.json
