#!/bin/bash
# Round-3 profile (GPU box): bench lines of every config, rocprofv3 kernel-trace stats of the headline command and of the
# configs[3] command, PMC passes (separate runs, --pmc only).   usage: bash scripts/profile_round3.sh [tag]
TAG=${1:-round3}
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
# ---- bench lines (the driver's command first)
python3 $R/bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
python3 $R/bench.py --steps 50 --warmup 10 --streams 1 > $OUT/bench_streams1.json 2> $OUT/bench_streams1.err
python3 $R/bench.py --config c1 --steps 50 --warmup 10 > $OUT/bench_c1_vgg16_rpn.json 2> $OUT/bench_c1.err
python3 $R/bench.py --config c4 --steps 50 --warmup 10 --conv-table > $OUT/bench_c4_default.json 2> $OUT/bench_c4.err
grep "^conv" $OUT/bench_c4.err > $OUT/bench_c4_conv_table.txt
python3 $R/bench.py --config c4 --steps 50 --warmup 10 --batch 1 --streams 4 --no-cpu-baseline > $OUT/bench_c4_batch1_streams4.json 2>> $OUT/bench_c4.err
python3 $R/bench.py --config c4 --steps 50 --warmup 10 --batch 1 --streams 1 --no-cpu-baseline > $OUT/bench_c4_batch1_streams1.json 2>> $OUT/bench_c4.err
python3 $R/scripts/bench_train.py > $OUT/bench_train_f32.json 2> $OUT/bench_train.err
python3 $R/scripts/bench_train.py --bf16 > $OUT/bench_train_mixed_bf16.json 2>> $OUT/bench_train.err
# ---- kernel traces (the profiler starts the runtime before bench.py does: ask for the queues here)
export GPU_MAX_HW_QUEUES=8
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_default -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-io > $OUT/trace_default.log 2>&1
unset GPU_MAX_HW_QUEUES
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_streams1 -- python3 $R/bench.py --steps 20 --warmup 5 --streams 1 --no-cpu-baseline --no-io > $OUT/trace_streams1.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_c4 -- python3 $R/bench.py --config c4 --steps 20 --warmup 5 --no-cpu-baseline --no-io > $OUT/trace_c4.log 2>&1
# ---- PMC: separate passes, eager single stream with the launch forms of the default (multi-image) run
i=0
for ctrs in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $ctrs --output-format csv -d $OUT/pmc$i -- python3 $R/bench.py --steps 2 --warmup 1 --streams 1 --no-cpu-baseline --no-graph --shared-tiles > $OUT/pmc$i.log 2>&1
done
# configs[3]: the batched pass, eager (one batch of eight per step)
for ctrs in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $ctrs --output-format csv -d $OUT/pmc$i -- python3 $R/scripts/c4_batch_eager.py 3 > $OUT/pmc$i.log 2>&1
done
python3 $R/scripts/profile_summary.py $OUT > $OUT/summary.txt 2>&1
head -60 $OUT/summary.txt
for t in trace_default trace_streams1 trace_c4; do
  f=$(find $OUT/$t -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/${t}_kernel_stats.csv
done
find $OUT -name "*kernel_trace.csv" -delete
find $OUT -name "*.db" -delete
find $OUT -name "*counter_collection.csv" -size +20M -delete
du -sh $OUT
