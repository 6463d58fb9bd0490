#!/bin/bash
# Round-6 profile (GPU box): bench lines of every config (fp32 default = f16x3 engine; the exact bf16x6 split and native beside it), rocprofv3
# kernel-trace stats of the headline command and of the training steps, PMC passes (separate runs, --pmc only).
#   usage: bash scripts/profile_round6.sh [tag]
TAG=${1:-round6}
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
# ---- bench lines (the driver's command first)
python3 $R/bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
python3 $R/bench.py --f32-engine native --steps 50 --warmup 10 > $OUT/bench_native_f32_mfma.json 2> $OUT/bench_native.err
python3 $R/bench.py --f32-engine bf16x6 --steps 50 --warmup 10 --no-cpu-baseline > $OUT/bench_bf16x6_exact_split.json 2> $OUT/bench_bf16x6.err
python3 $R/bench.py --steps 50 --warmup 10 --streams 1 > $OUT/bench_streams1.json 2> $OUT/bench_streams1.err
python3 $R/bench.py --steps 50 --warmup 10 --batch 1 --streams 12 --no-cpu-baseline > $OUT/bench_batch1_streams12.json 2> $OUT/bench_batch1_streams12.err
python3 $R/bench.py --config c1 --steps 50 --warmup 10 > $OUT/bench_c1_vgg16_rpn.json 2> $OUT/bench_c1.err
python3 $R/bench.py --config c4 --steps 50 --warmup 10 --conv-table > $OUT/bench_c4_default.json 2> $OUT/bench_c4.err
grep "^conv" $OUT/bench_c4.err > $OUT/bench_c4_conv_table.txt
python3 $R/bench.py --dtype bf16 --steps 50 --warmup 10 > $OUT/bench_c2_shapes_on_bf16.json 2> $OUT/bench_c2bf16.err
python3 $R/scripts/bench_train.py --through-loop > $OUT/bench_train_f32.json 2> $OUT/bench_train.err
python3 $R/scripts/bench_train.py --bf16 --through-loop > $OUT/bench_train_mixed_bf16.json 2>> $OUT/bench_train.err
# the same steps launched eagerly (round 5's form), for the difference the replayed step makes through train_util's loops
FRCNN_TRAIN_GRAPH=0 python3 $R/scripts/bench_train.py --through-loop --no-host-feed > $OUT/bench_train_f32_eager_launch.json 2>> $OUT/bench_train.err
FRCNN_TRAIN_GRAPH=0 python3 $R/scripts/bench_train.py --bf16 --through-loop --no-host-feed > $OUT/bench_train_mixed_bf16_eager_launch.json 2>> $OUT/bench_train.err
# where the pieces of a replayed step run, WITHOUT a profiler (timing events; rocprofv3 slows hipGraphLaunch enough to change the picture)
python3 $R/scripts/dev/r6_event_timeline.py bf16 2>&1 | grep -v amdgpu > $OUT/train_step_rpn_mixed_bf16_event_timeline.txt
python3 $R/scripts/dev/r6_event_timeline.py f32 2>&1 | grep -v amdgpu > $OUT/train_step_rpn_f32_event_timeline.txt
python3 $R/bench.py --no-extra --steps 20 --warmup 5 --no-cpu-baseline --no-io --conv-table > /dev/null 2> $OUT/bench_default_conv_table.err
grep "^conv" $OUT/bench_default_conv_table.err > $OUT/bench_default_conv_table.txt
# ---- kernel traces (fp32 default: four passes of four images on the runtime's four queues)
export FRCNN_BENCH_NO_ENTRY=1
export FRCNN_BENCH_NO_NATIVE=1       # the traces hold the timed launch forms only
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_default -- python3 $R/bench.py --no-extra --steps 20 --warmup 5 --no-cpu-baseline --no-io > $OUT/trace_default.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_streams1 -- python3 $R/bench.py --no-extra --steps 20 --warmup 5 --streams 1 --no-cpu-baseline --no-io > $OUT/trace_streams1.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_c4 -- python3 $R/bench.py --no-extra --config c4 --steps 20 --warmup 5 --no-cpu-baseline --no-io > $OUT/trace_c4.log 2>&1
unset FRCNN_BENCH_NO_ENTRY FRCNN_BENCH_NO_NATIVE
# training steps, grouped by (kernel, grid): what a step is made of
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_train_f32 -- python3 $R/scripts/bench_train.py --steps 10 --warmup 5 > $OUT/trace_train_f32.log 2>&1
python3 $R/scripts/trace_by_grid.py $OUT/trace_train_f32 60 > $OUT/train_f32_trace_by_grid.txt 2>&1
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_train_mixed -- python3 $R/scripts/bench_train.py --bf16 --steps 10 --warmup 5 > $OUT/trace_train_mixed.log 2>&1
python3 $R/scripts/trace_by_grid.py $OUT/trace_train_mixed 60 > $OUT/train_mixed_bf16_trace_by_grid.txt 2>&1
# one replayed step cut out of the trace, per stream (kernels of a step, union of intervals; host-bound under the profiler, see above)
for m in rpn det; do
  rocprofv3 --kernel-trace --output-format csv -d $OUT/ts_mixed_$m -- python3 $R/scripts/bench_train.py --bf16 --only $m --steps 30 --warmup 10 > $OUT/ts.log 2>&1
  python3 $R/scripts/trace_step.py $OUT/ts_mixed_$m 40 > $OUT/train_step_${m}_mixed_bf16.txt 2>&1
  rocprofv3 --kernel-trace --output-format csv -d $OUT/ts_f32_$m -- python3 $R/scripts/bench_train.py --only $m --steps 30 --warmup 10 > $OUT/ts.log 2>&1
  python3 $R/scripts/trace_step.py $OUT/ts_f32_$m 40 > $OUT/train_step_${m}_f32.txt 2>&1
done
# ---- PMC: separate passes, eager single stream with the launch forms of the default run (four images per pass)
i=0
for ctrs in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum" \
            "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $ctrs --output-format csv -d $OUT/pmc$i -- python3 $R/bench.py --no-extra --steps 2 --warmup 1 --streams 1 --batch 4 --no-cpu-baseline --no-graph --shared-tiles > $OUT/pmc$i.log 2>&1
done
# configs[3]: the batched pass, eager (one batch of eight per step)
for ctrs in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $ctrs --output-format csv -d $OUT/pmc$i -- python3 $R/scripts/c4_batch_eager.py 3 > $OUT/pmc$i.log 2>&1
done
# FETCH_SIZE / WRITE_SIZE per (kernel, grid) -- which SHAPE of a kernel carries the traffic -- and the counter's calibration on known byte counts
bash $R/scripts/dev/r6_pmc_by_grid.sh > /dev/null 2>&1
cp $R/gpurun_out/pmc_by_grid/FETCH_SIZE.by_grid.txt $OUT/pmc_FETCH_SIZE_by_grid.txt 2>/dev/null
cp $R/gpurun_out/pmc_by_grid/WRITE_SIZE.by_grid.txt $OUT/pmc_WRITE_SIZE_by_grid.txt 2>/dev/null
bash $R/scripts/micro/fetch_size_calibration.sh > $OUT/pmc_fetch_size_calibration.txt 2>&1
cd /tmp
python3 $R/scripts/profile_summary.py $OUT > $OUT/summary.txt 2>&1
head -40 $OUT/summary.txt | cut -c1-400
for t in trace_default trace_streams1 trace_c4; do
  f=$(ls -S $(find $OUT/$t -name "*kernel_stats.csv") 2>/dev/null | head -1); [ -n "$f" ] && cp $f $OUT/${t}_kernel_stats.csv
done
find $OUT -name "*kernel_trace.csv" -delete
find $OUT -name "*.db" -delete
find $OUT -name "*counter_collection.csv" -size +20M -delete
du -sh $OUT
