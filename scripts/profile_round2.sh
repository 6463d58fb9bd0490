#!/bin/bash
# Round-2 profile (GPU box): benches of every config, rocprofv3 kernel-trace stats of the headline command, PMC passes
# (separate runs, --pmc only), the per-shape conv table of the lab harness.   usage: bash scripts/profile_round2.sh [tag]
TAG=${1:-round2}
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
python3 $R/bench.py --config c1 --steps 50 --warmup 10 --streams 1 --no-cpu-baseline > $OUT/bench_c1_vgg16_rpn_streams1.json 2>> $OUT/bench_c1.err
python3 $R/bench.py --config c4 --steps 50 --warmup 10 > $OUT/bench_c4_default.json 2> $OUT/bench_c4.err
python3 $R/bench.py --config c4 --steps 50 --warmup 10 --streams 1 --no-cpu-baseline > $OUT/bench_c4_streams1.json 2>> $OUT/bench_c4.err
python3 $R/scripts/bench_train.py > $OUT/bench_train_f32.json 2> $OUT/bench_train.err
python3 $R/scripts/bench_train.py --bf16 > $OUT/bench_train_mixed_bf16.json 2>> $OUT/bench_train.err
# ---- per-shape conv table (C++ harness, 20 launches per hipGraph, best of 4)
$R/scripts/micro/_bin/conv_lab time all 0 > $OUT/conv_shapes_f32.txt 2>&1
$R/scripts/micro/_bin/conv_lab time trunk 0,22,122 > $OUT/conv_shapes_f32_trunk_variants.txt 2>&1
# ---- kernel traces of the headline command (the profiler starts the runtime before bench.py does: ask for the queues here)
export GPU_MAX_HW_QUEUES=8
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_default -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-io > $OUT/trace_default.log 2>&1
unset GPU_MAX_HW_QUEUES
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_streams1 -- python3 $R/bench.py --steps 20 --warmup 5 --streams 1 --no-cpu-baseline --no-io > $OUT/trace_streams1.log 2>&1
# ---- PMC: separate passes, eager single stream (latency-policy launch forms), then the launch forms of the default run
i=0
for ctrs in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU" \
            "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32" \
            "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $ctrs --output-format csv -d $OUT/pmc$i -- python3 $R/bench.py --steps 2 --warmup 1 --streams 1 --no-cpu-baseline --no-graph > $OUT/pmc$i.log 2>&1
done
for ctrs in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $ctrs --output-format csv -d $OUT/pmc$i -- python3 $R/bench.py --steps 2 --warmup 1 --streams 1 --no-cpu-baseline --no-graph --shared-tiles > $OUT/pmc$i.log 2>&1
done
# configs[3] (bf16 kernels): HBM-side traffic of its conv launches
for ctrs in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $ctrs --output-format csv -d $OUT/pmc$i -- python3 $R/bench.py --config c4 --steps 2 --warmup 1 --streams 1 --no-cpu-baseline --no-graph --shared-tiles > $OUT/pmc$i.log 2>&1
done
python3 $R/scripts/profile_summary.py $OUT > $OUT/summary.txt 2>&1
head -40 $OUT/summary.txt
for t in trace_default trace_streams1; do
  f=$(find $OUT/$t -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/${t}_kernel_stats.csv
done
# keep only small files for the merge back
find $OUT -name "*kernel_trace.csv" -delete
find $OUT -name "*.db" -delete
find $OUT -name "*counter_collection.csv" -size +20M -delete
du -sh $OUT
