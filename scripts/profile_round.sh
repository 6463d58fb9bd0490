#!/bin/bash
# Round profile: rocprofv3 kernel-trace stats of the bench command + PMC passes (separate runs).
# usage (on the GPU box): bash scripts/profile_round.sh <tag>
TAG=${1:-round1}
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
python3 $R/bench.py --steps 50 --warmup 10 > $OUT/bench_default.json 2> $OUT/bench_default.err
python3 $R/bench.py --steps 50 --warmup 10 --streams 1 --no-cpu-baseline > $OUT/bench_streams1.json 2> $OUT/bench_streams1.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_streams1 -- python3 $R/bench.py --steps 20 --warmup 5 --streams 1 --no-cpu-baseline > $OUT/trace_streams1.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_default -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/trace_default.log 2>&1
i=0
for ctrs in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU" \
            "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32" \
            "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $ctrs --output-format csv -d $OUT/pmc$i -- python3 $R/bench.py --steps 2 --warmup 1 --streams 1 --no-cpu-baseline --no-graph > $OUT/pmc$i.log 2>&1
done
# the launch forms of the default (several images in flight) run, one launch at a time: HBM-side traffic only
for ctrs in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $ctrs --output-format csv -d $OUT/pmc$i -- python3 $R/bench.py --steps 2 --warmup 1 --streams 1 --no-cpu-baseline --no-graph --shared-tiles > $OUT/pmc$i.log 2>&1
done
python3 $R/scripts/profile_summary.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt | head -60
# keep only small files for the merge back
find $OUT -name "*kernel_trace.csv" -delete
find $OUT -name "*counter_collection.csv" -size +20M -delete
