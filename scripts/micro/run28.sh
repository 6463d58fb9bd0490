GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
# kernel traces of the mixed-precision training steps (current code), grouped by (kernel, grid) + idle gaps
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/run28; mkdir -p $O
export TMPDIR=/tmp; cd /tmp
for M in bf16 f32; do
  FLAG=""; [ $M = bf16 ] && FLAG="--bf16"
  for W in rpn det; do
    rocprofv3 --kernel-trace --output-format csv -d $O/tr_${M}_$W -- python3 $R/scripts/bench_train.py $FLAG --only $W --steps 20 --warmup 3 --sync-each-step > $O/tr_${M}_$W.log 2>&1
    python3 $R/scripts/trace_by_grid.py $O/tr_${M}_$W 40 > $O/by_grid_${M}_$W.txt 2>&1
    python3 $R/scripts/trace_gaps.py $O/tr_${M}_$W > $O/gaps_${M}_$W.txt 2>&1
  done
done
find $O -name "*kernel_trace.csv" -delete
find $O -name "*.db" -delete
head -45 $O/by_grid_bf16_rpn.txt; head -6 $O/gaps_bf16_rpn.txt
