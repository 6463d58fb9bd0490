GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
# kernel traces of the f32 training steps with the 128x128 weight-gradient kernel, grouped by (kernel, grid)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/run19; mkdir -p $O
export TMPDIR=/tmp; cd /tmp
export FRCNN_WGRAD_TARGET_BIG=64
for W in rpn det; do
  rocprofv3 --kernel-trace --output-format csv -d $O/tr_f32_$W -- python3 $R/scripts/bench_train.py --only $W --steps 20 --warmup 3 > $O/tr_f32_$W.log 2>&1
  python3 $R/scripts/trace_by_grid.py $O/tr_f32_$W 30 > $O/by_grid_f32_$W.txt 2>&1
  python3 $R/scripts/trace_gaps.py $O/tr_f32_$W > $O/gaps_f32_$W.txt 2>&1
done
find $O -name "*kernel_trace.csv" -delete
find $O -name "*.db" -delete
head -30 $O/by_grid_f32_det.txt; head -5 $O/gaps_f32_det.txt
