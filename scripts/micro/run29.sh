GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
# configs[3] (ResNet-101 bf16): kernel traces grouped by (kernel, grid), one image in flight and four
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/run29; mkdir -p $O
export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $O/tr_s1 -- python3 $R/bench.py --config c4 --steps 20 --warmup 5 --streams 1 --no-cpu-baseline --no-io > $O/tr_s1.log 2>&1
python3 $R/scripts/trace_by_grid.py $O/tr_s1 40 > $O/by_grid_c4_streams1.txt 2>&1
python3 $R/scripts/trace_gaps.py $O/tr_s1 > $O/gaps_c4_streams1.txt 2>&1
rocprofv3 --kernel-trace --output-format csv -d $O/tr_s4 -- python3 $R/bench.py --config c4 --steps 20 --warmup 5 --no-cpu-baseline --no-io > $O/tr_s4.log 2>&1
python3 $R/scripts/trace_by_grid.py $O/tr_s4 30 > $O/by_grid_c4_default.txt 2>&1
find $O -name "*kernel_trace.csv" -delete
find $O -name "*.db" -delete
head -46 $O/by_grid_c4_streams1.txt; head -8 $O/gaps_c4_streams1.txt
