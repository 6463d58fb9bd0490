#!/bin/bash
# Copy what scripts/profile_round5.sh left under gpurun_out/<tag>/ (scratch) into profiles/ (tracked): bench lines, conv tables, the
# condensed summary, per-kernel trace stats, the training steps by (kernel, grid), and traffic.json (bench.py reads it).
#   usage: bash scripts/collect_profiles.sh [tag]
TAG=${1:-round6}
R=$(cd "$(dirname "$0")/.." && pwd)
S=$R/gpurun_out/$TAG
P=$R/profiles
for f in $S/bench_*.json $S/bench_*_conv_table.txt $S/*_trace_by_grid.txt $S/trace_*_kernel_stats.csv $S/train_step_*.txt $S/pmc_*.txt; do
  [ -s "$f" ] && cp "$f" $P/${TAG}_$(basename $f)
done
[ -s $S/summary.txt ] && cut -c1-400 $S/summary.txt > $P/${TAG}_summary.txt
[ -s $S/traffic.json ] && cp $S/traffic.json $P/traffic.json
ls $P | grep "^${TAG}_" | wc -l
