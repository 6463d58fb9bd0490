#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the eager four-image pass, per (kernel, grid): which SHAPE of the dominant kernel carries the extra traffic.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out/pmc_by_grid
mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $OUT/$c -- python3 $R/bench.py --steps 2 --warmup 1 --streams 1 --batch 4 --no-cpu-baseline --no-graph --shared-tiles --no-extra > $OUT/$c.log 2>&1
  f=$(ls -S $(find $OUT/$c -name "*counter_collection.csv") | head -1)
  python3 - "$f" "$c" > $OUT/$c.by_grid.txt <<'PY'
import csv, sys, collections
f, c = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(f)):
    if r["Counter_Name"] != c: continue
    k = (r["Kernel_Name"][:60], r["Grid_Size"], r.get("Workgroup_Size", ""))
    acc[k][0] += 1; acc[k][1] += float(r["Counter_Value"])
for k, (n, v) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:40]:
    print("%-60s grid %-10s wg %-5s n %4d  avg %10.1f KB(raw units)" % (k[0], k[1], k[2], n, v / n))
PY
  find $OUT/$c -name "*.csv" -size +5M -delete; find $OUT/$c -name "*.db" -delete
done
cat $OUT/FETCH_SIZE.by_grid.txt | head -20; cat $OUT/WRITE_SIZE.by_grid.txt | head -12
