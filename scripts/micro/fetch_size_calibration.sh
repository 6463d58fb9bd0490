#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out/fetch_calib
mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $OUT/$c -- python3 $R/scripts/micro/fetch_size_calibration.py > $OUT/$c.log 2>&1
  f=$(ls -S $(find $OUT/$c -name "*counter_collection.csv") | head -1)
  python3 - "$f" "$c" > $OUT/$c.txt <<'PY'
import csv, sys
f, c = sys.argv[1], sys.argv[2]
for r in csv.DictReader(open(f)):
    if r["Counter_Name"] != c: continue
    v = float(r["Counter_Value"])
    if v > 20000: print("%-70s grid %-10s %12.1f KB" % (r["Kernel_Name"][:70], r["Grid_Size"], v))
PY
done
tail -5 $OUT/FETCH_SIZE.log; cat $OUT/FETCH_SIZE.txt; cat $OUT/WRITE_SIZE.txt
