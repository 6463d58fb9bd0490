#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out/group_m_head
mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
for g in 0 1 2 4 8 115; do
  export FRCNN_GROUP_M=$g
  python3 $R/scripts/dev/r6_group_m_head.py 40
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/g$g -- python3 $R/scripts/dev/r6_group_m_head.py 2 > $OUT/g$g.log 2>&1
  f=$(ls -S $(find $OUT/g$g -name "*counter_collection.csv") | head -1)
  python3 - "$f" <<'PY'
import csv, sys
v = [float(r["Counter_Value"]) for r in csv.DictReader(open(sys.argv[1])) if r["Counter_Name"] == "FETCH_SIZE" and "h3_db" in r["Kernel_Name"]]
print("   FETCH_SIZE x 2 per launch: %.0f MB  (A 120 MB + residual 482 MB + weights 4 MB)" % (2 * sum(v) / len(v) * 1024 / 1e6))
PY
done
