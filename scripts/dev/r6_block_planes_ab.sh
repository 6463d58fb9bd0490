#!/bin/bash
# A/B on one box: head block outputs as planes (default) against f32 block outputs (FRCNN_HEAD_BLOCK_PLANES=0).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
for rep in 1 2; do
  for v in 1 0; do
    FRCNN_HEAD_BLOCK_PLANES=$v python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-io --conv-table --no-extra > /tmp/ab.json 2> /tmp/ab.err
    python3 -c "import json; d=json.load(open('/tmp/ab.json')); r=d['roofline']; print('block planes=$v', d['value'], r['frac'], r['avg_launch_us'], r['launches_per_image'], d.get('parity',{}).get('e2e'))"
    grep "^conv" /tmp/ab.err | head -4 | cut -c1-150
  done
done
