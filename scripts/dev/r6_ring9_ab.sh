#!/bin/bash
# A/B on one box: the head's 3x3 on the tap-reuse ring (default) against the plain ring (FRCNN_H3_RING9=0) and the double buffer (FRCNN_H3_RING=0).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
python3 -m pytest tests/test_conv_h3_gpu.py tests/test_h3_fences_gpu.py tests/test_canvas_gpu.py -q -x 2>&1 | tail -2
for rep in 1 2; do
  for v in "ring9:" "ring:FRCNN_H3_RING9=0" "db:FRCNN_H3_RING=0"; do
    env ${v##*:} python3 scripts/dev/r6_head_3x3.py ${v%%:*}
    env ${v##*:} python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-io --conv-table --no-extra > /tmp/ab.json 2> /tmp/ab.err
    python3 -c "import json; d=json.load(open('/tmp/ab.json')); print('${v%%:*}', d['value'], d['roofline']['frac'], d['roofline']['backbone_conv']['in_flight']['ms_per_image'])"
    grep "^conv" /tmp/ab.err | head -2 | cut -c1-150
  done
done
