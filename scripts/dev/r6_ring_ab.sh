#!/bin/bash
# A/B on one box: plane-input launches of the f16x3 engine on the three-stage direct-to-LDS ring (default) against the register-staged
# double buffer (FRCNN_H3_RING=0).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
python3 -m pytest tests/test_conv_h3_gpu.py tests/test_h3_fences_gpu.py -q -x 2>&1 | tail -2
for rep in 1 2; do
  for ring in 1 0; do
    export FRCNN_H3_RING=$ring
    python3 scripts/dev/r6_head_3x3.py ring=$ring
    python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-io --conv-table --no-extra > /tmp/ab.json 2> /tmp/ab.err
    python3 -c "import json; d=json.load(open('/tmp/ab.json')); print('ring=$ring', d['value'], d['roofline']['frac'], d['roofline']['backbone_conv']['in_flight']['ms_per_image'])"
    grep "^conv" /tmp/ab.err | head -3 | cut -c1-150
  done
done
