#!/bin/bash
# configs[0]: plane tensors between the convolutions of a VGG block (FRCNN_VGG_PLANES, default on) against f32 tensors
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
for rep in 1 2; do
  for v in 1 0; do
    FRCNN_VGG_PLANES=$v python3 bench.py --config c1 --steps 30 --warmup 5 --no-cpu-baseline --no-io --no-extra --conv-table > /tmp/ab.json 2> /tmp/ab.err
    python3 -c "import json; d=json.load(open('/tmp/ab.json')); print('VGG planes=$v:', d['value'], d['roofline']['frac'], d.get('parity',{}).get('ok'))"
    grep "^conv" /tmp/ab.err | head -5 | cut -c1-140
  done
done
