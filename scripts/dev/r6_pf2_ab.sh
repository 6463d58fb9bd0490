#!/bin/bash
# A/B on one box: the f16x3 256x128 loop with a second register stage (-DH3_PF2, scripts/dev/r6_a_once_lab.sh build) against the product.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
B=$R/scripts/micro/_bin
cd $R
FRCNN_LIB_PATH=$B/libfrcnn_hip_pf2.so python3 -m pytest tests/test_conv_h3_gpu.py -q -x 2>&1 | tail -2
for rep in 1 2; do
  for lib in product pf2; do
    if [ $lib != product ]; then export FRCNN_LIB_PATH=$B/libfrcnn_hip_$lib.so; else unset FRCNN_LIB_PATH; fi
    python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-io --conv-table --no-extra > /tmp/ab.json 2> /tmp/ab.err
    python3 -c "import json; d=json.load(open('/tmp/ab.json')); print('$lib', d['value'], d['roofline']['frac'], d['roofline']['backbone_conv']['in_flight']['ms_per_image'])"
    grep "^conv" /tmp/ab.err | head -4 | cut -c1-150
  done
done
