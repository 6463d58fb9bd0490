#!/bin/bash
# plane tensors inside the TRUNK's bottleneck blocks (FRCNN_TRUNK_PLANES=1) again, under round 6's shared-chip tile policy and with the ring
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
for rep in 1 2; do
  for v in 0 1; do
    FRCNN_TRUNK_PLANES=$v python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-io --no-extra 2>/dev/null | python3 -c "import json,sys; d=json.load(sys.stdin); print('trunk planes=$v:', d['value'], d['roofline']['backbone_conv']['in_flight']['ms_per_image'])"
  done
done
