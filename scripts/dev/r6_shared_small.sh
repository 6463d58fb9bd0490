#!/bin/bash
# In flight the chip is saturated (16 images per 29 ms against 1.9 ms of isolated conv time per image): does a launch too small for the
# 256x128 form do its FLOPs cheaper on 128x128 tiles (codes 81 / 83) than on 64x64 (84), now that idle CUs are other passes' to fill?
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
for rep in 1 2; do
  for v in 0 81 83; do
    FRCNN_H3_SHARED_SMALL=$v python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-io --no-extra 2>/dev/null | python3 -c "import json,sys; d=json.load(sys.stdin); r=d['roofline']; print('shared small=$v', d['value'], r['backbone_conv']['in_flight']['ms_per_image'])"
  done
done
