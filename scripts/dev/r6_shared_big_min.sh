#!/bin/bash
# Shared-chip policy: from how many 128x128 tiles on does a launch take the 256x128 form (below: 128x128 on eight waves)?
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
for rep in 1 2; do
  for v in 256 128 512 1024; do
    FRCNN_H3_BIG_MIN_TILES_SHARED=$v python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-io --no-extra 2>/dev/null | python3 -c "import json,sys; d=json.load(sys.stdin); r=d['roofline']; print('256x128 from $v tiles', d['value'], r['backbone_conv']['in_flight']['ms_per_image'])"
  done
done
