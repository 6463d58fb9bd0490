#!/bin/bash
# from how many chunks on a plane-input reduction walks the ring (FRCNN_H3_RING_MIN_CHUNKS), now that trunk and VGG layers read planes too
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
for rep in 1 2; do
  for v in 32 16 24 48 80; do
    FRCNN_H3_RING_MIN_CHUNKS=$v python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-io --no-extra 2>/dev/null | python3 -c "import json,sys; d=json.load(sys.stdin); print('ring from $v chunks: c2', d['value'], d['roofline']['backbone_conv']['in_flight']['ms_per_image'])"
  done
done
for v in 32 16 48; do
  FRCNN_H3_RING_MIN_CHUNKS=$v python3 bench.py --config c1 --steps 20 --warmup 4 --no-cpu-baseline --no-io --no-extra 2>/dev/null | python3 -c "import json,sys; d=json.load(sys.stdin); print('ring from $v chunks: c0', d['value'])"
done
