#!/bin/bash
# configs[0] (VGG16 RPN forward, 12 one-image passes in flight) under round 6's shared-chip tile policy against round 5's
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
for rep in 1 2; do
  python3 bench.py --config c1 --steps 30 --warmup 5 --no-cpu-baseline --no-io --no-extra 2>/dev/null | python3 -c "import json,sys; d=json.load(sys.stdin); print('new policy:', d['value'], d['roofline']['frac'])"
  FRCNN_H3_SHARED_SMALL=0 FRCNN_H3_BIG_MIN_TILES_SHARED=256 python3 bench.py --config c1 --steps 30 --warmup 5 --no-cpu-baseline --no-io --no-extra 2>/dev/null | python3 -c "import json,sys; d=json.load(sys.stdin); print('round-5 policy:', d['value'], d['roofline']['frac'])"
  FRCNN_H3_SHARED_SMALL=81 FRCNN_H3_BIG_MIN_TILES_SHARED=256 python3 bench.py --config c1 --steps 30 --warmup 5 --no-cpu-baseline --no-io --no-extra 2>/dev/null | python3 -c "import json,sys; d=json.load(sys.stdin); print('81 below 256 tiles:', d['value'], d['roofline']['frac'])"
done
