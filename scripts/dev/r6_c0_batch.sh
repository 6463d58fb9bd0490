#!/bin/bash
# configs[0]: images per captured pass x passes in flight (round 5: one image per pass, twelve passes on twelve queues)
#   usage: r6_c0_batch.sh [BxS ...]   e.g. 8x2 4x4
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
[ $# -eq 0 ] && set -- 1x12 4x4 2x6 4x3 8x2 1x12
for bs in "$@"; do
  b=${bs%x*}; s=${bs#*x}
  python3 bench.py --config c1 --batch $b --streams $s --steps 20 --warmup 4 --no-cpu-baseline --no-io --no-extra 2>/tmp/err.txt | python3 -c "import json,sys; d=json.load(sys.stdin); print('images per pass $b x passes in flight $s:', d['value'], d['roofline']['frac'] if d.get('roofline') else None, d['config'].get('hw_queues'))" || tail -3 /tmp/err.txt
done
