#!/bin/bash
# Hardware queues (GPU_MAX_HW_QUEUES; the runtime's default is 4) x passes in flight: the headline loop and the reference's entry point
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
for q in 4 8 16; do
  for s in 4 8; do
    GPU_MAX_HW_QUEUES=$q python3 bench.py --batch 4 --streams $s --steps 20 --warmup 4 --no-cpu-baseline --no-io --no-extra 2>/dev/null | python3 -c "import json,sys; d=json.load(sys.stdin); print('queues $q, passes in flight $s: headline', d['value'])"
  done
done
for q in 4 8 16; do
  for f in 4 8; do
    GPU_MAX_HW_QUEUES=$q FRCNN_ENTRY_IN_FLIGHT=$f python3 scripts/dev/r6_entry_host_share.py 2>&1 | grep "get_dets_by_cls over" | sed "s/^/queues $q, entry passes in flight $f: /"
  done
done
