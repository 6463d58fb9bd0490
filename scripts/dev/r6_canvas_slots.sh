#!/bin/bash
# how many captured passes a canvas class may hold: factor x in_flight x its share of the list (min 1 or 2)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
for rep in 1 2; do
for fm in "1.5 2" "1.0 1" "1.0 2" "0.75 1"; do
  set -- $fm
  GPU_MAX_HW_QUEUES=8 FRCNN_ENTRY_CANVAS_SLOT_FACTOR=$1 FRCNN_ENTRY_CANVAS_SLOT_MIN=$2 python3 bench.py --entry-only 2>/dev/null | python3 -c "
import json,sys; v=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); m=v['mixed_sizes']
print('factor $1 min $2: mixed first', m['first_call']['value'], 'captures', m['first_call']['captures'], 'again', m['second_call']['value'], 'GB', round(m['graph_cache_bytes']/1e9,1))"
done
done
