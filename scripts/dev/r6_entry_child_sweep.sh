#!/bin/bash
# bench.py --entry-only (the legs through voc_dets.get_dets_by_cls, in a process of their own) under hardware-queue counts x passes in flight
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
for qf in "12 12" "16 12" "16 8" "16 6" "24 12" "16 4"; do
  set -- $qf
  GPU_MAX_HW_QUEUES=$1 FRCNN_ENTRY_IN_FLIGHT=$2 python3 bench.py --entry-only 2>/dev/null | python3 -c "
import json,sys; v=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
m=v['mixed_sizes']; x=v['mixed_sizes_exact_geometry_passes']
print('queues $1 in flight $2: 32 frames', v['value'], ' 256 frames', v['long_list']['value'], ' files', v['from_files']['value'], ' mixed', m['first_call']['value'], m['second_call']['value'], 'captures', m['first_call']['captures'], ' exact', x['first_call']['value'], x['second_call']['value'], ' GB', round(v['graph_cache']['bytes']/1e9,1))"
done
