#!/bin/bash
# get_dets_by_cls: passes submitted beyond the engine's streams before the oldest is collected (FRCNN_ENTRY_WINDOW_EXTRA)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
for rep in 1 2; do
  for v in 0 1 2; do
    FRCNN_ENTRY_WINDOW_EXTRA=$v python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra 2>/dev/null | python3 -c "
import json,sys; d=json.load(sys.stdin); v=d['via_reference_entry']
print('window extra $v: 32 frames', v['value'], ' 256 frames', v['long_list']['value'], ' files', v['from_files']['value'], ' mixed', v['mixed_sizes']['first_call']['value'], v['mixed_sizes']['second_call']['value'], 'graphs', v['graph_cache']['graphs'])"
  done
done
