#!/bin/bash
# from-JPEG-files leg of the entry point on 8 hardware queues: inline decode against decode threads (forced by a low inline threshold)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
for rep in 1 2; do
for v in "4.0 4" "0.1 1" "0.1 2" "0.1 4"; do
  set -- $v
  GPU_MAX_HW_QUEUES=8 FRCNN_DECODE_INLINE_MS=$1 FRCNN_DECODE_THREADS=$2 python3 bench.py --entry-only 2>/dev/null | python3 -c "
import json,sys; v=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('inline threshold $1 ms, threads $2: files', v['from_files']['value'], ' 256 frames', v['long_list']['value'])"
done
done
