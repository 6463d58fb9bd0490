#!/bin/bash
# GPU_MAX_HW_QUEUES (default 4) for the other lines: configs[3] headline loop, training loops replayed from graphs
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
for q in 4 8 16; do
  GPU_MAX_HW_QUEUES=$q python3 bench.py --config c4 --steps 12 --warmup 3 --no-cpu-baseline --no-io --no-extra 2>/dev/null | python3 -c "import json,sys; d=json.load(sys.stdin); print('queues $q: configs[3]', d['value'])"
  GPU_MAX_HW_QUEUES=$q python3 scripts/bench_train.py --bf16 --through-loop --no-host-feed --steps 60 --warmup 40 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('queues $q: mixed', {k:(d[k]['ms_per_step'], d[k]['through_loop']['fast_feed']['ms_per_iteration']) for k in ('rpn_step1','det_step2')})"
  GPU_MAX_HW_QUEUES=$q python3 scripts/bench_train.py --through-loop --no-host-feed --steps 60 --warmup 40 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('queues $q: f32', {k:(d[k]['ms_per_step'], d[k]['through_loop']['fast_feed']['ms_per_iteration']) for k in ('rpn_step1','det_step2')})"
done
