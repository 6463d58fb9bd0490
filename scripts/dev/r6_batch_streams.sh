#!/bin/bash
# images per captured pass x passes in flight, with round 6's shared-chip tile policy
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
for bs in "4 4" "8 2" "8 3" "6 3" "4 6" "2 8" "4 4"; do
  set -- $bs
  python3 bench.py --batch $1 --streams $2 --steps 20 --warmup 4 --no-cpu-baseline --no-io --no-extra 2>/dev/null | python3 -c "import json,sys; d=json.load(sys.stdin); print('images per pass $1 x passes in flight $2:', d['value'])"
done
