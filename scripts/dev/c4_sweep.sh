GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $GRAFT_REPO_ROOT
for cfg in "4 8 4" "2 16 4" "3 8 4" "4 12 4" "4 16 4" "6 8 8" "8 4 8" "2 8 4"; do
  set -- $cfg
  GPU_MAX_HW_QUEUES=$3 timeout 300 python3 bench.py --config c4 --streams $1 --batch $2 --steps 40 --warmup 8 --no-cpu-baseline --no-io 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('graphs $1 x batch $2, queues $3:', d['value'], 'img/s')"
done
