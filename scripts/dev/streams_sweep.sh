GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $GRAFT_REPO_ROOT
export FRCNN_BENCH_NO_ENTRY=1 FRCNN_BENCH_NO_NATIVE=1
for cfg in "8 8" "6 8" "10 8" "12 8" "12 12" "16 8" "16 16"; do
  set -- $cfg
  GPU_MAX_HW_QUEUES=$2 python3 bench.py --streams $1 --steps 60 --warmup 10 --no-cpu-baseline --no-io 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('streams $1 queues $2:', d['value'], 'img/s')"
done
