GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $GRAFT_REPO_ROOT
for t in 74 0; do
  python3 scripts/conv_one.py 64 7 7 512 512 3 1 same $t 30 1
  python3 scripts/conv_one.py 1 38 63 1024 512 3 1 same $t 30
  python3 scripts/conv_one.py 1 38 63 256 256 3 1 same $t 50
done
export FRCNN_BENCH_NO_ENTRY=1 FRCNN_BENCH_NO_NATIVE=1
for i in 1 2; do
python3 bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-io 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('headline', d['value'])"
python3 scripts/bench_train.py --steps 60 --warmup 40 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('fp32 steps: rpn %.3f ms  det %.3f ms' % (d['rpn_step1']['ms_per_step'], d['det_step2']['ms_per_step']))"
done
