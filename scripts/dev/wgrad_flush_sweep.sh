GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $GRAFT_REPO_ROOT
for f in 8 3 4 6 12 8; do
  for m in "" "--bf16"; do
    FRCNN_WGRAD_FLUSH=$f python3 scripts/bench_train.py $m --steps 60 --warmup 40 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('flush $f $m: rpn %.3f ms  det %.3f ms' % (d['rpn_step1']['ms_per_step'], d['det_step2']['ms_per_step']))"
  done
done
