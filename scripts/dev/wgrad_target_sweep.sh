GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $GRAFT_REPO_ROOT
for t in 256 128 96 64 256 128; do
  FRCNN_WGRAD_TARGET=$t python3 scripts/bench_train.py --bf16 --steps 60 --warmup 40 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('mixed: wgrad target $t: rpn %.3f ms  det %.3f ms' % (d['rpn_step1']['ms_per_step'], d['det_step2']['ms_per_step']))"
done
