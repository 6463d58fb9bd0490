GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
  python3 scripts/bench_train.py --bf16 --steps 100 --warmup 40 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('mixed default: rpn %.3f ms  det %.3f ms' % (d['rpn_step1']['ms_per_step'], d['det_step2']['ms_per_step']))"
  FRCNN_TRAIN_F32_ENGINE=native FRCNN_TRAIN_WGRAD=native python3 scripts/bench_train.py --bf16 --steps 100 --warmup 40 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('mixed, native f32 knobs: rpn %.3f ms  det %.3f ms' % (d['rpn_step1']['ms_per_step'], d['det_step2']['ms_per_step']))"
done
