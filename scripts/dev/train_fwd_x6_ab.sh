GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $GRAFT_REPO_ROOT
for e in native bf16x6 native bf16x6; do
  python3 -c "
import sys, runpy
sys.path.insert(0, '.')
from faster_rcnn_amd import ops
ops.F32_ENGINE = '$e'
sys.argv = ['bench_train.py', '--steps', '60', '--warmup', '40']
runpy.run_path('scripts/bench_train.py', run_name='__main__')
" 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('forward engine $e: rpn %.3f ms  det %.3f ms' % (d['rpn_step1']['ms_per_step'], d['det_step2']['ms_per_step']))"
done
