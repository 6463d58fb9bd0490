#!/bin/bash
# f32 weight gradient: 128x128-tile kernel vs the 64x64 one (FRCNN_WGRAD_BIG), slice target sweep
cd "$(dirname "$0")/../.."
O=gpurun_out/run18; mkdir -p $O
python -m pytest tests/test_conv_bwd_gpu.py tests/test_train_gpu.py -m gpu -q -x > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt
tail -5 $O/pytest.txt
run() { echo "== $*"; env "$@" python scripts/bench_train.py --steps 30 --warmup 8 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1])
print({k:(v['ms_per_step'] if isinstance(v,dict) and 'ms_per_step' in v else None) for k,v in d.items() if isinstance(v,dict)})"; }
run FRCNN_WGRAD_BIG=0
run FRCNN_WGRAD_BIG=1 FRCNN_WGRAD_TARGET_BIG=64
run FRCNN_WGRAD_BIG=1 FRCNN_WGRAD_TARGET_BIG=128
run FRCNN_WGRAD_BIG=1 FRCNN_WGRAD_TARGET_BIG=256
run FRCNN_WGRAD_BIG=0
run FRCNN_WGRAD_BIG=1 FRCNN_WGRAD_TARGET_BIG=128
