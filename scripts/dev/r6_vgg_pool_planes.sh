#!/bin/bash
# configs[0]: plane tensors between a VGG block's convolutions AND out of its max-pools (default) against f32 everywhere
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
python3 -m pytest tests/test_conv_h3_gpu.py tests/test_nets_gpu.py -q -x 2>&1 | tail -2
for rep in 1 2; do
  for v in 1 0; do
    FRCNN_VGG_PLANES=$v python3 bench.py --config c1 --steps 30 --warmup 5 --no-cpu-baseline --no-io --no-extra 2>/dev/null | python3 -c "import json,sys; d=json.load(sys.stdin); print('VGG planes=$v:', d['value'], d['roofline']['frac'])"
  done
done
python3 bench.py --config c1 --steps 20 --warmup 5 --no-extra 2>/dev/null | python3 -c "import json,sys; d=json.load(sys.stdin); print('full line:', d['value'], d['parity'])"
