#!/bin/bash
# the 64-column layers beside other passes' launches: 128x64 on four waves (87) against 256x64 on eight (88); configs[0] and configs[1]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
python3 - <<'PY'
import numpy as np, torch, sys
sys.path.insert(0, ".")
from faster_rcnn_amd import ops
rs = np.random.RandomState(3)
x = torch.from_numpy(np.maximum(rs.randn(2, 150, 250, 64), 0).astype(np.float32)).cuda()
pc = ops.PackedConv((rs.randn(3, 3, 64, 64) * 0.06).astype(np.float32), (1 + 0.1 * rs.randn(64)).astype(np.float32), (0.1 * rs.randn(64)).astype(np.float32))
res = torch.from_numpy(rs.randn(2, 150, 250, 64).astype(np.float32)).cuda()
with ops.f32_engine("f16x3"), ops.conv_workspace(ops.NO_SPLIT_K):
    a = ops.conv2d(x, pc, 1, "same", "relu", residual=res, tile=87)
    b = ops.conv2d(x, pc, 1, "same", "relu", residual=res, tile=88)
print("256x64 form == 128x64 form bit for bit:", bool(torch.equal(a, b)), float(a.abs().max()))
PY
for rep in 1 2; do
  for v in 87 88; do
    FRCNN_H3_N64_SHARED=$v python3 bench.py --config c1 --steps 20 --warmup 4 --no-cpu-baseline --no-io --no-extra 2>/dev/null | python3 -c "import json,sys; d=json.load(sys.stdin); print('64-column tile $v: c0', d['value'])"
    FRCNN_H3_N64_SHARED=$v python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-io --no-extra 2>/dev/null | python3 -c "import json,sys; d=json.load(sys.stdin); print('64-column tile $v: c2', d['value'], d['roofline']['backbone_conv']['in_flight']['ms_per_image'])"
  done
done
