#!/usr/bin/env python3
"""The head's 512 -> 2048 layer (58 800 rows, f32 residual) under FRCNN_GROUP_M (tile order inside an XCD's run): time, and -- under
rocprofv3 --pmc FETCH_SIZE -- the memory-side reads (x2: scripts/micro/fetch_size_calibration.py).  Dev tool."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from faster_rcnn_amd import ops

rs = np.random.RandomState(0)
M, cin, cout = 58800, 512, 2048
x = torch.from_numpy(rs.randn(1, M // 100, 100, cin).astype(np.float32)).cuda()
wt = (rs.randn(1, 1, cin, cout) * np.sqrt(2.0 / cin)).astype(np.float32)
pc = ops.PackedConv(wt, np.ones(cout, np.float32), np.zeros(cout, np.float32))
res = torch.from_numpy(rs.randn(1, M // 100, 100, cout).astype(np.float32)).cuda()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
with ops.conv_workspace(ops.NO_SPLIT_K):
    y = ops.conv2d(x, pc, 1, "valid", "relu", residual=res, tile=86)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ops.conv2d(x, pc, 1, "valid", "relu", residual=res, out=y, tile=86)
    e1.record()
    torch.cuda.synchronize()
print("GROUP_M=%s: %.1f us" % (os.environ.get("FRCNN_GROUP_M", "auto"), e0.elapsed_time(e1) / reps * 1e3))
