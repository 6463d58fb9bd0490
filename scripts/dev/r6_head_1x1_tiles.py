#!/usr/bin/env python3
"""The head's 512 -> 2048 layer (58 800 rows: four images x 300 RoIs x 49) with a residual, f32 input, on the f16x3 engine's tile
forms: does a form with TWO workgroups per CU (128x128, single LDS buffer) overlap its epilogue with the neighbour's main loop where
the 256x128 double-buffered form (one workgroup per CU) cannot?   Dev tool."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from faster_rcnn_amd import ops

rs = np.random.RandomState(0)
for (M, cin, cout) in [(58800, 512, 2048), (58800, 2048, 512), (14700, 512, 2048)]:
    x = torch.from_numpy(rs.randn(1, 1, M, cin).astype(np.float32)).cuda().reshape(1, M // 100, 100, cin)
    wt = (rs.randn(1, 1, cin, cout) * np.sqrt(2.0 / cin)).astype(np.float32)
    pc = ops.PackedConv(wt, np.ones(cout, np.float32), np.zeros(cout, np.float32))
    res = torch.from_numpy(rs.randn(1, M // 100, 100, cout).astype(np.float32)).cuda()
    for tile in (86, 82, 81, 83, 0):
        with ops.conv_workspace(ops.NO_SPLIT_K):
            y = ops.conv2d(x, pc, 1, "valid", "relu", residual=res, tile=tile)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                ops.conv2d(x, pc, 1, "valid", "relu", residual=res, out=y, tile=tile)
            e1.record()
            torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        print("M=%d %d->%d tile=%d: %.1f us  %.1f TFLOP/s" % (M, cin, cout, tile, us, 2.0 * M * cout * cin / us / 1e6))
