#!/usr/bin/env python3
"""The head's 3x3 layer (512 -> 512 over 1 200 RoIs x 7 x 7, fp16-plane input and output, [roi][7][7][c] rows) alone on the chip.  Dev tool."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from faster_rcnn_amd import ops

tag = sys.argv[1] if len(sys.argv) > 1 else ""
rs = np.random.RandomState(0)
n, c = 1200, 512
x = torch.from_numpy(rs.randn(n, 7, 7, c).astype(np.float32)).cuda()
w1 = (rs.randn(1, 1, c, c) * np.sqrt(2.0 / c)).astype(np.float32)
w3 = (rs.randn(3, 3, c, c) * np.sqrt(2.0 / (9 * c))).astype(np.float32)
pc1 = ops.PackedConv(w1, np.ones(c, np.float32), np.zeros(c, np.float32))
pc3 = ops.PackedConv(w3, np.ones(c, np.float32), np.zeros(c, np.float32))
arena = ops.AmaxArena()
with ops.f32_engine("f16x3"), ops.conv_workspace(ops.NO_SPLIT_K), ops.amax_arena(arena), ops.tile_policy(True):
    t = ops.conv2d(x, pc1, 1, "same", "relu", planes_out=True)          # a producing launch writes the planes, as branch2a does
    y = ops.conv2d(t, pc3, 1, "same", "relu", planes_out=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        y = ops.conv2d(t, pc3, 1, "same", "relu", planes_out=True)
    e1.record()
    torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 20 * 1e3
print("%s 3x3 512->512 over %d rows: %.1f us  %.1f TFLOP/s" % (tag, n * 49, us, 2.0 * n * 49 * c * c * 9 / us / 1e6))
