#!/usr/bin/env python3
"""Time ONE bf16 conv shape (optionally with a residual), alone on the chip.  Dev tool.
usage: bf16_one.py n h w cin cout k stride padding tile iters layout residual(0/1)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from faster_rcnn_amd import ops

n, h, w, cin, cout, k, stride = (int(v) for v in sys.argv[1:8])
padding, tile, iters, layout, res = sys.argv[8], int(sys.argv[9]), int(sys.argv[10]), int(sys.argv[11]), int(sys.argv[12])
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.randn(((h, w, n, cin) if layout else (n, h, w, cin)), device="cuda", generator=g).to(torch.bfloat16)
rs = np.random.RandomState(0)
wt = (rs.randn(k, k, cin, cout) * np.sqrt(2.0 / (k * k * cin))).astype(np.float32)
pc = ops.PackedConvBf16(wt, np.ones(cout, np.float32), np.zeros(cout, np.float32))
y = ops.conv2d_bf16(x, pc, stride, padding, "relu", tile=tile, layout=layout)
r = torch.randn(y.shape, device="cuda", generator=g).to(torch.bfloat16) if res else None
for _ in range(3):
    ops.conv2d_bf16(x, pc, stride, padding, "relu", residual=r, tile=tile, layout=layout)
torch.cuda.synchronize()
best = 1e9
for rep in range(4):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        ops.conv2d_bf16(x, pc, stride, padding, "relu", residual=r, tile=tile, layout=layout)
    e1.record()
    torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) / iters * 1e3)
M = y.shape[0] * y.shape[1] * y.shape[2]
by = M * cout * 2 * (2 if res else 1) + M * cin * 2 * (1 if k == 1 else 1)
print("M=%d N=%d K=%d tile=%d res=%d %s: %.1f us  %.1f TF/s  %.2f TB/s (out%s + in bytes)" % (
    M, cout, k * k * cin, tile, res, os.environ.get("FRCNN_BF16_STAGGER", ""), best, 2.0 * M * cout * k * k * cin / best / 1e6, by / best / 1e6, "+res" if res else ""))
