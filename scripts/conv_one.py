#!/usr/bin/env python3
"""Run ONE conv shape repeatedly (for rocprofv3 --pmc / --kernel-trace).  Dev tool.
usage: conv_one.py n h w cin cout k stride padding tile iters [layout]   (layout 1: position-major x (h,w,n,cin))"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from faster_rcnn_amd import ops

n, h, w, cin, cout, k, stride = (int(v) for v in sys.argv[1:8])
padding, tile, iters = sys.argv[8], int(sys.argv[9]), int(sys.argv[10])
layout = int(sys.argv[11]) if len(sys.argv) > 11 else 0
rs = np.random.RandomState(0)
x = torch.from_numpy(rs.randn(*((h, w, n, cin) if layout else (n, h, w, cin))).astype(np.float32)).cuda()
wt = (rs.randn(k, k, cin, cout) * np.sqrt(2.0 / (k * k * cin))).astype(np.float32)
pc = ops.PackedConv(wt, np.ones(cout, np.float32), np.zeros(cout, np.float32))
y = ops.conv2d(x, pc, stride, padding, "relu", tile=tile, layout=layout)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters):
    ops.conv2d(x, pc, stride, padding, "relu", out=y, tile=tile, layout=layout)
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) / iters * 1e3
M = y.shape[0] * y.shape[1] * y.shape[2]
print("M=%d N=%d K=%d tile=%d: %.1f us  %.1f TF/s" % (M, cout, k * k * cin, tile, us, 2.0 * M * cout * k * k * cin / us / 1e6))
