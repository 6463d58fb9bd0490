#!/usr/bin/env python3
"""RoI crop-resize backward (gather form) at the detector training step's shape: 64 RoIs, 7x7, 1024 channels, 38x63 map.  Dev tool."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from faster_rcnn_amd import ops
rs = np.random.RandomState(0)
rows, cols, n = 38, 63, 64
for tag, wmax in (("small boxes (bench_train)", 7), ("large boxes", 40)):
    x1 = rs.randint(0, cols - wmax - 1, n); y1 = rs.randint(0, rows - min(wmax, 30) - 1, n)
    rois = np.stack([x1, y1, x1 + 1 + rs.randint(0, wmax, n), y1 + 1 + rs.randint(0, min(wmax, 30), n)], axis=1).astype(np.float32)
    r = torch.from_numpy(rois).cuda()
    for dt in (torch.float32, torch.bfloat16):
        g = torch.randn(n, 7, 7, 1024, device="cuda").to(dt)
        fn = ops.roi_crop_resize_bwd_bf16 if dt == torch.bfloat16 else ops.roi_crop_resize_bwd
        for _ in range(3): fn(g, r, rows, cols)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): fn(g, r, rows, cols)
        e1.record(); e1.synchronize()
        print("%-28s %-8s %7.1f us" % (tag, str(dt).split(".")[-1], e0.elapsed_time(e1) * 1e3 / 50))
