import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from faster_rcnn_amd import ops
rs = np.random.RandomState(0)
n, C = 300, 21
def t(f, it=20):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / it * 1e3
x1 = rs.randint(0, 50, n); y1 = rs.randint(0, 30, n)
rois = torch.from_numpy(np.stack([x1, y1, x1 + 1 + rs.randint(0, 10, n), y1 + 1 + rs.randint(0, 8, n)], 1).astype(np.float32)).cuda()
nr = torch.tensor([n], dtype=torch.int32, device="cuda")
reg = torch.from_numpy((rs.randn(n, 4 * (C - 1)) * 0.5).astype(np.float32)).cuda()
for name, logits in (("all one class", np.tile(np.eye(C)[3] * 5, (n, 1)) + rs.rand(n, C) * 0.1), ("uniform classes", rs.rand(n, C) * 3), ("mostly background", np.tile(np.eye(C)[C - 1] * 5, (n, 1)) + rs.rand(n, C))):
    p = torch.softmax(torch.from_numpy(logits.astype(np.float32)), 1).cuda()
    for thr in (0.0, 2.0):
        us = t(lambda: ops.detections(rois, nr, p, reg, 64, C - 1, thr, 16.0, 1.6))
        out = ops.detections(rois, nr, p, reg, 64, C - 1, thr, 16.0, 1.6)
        print("%-18s det_threshold %.0f: %6.1f us (incl. ~6 small fills), n_dets %d" % (name, thr, us, int(out["n_dets"].item())))
