import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from faster_rcnn_amd import ops
from oracle import keras_ref
rs = np.random.RandomState(0)
for (n, h, w, cin, cout, k) in [(1, 12, 16, 512, 512, 3), (1, 6, 8, 512, 512, 3), (1, 12, 16, 512, 512, 1), (1, 24, 32, 256, 256, 3)]:
    x = rs.randn(n, h, w, cin).astype(np.float32)
    wt = (rs.randn(k, k, cin, cout) / np.sqrt(k * k * cin)).astype(np.float32)
    want = keras_ref.conv2d(x, wt, None, 1, "same", dtype=torch.float64)
    pc = ops.PackedConv(wt)
    for tile in (2, 12, 22, 11, 21):
        got = ops.conv2d(torch.from_numpy(x).cuda(), pc, 1, "same", tile=tile).cpu().double()
        err = ((got - want).abs() / want.abs().clamp(min=1)).max().item()
        print((n, h, w, cin, cout, k), "tile", tile, "err %.2e" % err)
