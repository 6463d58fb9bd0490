#!/usr/bin/env python3
"""What does FETCH_SIZE count for THIS repo's access patterns?  (MI355X_MICROARCH.md: 'FETCH_SIZE reports exactly 1/2 of the bytes of a wide
coalesced streaming read ... other access widths are uncalibrated: calibrate on a known byte count in your own access pattern'.)
Known byte counts, run under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (scripts/micro/fetch_size_calibration.sh):
  * a 1 GiB float tensor copied by torch (16 B / lane streaming read, beyond the 256 MiB Infinity Cache)
  * the f16x3 256x128 conv form on 58 800 rows, 512 -> cout in {128, 512, 2048}, with and without the f32 residual: the A operand
    (120 MB) is the same in all of them, the residual and the output grow with cout -- the slope per output byte calibrates the residual read,
    the intercept the A read (and shows whether the column tiles of a row tile re-read it from memory)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from faster_rcnn_amd import ops

x1 = torch.empty(1 << 28, dtype=torch.float32, device="cuda").fill_(1.0)          # 1 GiB
y1 = x1.clone()
torch.cuda.synchronize()
rs = np.random.RandomState(0)
M, cin = 58800, 512
x = torch.from_numpy(rs.randn(1, M // 100, 100, cin).astype(np.float32)).cuda()
for cout in (128, 512, 2048):
    wt = (rs.randn(1, 1, cin, cout) * np.sqrt(2.0 / cin)).astype(np.float32)
    pc = ops.PackedConv(wt, np.ones(cout, np.float32), np.zeros(cout, np.float32))
    res = torch.from_numpy(rs.randn(1, M // 100, 100, cout).astype(np.float32)).cuda()
    with ops.conv_workspace(ops.NO_SPLIT_K):
        ops.conv2d(x, pc, 1, "valid", "relu", tile=86)
        ops.conv2d(x, pc, 1, "valid", "relu", residual=res, tile=86)
    torch.cuda.synchronize()
    print("cout %d: A %.1f MB, weights %.1f MB, residual = output %.1f MB" % (cout, M * cin * 4 / 1e6, cin * cout * 4 / 1e6, M * cout * 4 / 1e6))
