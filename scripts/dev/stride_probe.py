"""dev: does the row stride (cin * 2 bytes: a power of two on every ResNet layer) cost L2 -> LDS throughput?  Same GEMM with cin 512 vs 576, 2048 vs 2112."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from faster_rcnn_amd import ops
def run(M, cin, cout, tile, iters=20):
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.randn((1, M, 1, cin), device="cuda", generator=g).to(torch.bfloat16)
    wt = (np.random.RandomState(0).randn(1, 1, cin, cout) * np.sqrt(2.0 / cin)).astype(np.float32)
    pc = ops.PackedConvBf16(wt, None, None)
    for _ in range(3): ops.conv2d_bf16(x, pc, 1, "valid", "relu", tile=tile)
    torch.cuda.synchronize(); best = 1e9
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): ops.conv2d_bf16(x, pc, 1, "valid", "relu", tile=tile)
        e1.record(); torch.cuda.synchronize(); best = min(best, e0.elapsed_time(e1) / iters * 1e3)
    opb = M * cout * cin * 2 * (2 / 128.0)
    print("M=%d %d->%d tile %d: %.1f us  %.0f TF  operand stream %.1f TB/s (128x128 tiles)" % (M, cin, cout, tile, best, 2.0 * M * cin * cout / best / 1e6, opb / best / 1e6), flush=True)
for tile in (47, 55):
    for cin in (512, 576, 2048, 2112, 1024, 1088):
        run(117600, cin, 512, tile)
