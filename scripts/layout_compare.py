import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from faster_rcnn_amd import ops
rs = np.random.RandomState(0)
def timeit(f, it=10, rounds=4):
    best = 1e9
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(it): f()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / it * 1e3)
    return best
for (n, cin, cout, k) in [(300, 512, 512, 3), (300, 512, 2048, 1), (300, 2048, 512, 1), (300, 1024, 512, 1)]:
    x = torch.from_numpy(rs.randn(n, 7, 7, cin).astype(np.float32)).cuda()
    xp = x.permute(1, 2, 0, 3).contiguous()
    wt = (rs.randn(k, k, cin, cout) * 0.02).astype(np.float32)
    pc = ops.PackedConv(wt, np.ones(cout, np.float32), np.zeros(cout, np.float32))
    pad = "same" if k == 3 else "valid"
    y0 = ops.conv2d(x, pc, 1, pad, "relu"); y1 = ops.conv2d(xp, pc, 1, pad, "relu", layout=1)
    for tile in (21, 22, 61, 62):
        code = tile if tile > 60 else 100 + tile                 # 61 / 62: balanced launch; 1xx: never split
        ws = ops.ConvWorkspace()
        with ops.conv_workspace(ws):
            t0 = timeit(lambda: ops.conv2d(x, pc, 1, pad, "relu", out=y0, tile=code))
            t1 = timeit(lambda: ops.conv2d(xp, pc, 1, pad, "relu", out=y1, tile=code, layout=1))
        fl = 2.0 * n * 49 * cin * cout * k * k
        print("n=%d cin=%d cout=%d k=%d tile=%d  nhwc %.1f us (%.1f TF)  pos-major %.1f us (%.1f TF nominal)" % (n, cin, cout, k, tile, t0, fl / t0 / 1e6, t1, fl / t1 / 1e6))
