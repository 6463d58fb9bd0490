import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch, ctypes
from faster_rcnn_amd import _lib, ops
rs = np.random.RandomState(0)
cells, A = 38 * 63, 9
yc = torch.from_numpy((rs.rand(cells, 2 * A) < 0.1).astype(np.float32)).cuda()
p = torch.from_numpy(rs.rand(cells, A).astype(np.float32)).cuda()
yr = torch.from_numpy(rs.randn(cells, 8 * A).astype(np.float32)).cuda()
pr = torch.from_numpy(rs.randn(cells, 4 * A).astype(np.float32)).cuda()
l = torch.zeros(1, device="cuda"); g1 = torch.empty_like(p); g2 = torch.empty_like(pr)
P = lambda t: ctypes.c_void_p(t.data_ptr())
def t(f, n=20):
    f(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3
print("rpn_cls %.1f us" % t(lambda: _lib.call("frcnn_loss_rpn_cls", P(yc), P(p), cells, A, P(l), P(g1), ops._stream())))
print("rpn_reg %.1f us" % t(lambda: _lib.call("frcnn_loss_rpn_reg", P(yr), P(pr), cells, A, P(l), P(g2), ops._stream())))
