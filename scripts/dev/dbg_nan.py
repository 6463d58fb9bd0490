import numpy as np, torch, sys
sys.path.insert(0, ".")
from faster_rcnn_amd import _lib, ops
rs = np.random.RandomState(7)
x = rs.randn(1, 24, 24, 64).astype(np.float32)
wt = (rs.randn(3, 3, 64, 128) * 0.05).astype(np.float32)
pc = ops.PackedConv(wt)
for bad in (np.nan, np.inf):
    xb = x.copy(); xb[0, 10, 11, 5] = bad
    xd = torch.from_numpy(xb).cuda()
    rec = ops.amax_of(xd); xd._amax = rec
    with ops.conv_workspace(ops.NO_SPLIT_K):
        got = ops.conv2d(xd, pc, 1, "same", tile=84)
        nat = ops.conv2d(xd, pc, 1, "same", tile=0)
    g, n = got.cpu().numpy(), nat.cpu().numpy()
    print(bad, "rec max", float(rec.max()), "status", int(rec.view(torch.int32)[1].item()), "nan g", np.isnan(g).sum(), "nan n", np.isnan(n).sum(), "inf g", np.isinf(g).sum(), "inf n", np.isinf(n).sum())
    print("  g at center", g[0, 10, 11, :4], "n", n[0, 10, 11, :4])
