"""dev: the split-engine weight gradient against fp64 and against the native kernel, and its time on the training steps' layers."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from faster_rcnn_amd import ops
F = torch.nn.functional
def ref(x, g, k, stride, padding):
    xt = torch.from_numpy(x).double().permute(0, 3, 1, 2).requires_grad_(False)
    cin, cout = x.shape[-1], g.shape[-1]
    w = torch.zeros(cout, cin, k, k, dtype=torch.float64, requires_grad=True)
    if padding == "same":
        _, pt = ops.same_pad(x.shape[1], k, stride); _, pl = ops.same_pad(x.shape[2], k, stride)
        ho, wo = -(-x.shape[1] // stride), -(-x.shape[2] // stride)
        pb = max((ho - 1) * stride + k - x.shape[1] - pt, 0); pr = max((wo - 1) * stride + k - x.shape[2] - pl, 0)
        xt = F.pad(xt, (pl, pr, pt, pb))
        xa = F.pad(torch.from_numpy(np.abs(x)).double().permute(0, 3, 1, 2), (pl, pr, pt, pb))
    else:
        xa = torch.from_numpy(np.abs(x)).double().permute(0, 3, 1, 2)
    gt = torch.from_numpy(g).double().permute(0, 3, 1, 2)
    y = F.conv2d(xt, w, stride=stride); (y * gt).sum().backward()
    dw = w.grad.permute(2, 3, 1, 0).numpy()
    w2 = torch.zeros_like(w, requires_grad=True)
    y2 = F.conv2d(xa, w2, stride=stride); (y2 * gt.abs()).sum().backward()
    return dw, w2.grad.permute(2, 3, 1, 0).numpy()
rs = np.random.RandomState(0)
for (n, h, w, cin, cout, k, stride, padding) in [(1, 38, 63, 256, 256, 3, 1, "same"), (1, 38, 63, 1024, 256, 1, 1, "valid"), (1, 38, 63, 256, 1024, 1, 1, "valid"),
                                                 (1, 75, 125, 512, 256, 1, 2, "valid"), (64, 7, 7, 512, 512, 3, 1, "same"), (1, 20, 31, 128, 192, 3, 1, "same"), (2, 17, 23, 160, 136, 3, 2, "same")]:
    x = rs.randn(n, h, w, cin).astype(np.float32)
    ho, wo = (-(-h // stride), -(-w // stride)) if padding == "same" else ((h - k) // stride + 1, (w - k) // stride + 1)
    g = (rs.randn(n, ho, wo, cout) * 0.1).astype(np.float32)
    xd, gd = torch.from_numpy(x).cuda(), torch.from_numpy(g).cuda()
    out = {}
    for eng in ("native", "bf16x6"):
        ops.WGRAD_ENGINE = eng
        dw, _ = ops.conv2d_wgrad(xd, gd, k, k, stride, padding, want_bias=False)
        dw2, _ = ops.conv2d_wgrad(xd, gd, k, k, stride, padding, want_bias=False)
        assert torch.equal(dw, dw2)
        torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): ops.conv2d_wgrad(xd, gd, k, k, stride, padding, dw=dw2, want_bias=False)
        e1.record(); torch.cuda.synchronize()
        out[eng] = (dw.cpu().numpy(), e0.elapsed_time(e1) / 10 * 1e3)
    r, mag = ref(x, g, k, stride, padding)
    en = float((np.abs(out["native"][0] - r) / np.maximum(mag, 1e-30)).max()); ex = float((np.abs(out["bf16x6"][0] - r) / np.maximum(mag, 1e-30)).max())
    fl = 2.0 * n * ho * wo * cin * cout * k * k
    print("x %s k%d s%d %s -> cout %d: native %.1f us (%.0f TF) err %.3g | bf16x6 %.1f us (%.0f TF) err %.3g" % (
        (n, h, w, cin), k, stride, padding, cout, out["native"][1], fl / out["native"][1] / 1e6, en, out["bf16x6"][1], fl / out["bf16x6"][1] / 1e6, ex), flush=True)
