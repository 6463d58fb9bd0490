import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from tests.test_train_gpu import image
from faster_rcnn_amd import vgg, train, nets, ops
from faster_rcnn_amd.weights import synthetic_vgg16
from oracle.keras_ref import KerasGraphs
from oracle import keras_train_ref as kt
x = image(96, 128, seed=5)
w0 = synthetic_vgg16(seed=22, with_classifier=False)
names = kt.vgg_conv_names((3, 4, 5))
w, params = kt._prepare(w0, set(names), torch.float64)
g = KerasGraphs(w, torch.float64)
feat = g.vgg_base(x)
rs = np.random.RandomState(1)
R = rs.randn(*feat.shape)
L = (feat * torch.from_numpy(R)).sum()
grads = dict(zip(params.keys(), torch.autograd.grad(L, list(params.values()))))
base = vgg.vgg16_base(weights={k: [a.copy() for a in v] for k, v in w0.items()})
ps = train.ParamSet(base.weights, names)
bt = train.VggBaseTrain(base, ps)
f = bt.forward(nets.to_device_image(x))
print("feat err", float((f.cpu().double() - feat).abs().max()))
gR = torch.from_numpy(R.astype(np.float32)).cuda() * (f > 0)
bt.backward(gR.contiguous())
for n in names:
    for i in (0, 1):
        got = ps.views[n][i][1].cpu().double()
        want = grads[(n, i)]
        print("%-14s %d relfro %.2e" % (n, i, float((got - want).norm() / want.norm())))
# ---- intermediate gradients: oracle d L / d (pre-activation) for each block-4/5 conv
from oracle.keras_ref import conv2d, pool2d
ws, _ = kt._prepare(w0, set(), torch.float64)
t = torch.tensor(x, dtype=torch.float64)
pre = {}
for blk, n in ((1, 2), (2, 2), (3, 3), (4, 3), (5, 3)):
    for i in range(1, n + 1):
        nm = "block%d_conv%d" % (blk, i)
        z = conv2d(t, ws[nm][0], ws[nm][1], 1, "same", torch.float64)
        z.requires_grad_(True) if not z.requires_grad else None
        z.retain_grad()
        pre[nm] = z
        t = z.clamp(min=0)
    if blk < 5:
        t = pool2d(t, 2, 2, True)
L2 = (t * torch.from_numpy(R)).sum()
L2.backward()
# product-side intermediates: re-run backward by hand
layers = {n: l for n, l, tr in bt.layers if tr}
gcur = gR.contiguous()
for nm in ["block5_conv3", "block5_conv2", "block5_conv1", "block4_conv3", "block4_conv2", "block4_conv1"]:
    l = layers[nm]
    if nm in bt.POOL_AFTER:
        xx, yy = bt.pools[nm]
        gx = torch.empty_like(xx)
        from faster_rcnn_amd import _lib
        from faster_rcnn_amd.ops import _p, _stream
        _lib.call("frcnn_maxpool_bwd", _p(xx), _p(yy), _p(gcur.contiguous()), xx.shape[0], xx.shape[1], xx.shape[2], xx.shape[3], 2, _p(gx), _stream())
        _lib.call("frcnn_relu_bwd_inplace", _p(gx), _p(xx), gx.numel(), _stream())
        gcur = gx
    want = pre[nm].grad
    got = gcur.cpu().double()
    d = (got - want)
    print(nm, "g relfro %.2e" % float(d.norm() / want.norm()), "nnz got %d want %d" % (int((got != 0).sum()), int((want != 0).sum())),
          "mask mismatch", int(((got != 0) != (want != 0)).sum()))
    src = l.x
    prev_pooled = any(src is py for (_, py) in bt.pools.values())
    gcur = l.dgrad(gcur, mask=None if prev_pooled else src)
# ---- isolate block4_conv3's dgrad on the ACTUAL gradient
import torch.nn.functional as F
l = layers["block4_conv3"]
xx, yy = bt.pools["block4_conv3"]
gy = pre["block4_conv3"].grad                       # oracle's gradient (matches the product's to 1e-6)
wk = torch.tensor(w0["block4_conv3"][0], dtype=torch.float64)      # HWIO
z = pre["block4_conv2"].detach().clamp(min=0).permute(0, 3, 1, 2).requires_grad_(True)
out = F.conv2d(F.pad(z, (1, 1, 1, 1)), wk.permute(3, 2, 0, 1))
out.backward(gy.permute(0, 3, 1, 2))
want_dx = z.grad.permute(0, 2, 3, 1)                 # gradient w.r.t. block4_conv2's ReLU OUTPUT (no mask)
got_dx = l.dgrad(gy.float().cuda().contiguous(), mask=None).cpu().double()
print("dgrad vs conv_transpose relfro %.2e" % float((got_dx - want_dx).norm() / want_dx.norm()))
msk = (pre["block4_conv2"].detach() > 0)
print("oracle chain check relfro %.2e" % float((want_dx * msk - pre["block4_conv2"].grad).norm() / pre["block4_conv2"].grad.norm()))
print("product mask vs oracle mask mismatches", int(((l.x.cpu() > 0) != msk).sum()))
