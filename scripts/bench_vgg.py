import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from faster_rcnn_amd import vgg, util, ops
from faster_rcnn_amd.weights import synthetic_vgg16
anchors = util.get_anchors([128, 256, 512])
w = synthetic_vgg16(anchors_per_loc=9, seed=1)
base = vgg.vgg16_base(weights=w)
rpn = vgg.vgg16_rpn(base, include_conv=True, anchors_per_loc=9)
rs = np.random.RandomState(0)
x = torch.from_numpy((rs.randint(0, 256, (1, 600, 1000, 3)).astype(np.float32) - 110.0)).cuda()
for _ in range(3): out = rpn.forward_dev(x)
torch.cuda.synchronize()
ops.CONV_PROFILE = []
rpn.forward_dev(x); torch.cuda.synchronize()
prof, ops.CONV_PROFILE = ops.CONV_PROFILE, None
t0 = time.perf_counter()
for _ in range(20): rpn.forward_dev(x)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 20
print("VGG16 RPN forward 600x1000: %.2f ms/img, %.1f img/s, %.1f GFLOP -> %.1f TF/s" % (dt * 1e3, 1 / dt, sum(p["flops"] for p in prof) / 1e9, sum(p["flops"] for p in prof) / dt / 1e12))
for p in prof:
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    p["relaunch"](); e0.record()
    for _ in range(5): p["relaunch"]()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 5 * 1e3
    print("  %-34s M=%7d N=%4d K=%5d  %7.1f us  %6.1f TF" % (p["kernel"], p["shape"][0], p["shape"][1], p["shape"][2], us, p["flops"] / us / 1e6))
