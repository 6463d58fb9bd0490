#!/usr/bin/env python3
"""Network-level parity on random image sizes (odd / even, tiny, aspect ratios the fixed tests do not use): the
lowered ResNet-50 / VGG16 RPN graphs on the GPU against the f64 oracle graph, 1e-4 bar.  Dev tool."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from faster_rcnn_amd import resnet, vgg
from faster_rcnn_amd.weights import synthetic_resnet, synthetic_vgg16
from oracle.keras_ref import KerasGraphs


def rel_err(got, want):
    got = torch.as_tensor(np.asarray(got)).double()
    return ((got - want.double()).abs() / want.double().abs().clamp(min=1.0)).max().item()


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    wr = synthetic_resnet(50, anchors_per_loc=9, num_classes=21, seed=3)
    rbase = resnet.resnet50_base(weights=wr)
    rrpn = resnet.resnet50_rpn(rbase, include_conv=True, anchors_per_loc=9)
    wv = synthetic_vgg16(seed=4)
    vbase = vgg.vgg16_base(weights=wv)
    vrpn = vgg.vgg16_rpn(vbase, include_conv=True, anchors_per_loc=9)
    gr, gv = KerasGraphs(wr, torch.float64), KerasGraphs(wv, torch.float64)
    fails = 0
    for it in range(n_cases):
        h, w = int(rs.randint(33, 260)), int(rs.randint(33, 330))
        x = (rs.randint(0, 256, (1, h, w, 3)).astype(np.float64) - np.array([103.939, 116.779, 123.68]))
        cls, reg, feat = rrpn.predict_on_batch(x)
        f64 = gr.resnet_base(x, 50)
        c64, r64 = gr.rpn(f64)
        e = (rel_err(feat, f64), rel_err(cls, c64), rel_err(reg, r64))
        ok = tuple(feat.shape) == tuple(f64.shape) and max(e) < 1e-4
        print("resnet50 %dx%d -> %s  feat %.2e cls %.2e reg %.2e %s" % (h, w, tuple(feat.shape[1:3]), *e, "" if ok else "FAIL"))
        fails += not ok
        cls, reg, feat = vrpn.predict_on_batch(x)
        f64 = gv.vgg_base(x)
        c64, r64 = gv.rpn(f64)
        e = (rel_err(feat, f64), rel_err(cls, c64), rel_err(reg, r64))
        ok = tuple(feat.shape) == tuple(f64.shape) and max(e) < 1e-4
        print("vgg16    %dx%d -> %s  feat %.2e cls %.2e reg %.2e %s" % (h, w, tuple(feat.shape[1:3]), *e, "" if ok else "FAIL"))
        fails += not ok
    print("cases %d  failures %d" % (n_cases, fails))
    sys.exit(1 if fails else 0)


if __name__ == "__main__":
    main()
