#!/usr/bin/env python3
"""Numerical what-if (CPU, oracle graphs): ResNet-50 RPN + detector logits when every convolution multiplies
split-bf16 operands (x = hi + lo, w = hi + lo; hi*hi + hi*lo + lo*hi with f32 accumulation) instead of fp32,
against the f64 graph.  Error metric of the parity tests: max |a - b| / max(|b|, 1)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from faster_rcnn_amd.weights import synthetic_resnet
from oracle import keras_ref
from oracle.keras_ref import KerasGraphs


def split2(x):
    hi = x.to(torch.bfloat16).float()
    lo = (x - hi).to(torch.bfloat16).float()
    return hi, lo


class SplitGraphs(KerasGraphs):
    mode = "x3"

    def conv(self, x, name, stride=1, padding="valid"):
        w = self.w[name]
        k = torch.as_tensor(np.asarray(w[0]), dtype=torch.float32)
        b = w[1] if len(w) > 1 else None
        xh, xl = split2(torch.as_tensor(np.asarray(x) if not isinstance(x, torch.Tensor) else x).float())
        kh, kl = split2(k)
        if self.mode == "bf16":
            return keras_ref.conv2d(xh, kh, b, stride, padding, dtype=torch.float32)
        y = keras_ref.conv2d(xh, kh, b, stride, padding, dtype=torch.float32)
        y = y + keras_ref.conv2d(xh, kl, None, stride, padding, dtype=torch.float32)
        y = y + keras_ref.conv2d(xl, kh, None, stride, padding, dtype=torch.float32)
        return y

    def _dense(self, x, name):                              # the dense heads stay fp32
        return KerasGraphs._dense(self, x, name)


def main():
    H, W = (int(a) for a in (sys.argv[1:3] if len(sys.argv) > 2 else (240, 352)))
    w = synthetic_resnet(50, anchors_per_loc=9, num_classes=21, seed=5)
    rs = np.random.RandomState(3)
    x = (rs.randint(0, 256, (1, H, W, 3)).astype(np.float32) - np.array([103.939, 116.779, 123.68], np.float32))
    fh, fw = -(-H // 16), -(-W // 16)
    rois = np.array([[0, 0, fw - 1, fh - 1], [2, 3, 9, 12], [5, 1, 14, 8], [1, 1, 3, 2]] * 16, np.float32)[:64]
    err = lambda a, b: float(((a.double() - b).abs() / b.abs().clamp(min=1)).max())
    ref = KerasGraphs(w, torch.float64)
    f64 = ref.resnet_base(x, 50)
    c64, r64 = ref.rpn(f64)
    oc64, or64 = ref.resnet_classifier(f64, rois, 21, 50)
    for tag, g in (("fp32", KerasGraphs(w, torch.float32)), ("split-bf16 x3", SplitGraphs(w, torch.float32))):
        f = g.resnet_base(x, 50)
        c, r = g.rpn(f)
        oc, orr = g.resnet_classifier(f, rois, 21, 50)
        print("%-14s conv4 %.2e  rpn cls %.2e reg %.2e  det cls %.2e reg %.2e" % (tag, err(f, f64), err(c, c64), err(r, r64), err(oc, oc64), err(orr, or64)))


if __name__ == "__main__":
    main()
