#!/usr/bin/env python3
"""Weight-gradient launches of the training steps, one layer at a time alone on the chip (HIP events, 20 reps each):
us and TFLOP/s per layer.  FRCNN_WGRAD_BIG=0 selects the 64x64-tile kernel everywhere.  Dev tool."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from faster_rcnn_amd import ops

SHAPES = [  # name, n, h, w, cin, cout, k, stride, padding
    ("s4_2a", 1, 38, 63, 1024, 256, 1, 1, "valid"), ("s4_2b", 1, 38, 63, 256, 256, 3, 1, "same"), ("s4_2c", 1, 38, 63, 256, 1024, 1, 1, "valid"),
    ("s4a_2a", 1, 75, 125, 512, 256, 1, 2, "valid"), ("s4a_1", 1, 75, 125, 512, 1024, 1, 2, "valid"),
    ("rpn_conv1", 1, 38, 63, 1024, 512, 3, 1, "same"),
    ("t5a_2a", 64, 14, 14, 1024, 512, 1, 2, "valid"), ("t5a_1", 64, 14, 14, 1024, 2048, 1, 2, "valid"),
    ("t5_2b", 64, 7, 7, 512, 512, 3, 1, "same"), ("t5_2c", 64, 7, 7, 512, 2048, 1, 1, "valid"), ("t5x_2a", 64, 7, 7, 2048, 512, 1, 1, "valid"),
    ("vgg3", 1, 150, 250, 256, 256, 3, 1, "same"), ("vgg4", 1, 75, 125, 512, 512, 3, 1, "same"),
]
bf16 = "--bf16" in sys.argv
tot = 0.0
for name, n, h, w, cin, cout, k, stride, padding in SHAPES:
    ho = -(-h // stride) if padding == "same" else (h - k) // stride + 1
    wo = -(-w // stride) if padding == "same" else (w - k) // stride + 1
    x = torch.randn(n, h, w, cin, device="cuda")
    g = torch.randn(n, ho, wo, cout, device="cuda")
    if bf16:
        x, g = x.bfloat16(), g.bfloat16()
    fn = ops.conv2d_wgrad_bf16 if bf16 else ops.conv2d_wgrad
    for _ in range(3):
        fn(x, g, k, k, stride, padding, want_bias=False)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn(x, g, k, k, stride, padding, want_bias=False)
    e1.record(); e1.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    gf = 2.0 * n * ho * wo * cin * cout * k * k / 1e9
    tot += us
    print("%-10s M=%6d %4d->%4d k%d  %8.1f us  %6.1f TF/s" % (name, n * ho * wo, cin, cout, k, us, gf / us * 1e-3 * 1e3 / 1e3 * 1e3 / 1e3 if False else gf / (us * 1e-6) / 1e3))
print("total %.1f us" % tot)
