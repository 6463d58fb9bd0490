#!/usr/bin/env python3
"""Stem layers (3-channel input, conv_igemm.hip CIN3) against a 4-channel copy on the generic gather kernel.  Dev tool."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from faster_rcnn_amd import ops


def timed(run, iters=20):
    for _ in range(3):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(4):
        e0.record()
        for _ in range(iters):
            run()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters * 1e3)
    return best


def main():
    rs = np.random.RandomState(0)
    for name, h, w, k, stride in [("resnet conv1", 600, 1000, 7, 2), ("vgg block1_conv1", 600, 1000, 3, 1)]:
        x = torch.from_numpy(rs.randn(1, h, w, 3).astype(np.float32)).cuda()
        wt = (rs.randn(k, k, 3, 64) * 0.1).astype(np.float32)
        w4 = np.concatenate([wt, np.zeros((k, k, 1, 64), np.float32)], axis=2)
        pc3, pc4 = ops.PackedConv(wt, None, None), ops.PackedConv(w4, None, None)
        y = ops.conv2d(x, pc3, stride, "same", "relu")
        x4 = torch.cat([x, torch.zeros_like(x[..., :1])], dim=-1).contiguous()
        gf = 2.0 * y.numel() * k * k * 3 / 1e9
        t3 = timed(lambda: ops.conv2d(x, pc3, stride, "same", "relu", out=y))
        t4 = timed(lambda: ops.conv2d(x4, pc4, stride, "same", "relu", out=y))
        print("%-18s stem kernel %.1f us (%.1f TF)   generic on 4 channels %.1f us" % (name, t3, gf / t3 * 1e3, t4))


def fused_bf16():
    """The fused bf16 stem (conv1 + BN + ReLU + 3x3/2 max-pool + bf16 store, one launch) against the three launches it replaces."""
    rs = np.random.RandomState(1)
    wt = (rs.randn(7, 7, 3, 64) * 0.05).astype(np.float32)
    sc, sh = (rs.rand(64) + 0.5).astype(np.float32), rs.randn(64).astype(np.float32)
    ps, pc = ops.PackedStemBf16(wt, sc, sh), ops.PackedConv(wt, sc, sh)
    for n, h, w in ((1, 600, 1000), (1, 600, 1500), (8, 600, 1500)):
        x = torch.from_numpy(rs.randn(n, h, w, 3).astype(np.float32) * 50).cuda()
        t_f = timed(lambda: ops.stem_bf16(x, ps))
        t_o = timed(lambda: ops.cast_bf16(ops.pool2d(ops.conv2d(x, pc, 2, "same", "relu"), 3, 2, True)))
        print("stem %dx%dx%d: fused bf16 %.1f us (%.1f per image)   f32 conv + pool + cast %.1f us" % (n, h, w, t_f, t_f / n, t_o))


if __name__ == "__main__":
    main()
    fused_bf16()
