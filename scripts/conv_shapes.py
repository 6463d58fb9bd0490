#!/usr/bin/env python3
"""Per-shape timing of the conv engine on the GPU: every distinct conv launch of the C2
pipeline (ResNet-50 600x1000, 300 RoIs) under each tile configuration.  Dev tool."""
import os
import sys
import json

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from faster_rcnn_amd import ops

# (name, count, n, h, w, cin, cout, k, stride, padding)
def c2_shapes(rois=300):
    s = [("conv1", 1, 1, 600, 1000, 3, 64, 7, 2, "same")]
    def stage(tag, h, w, cin, f1, f3, nblocks, stride):
        ho, wo = (h - 1) // stride + 1, (w - 1) // stride + 1
        out = [(tag + "a_2a", 1, 1, h, w, cin, f1, 1, stride, "valid"), (tag + "a_1", 1, 1, h, w, cin, f3, 1, stride, "valid"),
               (tag + "_2b", nblocks, 1, ho, wo, f1, f1, 3, 1, "same"), (tag + "_2c", nblocks, 1, ho, wo, f1, f3, 1, 1, "valid"),
               (tag + "x_2a", nblocks - 1, 1, ho, wo, f3, f1, 1, 1, "valid")]
        return out, ho, wo
    o, h, w = stage("s2", 149, 249, 64, 64, 256, 3, 1); s += o
    o, h, w = stage("s3", h, w, 256, 128, 512, 4, 2); s += o
    o, h, w = stage("s4", h, w, 512, 256, 1024, 6, 2); s += o
    s += [("rpn_conv1", 1, 1, h, w, 1024, 512, 3, 1, "same"), ("rpn_cls", 1, 1, h, w, 512, 9, 1, 1, "valid"), ("rpn_reg", 1, 1, h, w, 512, 36, 1, 1, "valid")]
    s += [("s5a_2a", 1, rois, 7, 7, 1024, 512, 1, 1, "valid"), ("s5a_1", 1, rois, 7, 7, 1024, 2048, 1, 1, "valid"),
          ("s5_2b", 3, rois, 7, 7, 512, 512, 3, 1, "same"), ("s5_2c", 3, rois, 7, 7, 512, 2048, 1, 1, "valid"),
          ("s5x_2a", 2, rois, 7, 7, 2048, 512, 1, 1, "valid"), ("dense", 1, rois, 1, 1, 2048, 101, 1, 1, "valid")]
    return s


BF16 = "--bf16" in sys.argv


def time_conv(x, pc, stride, padding, tile, iters=10):
    """-> a callable that times `iters` back-to-back launches (us per launch), or None if the config refuses the shape."""
    if BF16:
        run = lambda: ops.conv2d_bf16(x, pc, stride, padding, "relu", tile=tile)
    else:
        run = lambda: ops.conv2d(x, pc, stride, padding, "relu", tile=tile)
    try:
        for _ in range(3):
            y = run()
    except Exception as e:
        return None
    if not BF16:
        run = lambda: ops.conv2d(x, pc, stride, padding, "relu", out=y, tile=tile)

    def timed():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            run()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters * 1e3   # us
    return timed


def time_configs(x, pc, stride, padding, tiles, rounds=4):
    """Round-robin over the configs and keep each one's best round: the clock ramps while a shape is being
    measured, so timing the configs one after the other favours whichever comes last."""
    timers = [time_conv(x, pc, stride, padding, t) for t in tiles]
    best = [None] * len(tiles)
    for _ in range(rounds):
        for i, tm in enumerate(timers):
            if tm is not None:
                us = tm()
                best[i] = us if best[i] is None else min(best[i], us)
    return best


def main():
    argv = [a for a in sys.argv[1:] if not a.startswith("--")]
    tiles = [int(t) for t in (argv[0].split(",") if argv else "0,1,2,3,4".split(","))]
    rs = np.random.RandomState(0)
    tot = {t: 0.0 for t in tiles}
    best_tot = 0.0
    tot_flops = 0.0
    print("%-10s %3s %7s %5s %6s %9s | " % ("layer", "cnt", "M", "N", "K", "GFLOP") + " ".join("t%d:us/TF" % t for t in tiles))
    for name, cnt, n, h, w, cin, cout, k, stride, padding in c2_shapes():
        if BF16 and cin % 64:
            continue
        x = torch.from_numpy(rs.randn(n, h, w, cin).astype(np.float32)).cuda()
        wt = (rs.randn(k, k, cin, cout) * np.sqrt(2.0 / (k * k * cin))).astype(np.float32)
        pc = (ops.PackedConvBf16 if BF16 else ops.PackedConv)(wt, np.ones(cout, np.float32), np.zeros(cout, np.float32))
        res = []
        if BF16:
            x = x.to(torch.bfloat16)
            y = ops.conv2d_bf16(x, pc, stride, padding)
        else:
            y = ops.conv2d(x, pc, stride, padding)
        M = y.shape[0] * y.shape[1] * y.shape[2]
        flops = 2.0 * M * cout * k * k * cin
        res = time_configs(x, pc, stride, padding, [t if not (cin % 32 and t in (1, 4)) else -1 for t in tiles])
        cells = []
        for t, us in zip(tiles, res):
            if us is None:
                cells.append("    -    ")
            else:
                tot[t] += us * cnt
                cells.append("%6.1f/%5.1f" % (us, flops / us / 1e6))
        valid = [u for u in res if u is not None]
        best_tot += min(valid) * cnt
        tot_flops += flops * cnt
        print("%-10s %3d %7d %5d %6d %9.3f | " % (name, cnt, M, cout, k * k * cin, flops / 1e9) + " ".join(cells))
    print("total GFLOP %.1f; per-tile totals (us):" % (tot_flops / 1e9), {t: round(v) for t, v in tot.items()}, "best-of:", round(best_tot),
          "-> %.1f TF/s" % (tot_flops / best_tot / 1e6))


if __name__ == "__main__":
    main()
