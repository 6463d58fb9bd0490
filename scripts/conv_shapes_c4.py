#!/usr/bin/env python3
"""Per-shape timing of the bf16 conv engine on the BATCHED shapes of configs[3] (ResNet-101 600x1500, eight images per pass, 2 400 RoI
rows x 49 positions): every distinct launch under each tile code, round-robin, each alone on the chip.  Dev tool.
usage: conv_shapes_c4.py [tile,tile,...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from faster_rcnn_amd import ops

B = 8
SHAPES = [  # name, count, (n, h, w), cin, cout, k, stride, padding, layout, residual
    ("s2a_2a", 1, (B, 149, 374), 64, 64, 1, 1, "valid", 0, 0), ("s2_2b", 3, (B, 149, 374), 64, 64, 3, 1, "same", 0, 0),
    ("s2_2c", 4, (B, 149, 374), 64, 256, 1, 1, "valid", 0, 1), ("s2x_2a", 2, (B, 149, 374), 256, 64, 1, 1, "valid", 0, 0),
    ("s3a_2a", 1, (B, 149, 374), 256, 128, 1, 2, "valid", 0, 0), ("s3a_1", 1, (B, 149, 374), 256, 512, 1, 2, "valid", 0, 0),
    ("s3_2b", 4, (B, 75, 187), 128, 128, 3, 1, "same", 0, 0), ("s3_2c", 4, (B, 75, 187), 128, 512, 1, 1, "valid", 0, 1),
    ("s3x_2a", 3, (B, 75, 187), 512, 128, 1, 1, "valid", 0, 0),
    ("s4a_2a", 1, (B, 75, 187), 512, 256, 1, 2, "valid", 0, 0), ("s4a_1", 1, (B, 75, 187), 512, 1024, 1, 2, "valid", 0, 0),
    ("s4_2b", 23, (B, 38, 94), 256, 256, 3, 1, "same", 0, 0), ("s4_2c", 23, (B, 38, 94), 256, 1024, 1, 1, "valid", 0, 1),
    ("s4x_2a", 22, (B, 38, 94), 1024, 256, 1, 1, "valid", 0, 0), ("rpn_conv1", 1, (B, 38, 94), 1024, 512, 3, 1, "same", 0, 0),
    ("s5a_map", 1, (B, 38, 94), 1024, 2048, 1, 1, "valid", 0, 0),
    ("s5_2b", 3, (300 * B, 7, 7), 512, 512, 3, 1, "same", 1, 0), ("s5_2c", 3, (300 * B, 7, 7), 512, 2048, 1, 1, "valid", 1, 1),
    ("s5x_2a", 2, (300 * B, 7, 7), 2048, 512, 1, 1, "valid", 1, 0),
]


def main():
    argv = [a for a in sys.argv[1:] if not a.startswith("--")]
    tiles = [int(t) for t in (argv[0].split(",") if argv else "0,2,48,43,42,47,45,46".split(","))]
    g = torch.Generator(device="cuda").manual_seed(0)
    rs = np.random.RandomState(0)
    tot = {t: 0.0 for t in tiles}
    best_tot = 0.0
    print("%-9s %3s %7s %5s %5s | " % ("layer", "cnt", "M", "N", "K") + " ".join("%11s" % ("t%d:us/TF" % t) for t in tiles))
    for name, cnt, (n, h, w), cin, cout, k, stride, padding, layout, res in SHAPES:
        x = torch.randn(((h, w, n, cin) if layout else (n, h, w, cin)), device="cuda", generator=g).to(torch.bfloat16)
        wt = (rs.randn(k, k, cin, cout) * np.sqrt(2.0 / (k * k * cin))).astype(np.float32)
        pc = ops.PackedConvBf16(wt, np.ones(cout, np.float32), np.zeros(cout, np.float32))
        y = ops.conv2d_bf16(x, pc, stride, padding, "relu", layout=layout)
        r = torch.randn(y.shape, device="cuda", generator=g).to(torch.bfloat16) if res else None
        M = y.shape[0] * y.shape[1] * y.shape[2]
        fl = 2.0 * M * cout * k * k * cin
        del y
        timers = []
        for t in tiles:
            def run(t=t):
                return ops.conv2d_bf16(x, pc, stride, padding, "relu", residual=r, tile=t, layout=layout)
            try:
                for _ in range(2):
                    run()
                torch.cuda.synchronize()
                timers.append(run)
            except Exception:
                timers.append(None)
        best = [None] * len(tiles)
        for _ in range(3):
            for i, run in enumerate(timers):
                if run is None:
                    continue
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(8):
                    run()
                e1.record(); torch.cuda.synchronize()
                us = e0.elapsed_time(e1) / 8 * 1e3
                best[i] = us if best[i] is None else min(best[i], us)
        for t, us in zip(tiles, best):
            if us is not None:
                tot[t] += us * cnt
        best_tot += min(u for u in best if u is not None) * cnt
        print("%-9s %3d %7d %5d %5d | " % (name, cnt, M, cout, k * k * cin) + " ".join("     -     " if u is None else "%6.1f/%4.0f" % (u, fl / u / 1e6) for u in best), flush=True)
    print("per-code totals per pass of eight images (us):", {t: round(v) for t, v in tot.items()}, "best-of:", round(best_tot))


if __name__ == "__main__":
    main()
