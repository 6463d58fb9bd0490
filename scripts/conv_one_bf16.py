#!/usr/bin/env python3
"""One bf16 1x1 conv shape (a GEMM over M rows) under several tile codes, checked against each other and timed from a
hipGraph of back-to-back launches.  Dev tool.   usage: conv_one_bf16.py M N K residual(0/1) tile[,tile...] [reps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from faster_rcnn_amd import ops

M, N, K, res = (int(v) for v in sys.argv[1:5])
tiles = [int(v) for v in sys.argv[5].split(",")]
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 10
rs = np.random.RandomState(0)
x = torch.from_numpy(rs.randn(1, M, 1, K).astype(np.float32)).cuda().to(torch.bfloat16)
w = (rs.randn(1, 1, K, N) * np.sqrt(2.0 / K)).astype(np.float32)
pc = ops.PackedConvBf16(w, (rs.rand(N) + 0.5).astype(np.float32), rs.randn(N).astype(np.float32))
r = torch.from_numpy(rs.randn(1, M, 1, N).astype(np.float32)).cuda().to(torch.bfloat16) if res else None
ref = None
for tile in tiles:
    with ops.conv_workspace(ops.NO_SPLIT_K):
        y = ops.conv2d_bf16(x, pc, 1, "valid", "relu", r, tile=tile)
        torch.cuda.synchronize()
        same = "" if ref is None else ("  == first" if torch.equal(y, ref) else "  DIFFERS from first (max %.4g)" % float((y.float() - ref.float()).abs().max()))
        ref = y if ref is None else ref
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(reps):
                ops.conv2d_bf16(x, pc, 1, "valid", "relu", r, tile=tile)
        g.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = 1e9
        for _ in range(3):
            e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / reps)
    byts = (M * K + N * K + M * N * (2 if res else 1)) * 2
    print("M=%d N=%d K=%d res=%d tile=%-3d %8.1f us %7.1f TFLOP/s %6.2f TB/s%s" % (M, N, K, res, tile, best * 1e3, 2.0 * M * N * K / best / 1e9, byts / best / 1e9, same), flush=True)
