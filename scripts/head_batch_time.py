#!/usr/bin/env python3
"""VERDICT r3 item 4, measured before building: (b) the fp32 detector head over B x 300 RoIs in ONE pass (what an fp32
BatchedInferencePipeline would launch; the crops' source image does not change a GEMM's time, so one feature map serves)
against B head passes of 300 RoIs on B streams; (a) the GEMM a Winograd F(2x2,3x3) head layer would run -- 16 positions x
(300 RoIs x 16 tiles) rows, 512 -> 512 -- on the existing 1x1 kernel, beside the direct 3x3 launch it would replace."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np
import torch

import bench
from faster_rcnn_amd import ops

pipe, w, anchors = bench.build_pipeline(calibrate=False)
x = torch.from_numpy(bench.synth_image(100)).cuda()
out = pipe.forward_dev(x)
torch.cuda.synchronize()
feat, rois300 = out["feat"], out["rois"]
head = pipe.det.head


def graph_of(fn, throughput):
    ws = ops.NO_SPLIT_K if throughput else ops.ConvWorkspace()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side), ops.conv_workspace(ws), ops.tile_policy(throughput):
        fn(); fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="thread_local"), ops.conv_workspace(ws), ops.tile_policy(throughput):
        keep = fn()
    return g, keep, ws


def wall(step, n=20):
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


print("== (b) fp32 head: one pass over B x 300 RoIs vs B passes on B streams (ms per image)")
for B in (1, 2, 4, 8):
    rois = rois300.repeat(B, 1).contiguous()
    g, keep, ws = graph_of(lambda: head(feat, rois), B > 1)
    one = wall(lambda: g.replay()) / B
    streams = [torch.cuda.Stream() for _ in range(B)]
    graphs = [graph_of(lambda: head(feat, rois300), B > 1) for _ in range(B)]

    def step():
        for (gg, _, _), st in zip(graphs, streams):
            with torch.cuda.stream(st):
                gg.replay()
    many = wall(step) / B
    print("B=%d: one pass %.3f ms/img   %d streams %.3f ms/img" % (B, one, B, many))

print("== (a) Winograd F(2x2,3x3) GEMM stand-in vs the direct 3x3 head launch")
rs = np.random.RandomState(0)


def conv_us(n, h, wd, cin, cout, k, padding, layout, tile=0, iters=20):
    xx = torch.from_numpy(rs.randn(*((h, wd, n, cin) if layout else (n, h, wd, cin))).astype(np.float32)).cuda()
    wt = (rs.randn(k, k, cin, cout) * np.sqrt(2.0 / (k * k * cin))).astype(np.float32)
    pc = ops.PackedConv(wt, np.ones(cout, np.float32), np.zeros(cout, np.float32))
    g, keep, ws = graph_of(lambda: [ops.conv2d(xx, pc, 1, padding, "relu", tile=tile, layout=layout) for _ in range(5)], False)
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 5 * 1e3


direct = conv_us(300, 7, 7, 512, 512, 3, "same", 1)
print("direct 3x3 512->512 on 300 RoIs (position-major, tap skipping): %.1f us (%.1f TF/s nominal 69.4 GFLOP)" % (direct, 69.363e3 / direct))
for rows in (4800 * 16,):
    gemm = conv_us(1, 1, rows, 512, 512, 1, "valid", 0)
    print("1x1 512->512 on %d rows (16 Winograd positions x 4800 tiles; one filter stands in for the 16): %.1f us (%.1f TF/s on %.1f GFLOP)"
          % (rows, gemm, 2.0 * rows * 512 * 512 / gemm / 1e6, 2.0 * rows * 512 * 512 / 1e9))
# the transforms are streaming passes: input d (30.1 MB) -> V (16 x 4800 x 512 x 4 B = 157 MB); M (157 MB) -> y (30.1 MB)
a = torch.empty(157286400 // 4, dtype=torch.float32, device="cuda")
b = torch.empty(30105600 // 4, dtype=torch.float32, device="cuda")
torch.cuda.synchronize()
for name, src, dst in (("input transform stand-in: read 30 MB, write 157 MB", b, a), ("output transform stand-in: read 157 MB, write 30 MB", a, b)):
    reps = -(-dst.numel() // src.numel())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn = (lambda: dst.copy_(src.repeat(reps)[:dst.numel()])) if dst.numel() > src.numel() else (lambda: dst.copy_(src[:dst.numel()]))
    fn(); torch.cuda.synchronize()
    e0.record()
    for _ in range(10):
        fn()
    e1.record(); torch.cuda.synchronize()
    print("%s: %.1f us (torch copy kernels; a fused transform moves the same bytes)" % (name, e0.elapsed_time(e1) / 10 * 1e3))
