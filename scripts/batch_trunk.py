#!/usr/bin/env python3
"""Trunk + RPN head on a batch of images in ONE pass against one image per stream.  Dev experiment."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from faster_rcnn_amd import ops, resnet, util
from faster_rcnn_amd.weights import synthetic_resnet


def graph_of(f, throughput):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    ws = ops.ConvWorkspace()
    with torch.cuda.stream(side), ops.conv_workspace(ws), ops.tile_policy(throughput):
        f(); f()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="thread_local"), ops.conv_workspace(ws), ops.tile_policy(throughput):
        out = f()
    return g, out, ws


def timed(step, n=30):
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        for _ in range(n):
            step()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n)
    return best


def main():
    c4 = "--c4" in sys.argv                      # configs[3]: ResNet-101, 600x1500, 18 anchors, bf16
    H, W = (600, 1500) if c4 else (600, 1000)
    if c4:
        w = synthetic_resnet(101, anchors_per_loc=18, num_classes=10, seed=1)
        base = resnet.resnet101_base(weights=w, dtype="bf16")
        rpn = resnet.resnet101_rpn(base, include_conv=True, anchors_per_loc=18)
    else:
        w = synthetic_resnet(50, anchors_per_loc=9, num_classes=21, seed=1)
        base = resnet.resnet50_base(weights=w)
        rpn = resnet.resnet50_rpn(base, include_conv=True, anchors_per_loc=9)
    rs = np.random.RandomState(0)
    for B in (1, 2, 4, 8):
        x = torch.from_numpy(rs.randn(B, H, W, 3).astype(np.float32) * 50).cuda()
        for thr in (False, True):
            g, out, ws = graph_of(lambda: rpn.forward_dev(x), thr)
            ms = timed(g.replay)
            print("batch %d in one pass (tile policy %s): %.3f ms = %.3f ms per image" % (B, "shared" if thr else "alone", ms, ms / B))
    x1 = [torch.from_numpy(rs.randn(1, H, W, 3).astype(np.float32) * 50).cuda() for _ in range(4)]
    for thr in (False, True):
        gs = [graph_of(lambda xx=xx: rpn.forward_dev(xx), thr) for xx in x1]
        streams = [torch.cuda.Stream() for _ in range(4)]

        def step():
            for (g, _, _), st in zip(gs, streams):
                with torch.cuda.stream(st):
                    g.replay()
        for st in streams:
            st.wait_stream(torch.cuda.current_stream())
        for _ in range(5):
            step()
        torch.cuda.synchronize()
        import time
        t0 = time.perf_counter()
        for _ in range(30):
            step()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 30 * 1e3
        print("4 streams x 1 image (tile policy %s): %.3f ms per step = %.3f ms per image" % ("shared" if thr else "alone", ms, ms / 4))


if __name__ == "__main__":
    main()
