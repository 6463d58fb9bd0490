#!/usr/bin/env python3
"""configs[3] (ResNet-101, 600x1500, bf16) piece by piece at batch B: stem + pool + cast, stages 2 / 3 / 4, the RPN head, and the
detector head over B x 300 RoIs, each captured into its own hipGraph and replayed alone on the chip.  Dev tool: where a batched
pipeline's time would go.   usage: c4_stage_times.py [B ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from faster_rcnn_amd import nets, ops, resnet
from faster_rcnn_amd.weights import resnet_block_names, synthetic_resnet

H, W, A, C = 600, 1500, 18, 10


def graph_time(f, throughput, reps=20):
    ws = ops.ConvWorkspace()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side), ops.conv_workspace(ws), ops.tile_policy(throughput):
        f(); f()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="thread_local"), ops.conv_workspace(ws), ops.tile_policy(throughput):
        out = f()
    for _ in range(3):
        g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        for _ in range(reps):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps)
    return best, out


def main():
    batches = [int(v) for v in sys.argv[1:]] or [1, 4, 8]
    w = synthetic_resnet(101, anchors_per_loc=A, num_classes=C, seed=1)
    base = resnet.resnet101_base(weights=w, dtype="bf16")
    rpn = resnet.resnet101_rpn(base, include_conv=True, anchors_per_loc=A)
    det = resnet.resnet101_classifier(300, C, weights=w, dtype="bf16")
    net, head = base.net, det.head
    stages = {}
    for (stage, block, _), units in zip(resnet_block_names(101), net.blocks):
        stages.setdefault(stage, []).append(units)
    rs = np.random.RandomState(0)
    for B in batches:
        x = torch.from_numpy((rs.rand(B, H, W, 3) * 255 - 110).astype(np.float32)).cuda()
        for thr in (False, True):
            tag = "B=%d %s" % (B, "shared" if thr else "alone ")
            t, s0 = graph_time(lambda: ops.cast_bf16(ops.pool2d(net.stem(x), 3, 2, True)), thr)
            line = ["stem+pool+cast %.3f" % t]
            cur, total = s0, t
            for st in (2, 3, 4):
                def run(cur=cur, st=st):
                    y = cur
                    for u in stages[st]:
                        y = nets.run_block(u, y)
                    return y
                t, cur = graph_time(run, thr)
                total += t
                line.append("stage%d %.3f" % (st, t))
            feat = cur
            t, _ = graph_time(lambda: rpn.head(feat), thr)
            total += t
            line.append("rpn %.3f" % t)
            print("%s trunk: %s | total %.3f ms = %.3f ms/img" % (tag, "  ".join(line), total, total / B), flush=True)
        # detector head over B x 300 RoIs (all RoIs against image 0's map: the GEMM shapes are what matters here)
        rois = torch.from_numpy(np.stack([rs.randint(0, 60, B * 300), rs.randint(0, 20, B * 300), rs.randint(61, 94, B * 300), rs.randint(21, 38, B * 300)], 1).astype(np.float32)).cuda()
        f1 = feat[:1].contiguous()
        for thr, env in ((False, ""), (True, ""), (True, "wide")):
            if env:
                os.environ["FRCNN_BF16_WIDE"] = "1"
            t, _ = graph_time(lambda: head(f1, rois), thr)
            print("B=%d head over %d RoIs (%s%s): %.3f ms = %.3f ms/img" % (B, B * 300, "shared" if thr else "alone", " wide" if env else "", t, t / B), flush=True)


if __name__ == "__main__":
    main()
