#!/usr/bin/env python3
"""Which of the training streams share a hardware queue? (dev tool)  Runs a few bare RPN steps first (bench_train.py's order of stream
creation), then a loop leg, then prints for every pair (A, B): does an event on B complete while A spins?"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

import bench
from faster_rcnn_amd import feed, resnet, train, util
from faster_rcnn_amd.weights import synthetic_resnet


def beside(a, b):
    """does an event on b complete while a spins?"""
    e = torch.cuda.Event(); e.record(b); e.synchronize()
    with torch.cuda.stream(a):
        torch.cuda._sleep(4_000_000)
        busy = torch.cuda.Event(); busy.record(a)
    ev = torch.cuda.Event(); ev.record(b); ev.synchronize()
    r = not busy.query()
    busy.synchronize()
    return int(r)


order = sys.argv[1] if len(sys.argv) > 1 else "bare-first"
if order == "bare-first":
    anchors = util.get_anchors([128, 256, 512])
    A = len(anchors)
    rpn = resnet.resnet50_rpn(resnet.resnet50_base(weights=synthetic_resnet(50, anchors_per_loc=A, num_classes=21, seed=1),
                                                   weight_regularizer=resnet.WEIGHT_REGULARIZER, bias_regularizer=resnet.BIAS_REGULARIZER), anchors_per_loc=A)
    rpn.compile(optimizer=train.optimizer_from_str("sgd"), loss=None)
    rows, cols = resnet.get_conv_rows_cols(600, 1000)
    rs = np.random.RandomState(0)
    x = rs.randn(1, 600, 1000, 3)
    yc = rs.rand(1, rows, cols, 2 * A) < 0.1
    yb = rs.randn(1, rows, cols, 8 * A).astype(np.float32)
    for _ in range(10):
        rpn.train_on_batch(x, [yc, yb])
    torch.cuda.synchronize()
r = bench.train_loop_leg("rpn_step1", "f32", iterations=40, warm=64, fast=True)
print(order, "loop leg:", r["ms_per_iteration"], "ms per iteration")
named = {"main": torch.cuda.current_stream(), "prefix": train._PREFIX_STREAM, "wgrad": train._WGRAD_STREAM, "loss": train._LOSS_STREAM, "manager": feed._MANAGER_STREAM}
named = {k: v for k, v in named.items() if v is not None}
print("tried", len(getattr(feed.manager_stream, "tried", [])), "candidates")
print("%-8s" % "", " ".join("%-8s" % k for k in named))
for ka, a in named.items():
    print("%-8s" % ka, " ".join("%-8s" % ("-" if a is b else beside(a, b)) for b in named.values()))
