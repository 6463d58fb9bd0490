#!/usr/bin/env python3
"""Where an iteration of train_util.train_rpn / train_detector_step2 spends its HOST time with the device-resident feed: wall clock
of the manager's calls, the step's enqueue and the loss read-back, per iteration (dev tool; bench.train_loop_leg times the whole loop)."""
import contextlib
import io
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import bench
from faster_rcnn_amd import det_util, rpn_util, train_util

kind = sys.argv[1] if len(sys.argv) > 1 else "rpn_step1"
dtype = sys.argv[2] if len(sys.argv) > 2 else "f32"
acc = {}


def timed(owner, name, label):
    fn = getattr(owner, name)

    def wrapper(*a, **k):
        t = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            acc[label] = acc.get(label, 0.0) + time.perf_counter() - t
    setattr(owner, name, wrapper)


timed(rpn_util.RpnTrainingManager, "rpn_inputs_dev", "manager inputs (sample + pack)")
timed(rpn_util.RpnTrainingManager, "prefetch", "manager prefetch")
timed(det_util.DetTrainingManager, "get_training_input_dev", "manager inputs")
timed(det_util.DetTrainingManager, "prefetch", "manager prefetch")
timed(train_util, "_enqueue_step", "enqueue step")
timed(train_util._LossLog, "flush", "loss read-back (previous step)")
timed(rpn_util, "sample_range", "  of which sample_range")
iters = 64
bench.train_loop_leg(kind, dtype, iterations=8, warm=8, fast=True)         # allocator pools, kernel images, pinned areas: first-use costs
acc.clear()
res = bench.train_loop_leg(kind, dtype, iterations=iters, fast=True)
n = iters + 24
print(kind, dtype, res["ms_per_iteration"], "ms per iteration")
for k, v in acc.items():
    print("  %-36s %.3f ms per iteration" % (k, 1e3 * v / n))
