#!/usr/bin/env python3
"""bench.train_loop_leg several times in ONE process (dev tool): how the timed loop depends on the process's age."""
import gc
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import bench
from faster_rcnn_amd import train

kind = sys.argv[1] if len(sys.argv) > 1 else "rpn_step1"
dtype = sys.argv[2] if len(sys.argv) > 2 else "f32"
mode = sys.argv[3] if len(sys.argv) > 3 else "plain"
for rep in range(7):
    if mode == "collect":
        gc.collect(); torch.cuda.empty_cache()
    t0 = time.perf_counter()
    r = bench.train_loop_leg(kind, dtype, iterations=40, warm=64, fast=True)
    print(kind, dtype, mode, "leg", rep, "->", r["ms_per_iteration"], "ms;  reserved %.2f GB allocated %.2f GB, live drivers %d, gc counts %s, leg %.1f s" % (
        torch.cuda.memory_reserved() / 2 ** 30, torch.cuda.memory_allocated() / 2 ** 30, len(train._LIVE_DRIVERS), gc.get_count(), time.perf_counter() - t0), flush=True)
