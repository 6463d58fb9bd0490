#!/usr/bin/env python3
"""Where a DetectionEntry capture spends its ~50 ms (dev): warm-up pass, torch.cuda.graph enter (synchronize + gc + empty_cache), the
captured pass, graph instantiation at exit -- for a four-image pass and a one-image pass of a fresh geometry."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from faster_rcnn_amd import entry, resnet, util, ops
from faster_rcnn_amd.data.voc_data_helpers import VOC_CLASS_MAPPING
from faster_rcnn_amd.det_util import DetTrainingManager
from faster_rcnn_amd.weights import synthetic_resnet

anchors = util.get_anchors([128, 256, 512])
w = synthetic_resnet(50, anchors_per_loc=9, num_classes=21, seed=1)
rpn = resnet.resnet50_rpn(resnet.resnet50_base(weights=w), include_conv=True, anchors_per_loc=9)
det = resnet.resnet50_classifier(64, 21, weights=w)
mgr = DetTrainingManager(rpn_model=rpn, class_mapping=VOC_CLASS_MAPPING, preprocess_func=resnet.preprocess, anchor_dims=anchors)
eng = entry.DetectionEntry(mgr, det, 64, 16, in_flight=4)
orig_graph = torch.cuda.graph
phases = {}


class timed_graph(orig_graph):
    def __enter__(self):
        torch.cuda.synchronize()
        t = time.perf_counter()
        r = super().__enter__()
        phases["enter"] = time.perf_counter() - t
        self._t_body = time.perf_counter()
        return r

    def __exit__(self, *a):
        phases["body"] = time.perf_counter() - self._t_body
        t = time.perf_counter()
        r = super().__exit__(*a)
        phases["exit"] = time.perf_counter() - t
        return r


torch.cuda.graph = timed_graph
for (H, W, B) in [(600, 800, 4), (600, 904, 4), (600, 898, 1), (608, 800, 4), (600, 802, 1)]:
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    s = eng._capture(H, W, (375, 500), False, B)
    torch.cuda.synchronize()
    tot = time.perf_counter() - t0
    print("capture %dx%d B=%d: total %.1f ms; enter %.1f, body %.1f, exit %.1f, rest (alloc + warm-up %d pass) %.1f ms; reserved %.1f GB" % (
        H, W, B, 1e3 * tot, 1e3 * phases["enter"], 1e3 * phases["body"], 1e3 * phases["exit"], entry.WARMUP_PASSES,
        1e3 * (tot - phases["enter"] - phases["body"] - phases["exit"]), torch.cuda.memory_reserved() / 1e9))

import cProfile, pstats
pr = cProfile.Profile()
pr.enable()
eng._capture(600, 906, (375, 500), False, 4)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
