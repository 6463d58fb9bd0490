#!/usr/bin/env python3
"""Time of the proposal stage (decode -> top-K -> NMS) at training settings (12000 -> 2000) and test settings
(8000 -> 300) on synthetic C2 RPN outputs.  Dev tool."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from faster_rcnn_amd import ops, util
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import synth

def t(f, n=10):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3

for tag in ("c2", "c4"):
    rows, cols, A = synth.SHAPES[tag]
    reg, cls = synth.rpn_outputs(tag)
    anchors = util.get_anchors([128, 256, 512] if A == 9 else [16, 32, 64, 128, 256, 512])
    regd, clsd = torch.from_numpy(reg).cuda(), torch.from_numpy(cls).cuda()
    rois_all, valid = ops.decode_proposals(regd, np.asarray(anchors) // 16)
    scores = clsd.reshape(-1)
    for pre, post in ((8000, 300), (12000, 2000)):
        order, n = ops.topk_order(scores, valid, pre)
        cand, cs = ops.gather_candidates(rois_all, scores, order, n, pre)
        keep, nk = ops.nms_sorted(cand, n, 0.7, post)
        print("%s pre=%d post=%d: kept %d; topk %.0f us, nms (mask+scan) %.0f us" % (
            tag, pre, post, int(nk.item()), t(lambda: ops.topk_order(scores, valid, pre)), t(lambda: ops.nms_sorted(cand, n, 0.7, post))))
