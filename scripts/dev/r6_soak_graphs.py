#!/usr/bin/env python3
"""Soak of the replayed training step (round 6): an RPN trainer and a detector trainer take turns over TWELVE image shapes with
train.STEP_GRAPH_SHAPES = 8 -- so captured steps are evicted and re-captured all the time (least recently used first out), eager and
replayed steps interleave, two trainers share the module's streams -- 720 steps each, mixed precision then fp32.  Device memory
must stay bounded (a captured step's private pool goes back when it is evicted) and every loss finite.  Dev check."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from faster_rcnn_amd import resnet, train
from faster_rcnn_amd.weights import synthetic_resnet
A, C, n = 9, 21, 16
rs = np.random.RandomState(0)
shapes = [(288 + 16 * (k % 4), 416 + 32 * (k // 4)) for k in range(12)]
train.STEP_GRAPH_SHAPES = 8
for dt in ("bf16", "f32"):
    w = synthetic_resnet(50, anchors_per_loc=A, num_classes=C, seed=1)
    reg = dict(weight_regularizer=resnet.WEIGHT_REGULARIZER, bias_regularizer=resnet.BIAS_REGULARIZER)
    rpn = resnet.resnet50_rpn(resnet.resnet50_base(weights=w, dtype=dt, **reg), anchors_per_loc=A)
    rpn.compile(train.SGD(1e-4, 0.9))
    det = resnet.resnet50_classifier(n, C, resnet.resnet50_base(weights=synthetic_resnet(50, anchors_per_loc=A, num_classes=C, seed=2), dtype=dt, **reg))
    det.compile(train.SGD(1e-4, 0.9))
    mem, captures = [], 0
    t0 = time.perf_counter()
    for it in range(720):
        H, W = shapes[(it * 5 + it // 36) % len(shapes)]            # walks all twelve shapes, each several times in a row now and then
        rows, cols = resnet.get_conv_rows_cols(H, W)
        x = (rs.randint(0, 256, (1, H, W, 3)).astype(np.float64) - np.array([103.939, 116.779, 123.68]))
        can_use = rs.rand(1, rows, cols, A) < 0.05; is_pos = rs.rand(1, rows, cols, A) < 0.03
        y_class = np.concatenate([can_use, is_pos], axis=3)
        y_bbreg = np.concatenate([np.repeat(can_use & is_pos, 4, axis=3).astype(np.float32), (rs.randn(1, rows, cols, 4 * A) * is_pos.repeat(4, axis=3)).astype(np.float32)], axis=3)
        l1 = rpn.train_on_batch(x, [y_class, y_bbreg], defer=bool(it & 1))
        x1 = rs.randint(0, cols - 8, n); y1 = rs.randint(0, rows - 8, n)
        rois = np.stack([x1, y1, x1 + 1 + rs.randint(0, 7, n), y1 + 1 + rs.randint(0, 7, n)], axis=1).astype(np.float32)[None]
        ci = rs.randint(0, C, n); yc = np.zeros((1, n, C), np.float32); yc[0, np.arange(n), ci] = 1
        yb = np.zeros((1, n, 8 * (C - 1)), np.float32)
        l2 = det.train_on_batch([x, rois], [yc, yb], defer=bool(it & 2))
        for l in (l1, l2):
            v = l.result() if hasattr(l, "result") else l
            assert all(np.isfinite(v)), (dt, it, v)
        if it % 120 == 119:
            torch.cuda.synchronize()
            mem.append((torch.cuda.memory_allocated() >> 20, torch.cuda.memory_reserved() >> 20))
    live = (len(rpn._trainer._graphs["graphs"]), len(det._trainer._graphs["graphs"]))
    print(dt, "%.1f s; allocated / reserved MiB every 120 steps:" % (time.perf_counter() - t0), mem, "captured steps alive:", live, "last losses", [round(float(t), 4) for t in v])
    assert max(live) <= 8
    assert mem[-1][1] <= mem[1][1] * 1.25 + 256, mem
    rpn._trainer.drop_step_graphs(); det._trainer.drop_step_graphs()
    del rpn, det
print("soak ok")
