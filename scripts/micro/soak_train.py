#!/usr/bin/env python3
"""Soak: 400 detector + 400 RPN training steps (mixed precision, random images each step) -- device memory must stay
flat (the step driver allocates on three streams and hands tensors across them) and the losses finite.  Dev check."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from faster_rcnn_amd import resnet, train
from faster_rcnn_amd.weights import synthetic_resnet
H, W, A, C, n = 600, 1000, 9, 21, 64
rs = np.random.RandomState(0)
rows, cols = resnet.get_conv_rows_cols(H, W)
imgs = [(rs.randint(0, 256, (H, W, 3)).astype(np.float64) - np.array([103.939, 116.779, 123.68]))[None] for _ in range(6)]
for dt in ("bf16", "f32"):
    w = synthetic_resnet(50, anchors_per_loc=A, num_classes=C, seed=1)
    base = resnet.resnet50_base(weights=w, weight_regularizer=resnet.WEIGHT_REGULARIZER, bias_regularizer=resnet.BIAS_REGULARIZER, dtype=dt)
    rpn = resnet.resnet50_rpn(base, anchors_per_loc=A)
    rpn.compile(train.SGD(1e-4, 0.9))
    det = resnet.resnet50_classifier(n, C, resnet.resnet50_base(weights=synthetic_resnet(50, anchors_per_loc=A, num_classes=C, seed=2),
                                     weight_regularizer=resnet.WEIGHT_REGULARIZER, bias_regularizer=resnet.BIAS_REGULARIZER, dtype=dt))
    det.compile(train.SGD(1e-4, 0.9))
    mem = []
    for it in range(400):
        x = imgs[it % len(imgs)]
        can_use = rs.rand(1, rows, cols, A) < 0.012; is_pos = rs.rand(1, rows, cols, A) < 0.01
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
        if it % 100 == 99:
            torch.cuda.synchronize()
            mem.append((torch.cuda.memory_allocated() >> 20, torch.cuda.memory_reserved() >> 20))
    print(dt, "allocated / reserved MiB after 100, 200, 300, 400 steps:", mem, "last losses", [round(float(t), 4) for t in v])
    assert mem[-1][1] <= mem[0][1] * 1.25 + 64, mem
    del rpn, det, base
print("soak ok")
