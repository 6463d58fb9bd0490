#!/usr/bin/env python3
"""Step time of the two training configs at C2 shapes (ResNet-50, 600x1000, 9 anchors, 21 classes):
configs[2] RPN step 1 and configs[4] detector step 2 (64 RoIs).  One image per GPU per step; under
torchrun the flat gradient buffer is all-reduced over RCCL every step.  Dev/measurement tool."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from faster_rcnn_amd import dp, resnet, train, util
from faster_rcnn_amd.weights import synthetic_resnet

rank, world = dp.init_from_env()
if world == 1:
    torch.cuda.set_device(0)
H, W, A, C = 600, 1000, 9, 21
from faster_rcnn_amd import ops
if "--big-tiles" in sys.argv:
    ops.AUTO_TILE = 50
DT = "bf16" if "--bf16" in sys.argv else "f32"     # --bf16: mixed precision (bf16 activations / gradients / packed filters, f32 masters)
rs = np.random.RandomState(rank)
x = (rs.randint(0, 256, (H, W, 3)).astype(np.float64) - np.array([103.939, 116.779, 123.68]))[None]
rows, cols = resnet.get_conv_rows_cols(H, W)
steps, warm = 20, 3
out = {}
# ---- RPN step 1
w = synthetic_resnet(50, anchors_per_loc=A, num_classes=C, seed=1)
base = resnet.resnet50_base(weights=w, weight_regularizer=resnet.WEIGHT_REGULARIZER, bias_regularizer=resnet.BIAS_REGULARIZER, dtype=DT)
rpn = resnet.resnet50_rpn(base, anchors_per_loc=A)
can_use = rs.rand(1, rows, cols, A) < 0.012; is_pos = rs.rand(1, rows, cols, A) < 0.01
y_class = np.concatenate([can_use, is_pos], axis=3)
y_bbreg = np.concatenate([np.repeat(can_use & is_pos, 4, axis=3).astype(np.float32), (rs.randn(1, rows, cols, 4 * A) * is_pos.repeat(4, axis=3)).astype(np.float32)], axis=3)
rpn.compile(train.SGD(1e-3, 0.9))
for i in range(warm + steps):
    if i == warm:
        torch.cuda.synchronize(); t0 = time.perf_counter()
    rpn.train_on_batch(x, [y_class, y_bbreg])
torch.cuda.synchronize()
out["rpn_step1_ms"] = 1e3 * (time.perf_counter() - t0) / steps
out["rpn_step1_params_MB"] = rpn._trainer.params.total * 4 / 1e6
# ---- detector step 2
dw = synthetic_resnet(50, anchors_per_loc=A, num_classes=C, seed=2)
dbase = resnet.resnet50_base(weights=dw, weight_regularizer=resnet.WEIGHT_REGULARIZER, bias_regularizer=resnet.BIAS_REGULARIZER, dtype=DT)
det = resnet.resnet50_classifier(64, C, dbase)
n = 64
x1 = rs.randint(0, cols - 8, n); y1 = rs.randint(0, rows - 8, n)
rois = np.stack([x1, y1, x1 + 1 + rs.randint(0, 7, n), y1 + 1 + rs.randint(0, 7, n)], axis=1).astype(np.float32)[None]
ci = rs.randint(0, C, n)
yc = np.zeros((1, n, C), np.float32); yc[0, np.arange(n), ci] = 1
lab = np.zeros((n, 4 * (C - 1)), np.float32); tg = np.zeros((n, 4 * (C - 1)), np.float32)
for i, c in enumerate(ci):
    if c < C - 1:
        lab[i, 4 * c:4 * c + 4] = 1; tg[i, 4 * c:4 * c + 4] = rs.randn(4)
yb = np.concatenate([lab, tg], axis=1)[None]
det.compile(train.SGD(1e-3, 0.9))
for i in range(warm + steps):
    if i == warm:
        torch.cuda.synchronize(); t0 = time.perf_counter()
    det.train_on_batch([x, rois], [yc, yb])
torch.cuda.synchronize()
out["det_step2_ms"] = 1e3 * (time.perf_counter() - t0) / steps
out["det_step2_params_MB"] = det._trainer.params.total * 4 / 1e6
out["world"] = world
out["dtype"] = DT
out["rpn_step1_img_s"] = world * 1e3 / out["rpn_step1_ms"]
out["det_step2_img_s"] = world * 1e3 / out["det_step2_ms"]
if rank == 0:
    print(json.dumps(out))
