import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from tests.test_train_gpu import image
from faster_rcnn_amd import vgg, train
from faster_rcnn_amd.weights import synthetic_vgg16
from oracle import keras_train_ref as kt
rs = np.random.RandomState(8)
x = image(96, 128, seed=5)
n, C = 8, 21
rows, cols = 6, 8
x1 = rs.randint(0, cols - 2, n); y1 = rs.randint(0, rows - 2, n)
rois = np.stack([x1, y1, np.minimum(cols - 1, x1 + 1 + rs.randint(0, 5, n)), np.minimum(rows - 1, y1 + 1 + rs.randint(0, 4, n))], axis=1).astype(np.float32)[None]
ci = rs.randint(0, C, n)
yc = np.zeros((1, n, C), np.int32); yc[0, np.arange(n), ci] = 1
lab = np.zeros((n, 4 * (C - 1)), np.float32); tg = np.zeros((n, 4 * (C - 1)), np.float32)
for i, c in enumerate(ci):
    if c < C - 1:
        lab[i, 4 * c:4 * c + 4] = 1; tg[i, 4 * c:4 * c + 4] = rs.randn(4)
yb = np.concatenate([lab, tg], axis=1)[None]
w0 = synthetic_vgg16(seed=22)
old = {k: [np.array(a, dtype=np.float64) for a in v] for k, v in w0.items()}
base = vgg.vgg16_base(weights={k: [a.copy() for a in v] for k, v in w0.items()})
det = vgg.vgg16_classifier(n, C, base_model=base)
det.compile(train.SGD(1e-3, 0.9))
print(det.train_on_batch([x, rois], [yc, yb]))
ref_w, rl, _ = kt.det_train_step(w0, x, rois, yc, yb, C, kt.Optim("sgd", 1e-3), freeze_blocks=(1, 2), arch="vgg")
print(rl)
det._trainer.sync_weights()
for nme in kt.vgg_conv_names((3, 4, 5)) + ["fc1", "fc2", "dense_class_21", "dense_reg_21"]:
    for i, (o, g, w) in enumerate(zip(old[nme], det.weights[nme], ref_w[nme])):
        dg, dw = np.asarray(g, np.float64) - o, np.asarray(w, np.float64) - o
        err = np.maximum(np.abs(dg - dw) - 2 * 2.0 ** -23 * np.abs(o), 0)
        print("%-16s %d maxupd %.2e fro %.2e max %.2e" % (nme, i, np.abs(dw).max(), np.sqrt((err ** 2).sum() / (dw ** 2).sum()), err.max() / np.abs(dw).max()))
