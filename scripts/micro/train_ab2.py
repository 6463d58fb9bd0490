"""RPN / detector step times, f32 and mixed, interleaved rounds in one process (FRCNN_WGRAD_TARGET etc. from the environment)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from faster_rcnn_amd import resnet, train
from faster_rcnn_amd.weights import synthetic_resnet
H, W, A, C = 600, 1000, 9, 21
rs = np.random.RandomState(0)
x = (rs.randint(0, 256, (H, W, 3)).astype(np.float64) - np.array([103.939, 116.779, 123.68]))[None]
rows, cols = resnet.get_conv_rows_cols(H, W)
models = {}
for DT in ("f32", "bf16"):
    w = synthetic_resnet(50, anchors_per_loc=A, num_classes=C, seed=1)
    base = resnet.resnet50_base(weights=w, weight_regularizer=resnet.WEIGHT_REGULARIZER, bias_regularizer=resnet.BIAS_REGULARIZER, dtype=DT)
    rpn = resnet.resnet50_rpn(base, anchors_per_loc=A)
    can_use = rs.rand(1, rows, cols, A) < 0.012; is_pos = rs.rand(1, rows, cols, A) < 0.01
    yc = np.concatenate([can_use, is_pos], axis=3)
    yb = np.concatenate([np.repeat(can_use & is_pos, 4, axis=3).astype(np.float32), (rs.randn(1, rows, cols, 4 * A) * is_pos.repeat(4, axis=3)).astype(np.float32)], axis=3)
    rpn.compile(train.SGD(1e-3, 0.9))
    models[DT + " rpn"] = (lambda rpn=rpn, yc=yc, yb=yb: rpn.train_on_batch(x, [yc, yb]))
    dw = synthetic_resnet(50, anchors_per_loc=A, num_classes=C, seed=2)
    dbase = resnet.resnet50_base(weights=dw, weight_regularizer=resnet.WEIGHT_REGULARIZER, bias_regularizer=resnet.BIAS_REGULARIZER, dtype=DT)
    det = resnet.resnet50_classifier(64, C, dbase)
    n = 64
    x1 = rs.randint(0, cols - 8, n); y1 = rs.randint(0, rows - 8, n)
    rois = np.stack([x1, y1, x1 + 1 + rs.randint(0, 7, n), y1 + 1 + rs.randint(0, 7, n)], axis=1).astype(np.float32)[None]
    ci = rs.randint(0, C, n)
    ycd = np.zeros((1, n, C), np.float32); ycd[0, np.arange(n), ci] = 1
    ybd = np.zeros((1, n, 8 * (C - 1)), np.float32)
    det.compile(train.SGD(1e-3, 0.9))
    models[DT + " det"] = (lambda det=det, rois=rois, ycd=ycd, ybd=ybd: det.train_on_batch([x, rois], [ycd, ybd]))
for f in models.values():
    for _ in range(4):
        f()
res = {k: [] for k in models}
for rnd in range(4):
    for k, f in models.items():
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(15):
            f()
        torch.cuda.synchronize()
        res[k].append(1e3 * (time.perf_counter() - t0) / 15)
print(os.environ.get("TAG", ""), {k: "%.3f" % min(v) for k, v in res.items()}, flush=True)
