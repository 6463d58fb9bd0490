"""A/B inside one process: the training step replayed from its hipGraph vs issued eagerly (rule: interleaved rounds, one device)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from faster_rcnn_amd import resnet, train
from faster_rcnn_amd.weights import synthetic_resnet
H, W, A, C = 600, 1000, 9, 21
rs = np.random.RandomState(0)
x = (rs.randint(0, 256, (H, W, 3)).astype(np.float64) - np.array([103.939, 116.779, 123.68]))[None]
rows, cols = resnet.get_conv_rows_cols(H, W)
for DT in ("f32", "bf16"):
    w = synthetic_resnet(50, anchors_per_loc=A, num_classes=C, seed=1)
    base = resnet.resnet50_base(weights=w, weight_regularizer=resnet.WEIGHT_REGULARIZER, bias_regularizer=resnet.BIAS_REGULARIZER, dtype=DT)
    rpn = resnet.resnet50_rpn(base, anchors_per_loc=A)
    can_use = rs.rand(1, rows, cols, A) < 0.012; is_pos = rs.rand(1, rows, cols, A) < 0.01
    yc = np.concatenate([can_use, is_pos], axis=3)
    yb = np.concatenate([np.repeat(can_use & is_pos, 4, axis=3).astype(np.float32), (rs.randn(1, rows, cols, 4 * A) * is_pos.repeat(4, axis=3)).astype(np.float32)], axis=3)
    rpn.compile(train.SGD(1e-3, 0.9))
    tr = rpn._trainer
    for _ in range(5):
        rpn.train_on_batch(x, [yc, yb])
    res = {True: [], False: []}
    for rnd in range(5):
        for g in (True, False):
            tr.use_graph = g
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(20):
                rpn.train_on_batch(x, [yc, yb])
            torch.cuda.synchronize()
            res[g].append(1e3 * (time.perf_counter() - t0) / 20)
    print(DT, "rpn step ms: graph", ["%.3f" % v for v in res[True]], "eager", ["%.3f" % v for v in res[False]], flush=True)
