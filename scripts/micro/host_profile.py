#!/usr/bin/env python3
"""cProfile of the host side of the mixed-precision RPN training step (deferred losses): where the Python time goes.  Dev tool."""
import cProfile, os, pstats, sys, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from faster_rcnn_amd import resnet, train
from faster_rcnn_amd.weights import synthetic_resnet
H, W, A, C = 600, 1000, 9, 21
DT = "bf16" if "--f32" not in sys.argv else "f32"
rs = np.random.RandomState(0)
x = (rs.randint(0, 256, (H, W, 3)).astype(np.float64) - np.array([103.939, 116.779, 123.68]))[None]
rows, cols = resnet.get_conv_rows_cols(H, W)
w = synthetic_resnet(50, anchors_per_loc=A, num_classes=C, seed=1)
base = resnet.resnet50_base(weights=w, weight_regularizer=resnet.WEIGHT_REGULARIZER, bias_regularizer=resnet.BIAS_REGULARIZER, dtype=DT)
rpn = resnet.resnet50_rpn(base, anchors_per_loc=A)
can_use = rs.rand(1, rows, cols, A) < 0.012
is_pos = rs.rand(1, rows, cols, A) < 0.01
y_class = np.concatenate([can_use, is_pos], axis=3)
y_bbreg = np.concatenate([np.repeat(can_use & is_pos, 4, axis=3).astype(np.float32), (rs.randn(1, rows, cols, 4 * A) * is_pos.repeat(4, axis=3)).astype(np.float32)], axis=3)
rpn.compile(train.SGD(1e-3, 0.9))
for _ in range(5):
    rpn.train_on_batch(x, [y_class, y_bbreg])
torch.cuda.synchronize()
import time
N = 30
t0 = time.perf_counter()
pend = [rpn.train_on_batch(x, [y_class, y_bbreg], defer=True) for _ in range(N)]
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print("host enqueue %.3f ms per step; until the GPU is done %.3f ms per step" % (1e3 * t_host / N, 1e3 * t_all / N))
pr = cProfile.Profile()
pr.enable()
pend = [rpn.train_on_batch(x, [y_class, y_bbreg], defer=True) for _ in range(N)]
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22)
print("\n".join(l[:150] for l in s.getvalue().splitlines()[:45]))
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(r"train\.py|models\.py|nets\.py", 24)
print("-- by cumulative time, train.py / models.py / nets.py")
print("\n".join(l[:150] for l in s.getvalue().splitlines()[:45]))
