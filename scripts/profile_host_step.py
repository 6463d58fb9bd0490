#!/usr/bin/env python3
"""cProfile of the host side of the mixed-precision RPN training step (where do the ~1.4 ms of Python per step go).  Dev tool."""
import cProfile
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from faster_rcnn_amd import resnet, train
from faster_rcnn_amd.weights import synthetic_resnet

DT = "bf16" if "--f32" not in sys.argv else "f32"
H, W, A = 600, 1000, 9
rs = np.random.RandomState(0)
x = (rs.randint(0, 256, (H, W, 3)).astype(np.float64) - np.array([103.939, 116.779, 123.68]))[None]
rows, cols = resnet.get_conv_rows_cols(H, W)
w = synthetic_resnet(50, anchors_per_loc=A, num_classes=21, seed=1)
base = resnet.resnet50_base(weights=w, weight_regularizer=resnet.WEIGHT_REGULARIZER, bias_regularizer=resnet.BIAS_REGULARIZER, dtype=DT)
rpn = resnet.resnet50_rpn(base, anchors_per_loc=A)
can_use = rs.rand(1, rows, cols, A) < 0.012
is_pos = rs.rand(1, rows, cols, A) < 0.01
y_class = np.concatenate([can_use, is_pos], axis=3)
y_bbreg = np.concatenate([np.repeat(can_use & is_pos, 4, axis=3).astype(np.float32), (rs.randn(1, rows, cols, 4 * A) * is_pos.repeat(4, axis=3)).astype(np.float32)], axis=3)
rpn.compile(train.SGD(1e-3, 0.9))
prev = None
for _ in range(30):
    cur = rpn.train_on_batch(x, [y_class, y_bbreg], defer=True)
    if prev is not None:
        prev.result()
    prev = cur
torch.cuda.synchronize()
N = 200
pr = cProfile.Profile()
pr.enable()
for _ in range(N):
    cur = rpn.train_on_batch(x, [y_class, y_bbreg], defer=True)
    prev.result()
    prev = cur
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime")
print("per step (us): total %.0f" % (st.total_tt / N * 1e6))
st.print_stats(28)
