#!/usr/bin/env python3
"""Where the pieces of a replayed training step run, WITHOUT a profiler (rocprofv3 slows hipGraphLaunch on the host enough to change the
picture): timing events around G0 (prefix stream), G1, the backward pieces and the update; offsets against the start of G1 of the
same step, averaged.   usage: r6_event_timeline.py [bf16|f32] [rpn|det]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from faster_rcnn_amd import resnet, train
from faster_rcnn_amd.weights import synthetic_resnet

DT = sys.argv[1] if len(sys.argv) > 1 else "bf16"
H, W, A, C = 600, 1000, 9, 21
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
xd = torch.from_numpy(x.astype(np.float32)).cuda()
ycd = torch.from_numpy(y_class.astype(np.float32)).cuda().reshape(-1, 2 * A)
ybd = torch.from_numpy(y_bbreg).cuda().reshape(-1, 8 * A)
step = lambda: rpn.train_on_batch(xd, [ycd, ybd], defer=True)
prev = None
for _ in range(20):
    cur = step()
    if prev is not None: prev.result()
    prev = cur
prev.result()
tr = rpn._trainer
sg = next(iter(tr._graphs["graphs"].values()))
marks = []          # per step dict name -> event


class R:
    def __init__(self, g, tag, first=False): self.g, self.tag, self.first = g, tag, first
    def replay(self):
        if self.first:
            marks.append({})
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); self.g.replay(); b.record()
        marks[-1].setdefault(self.tag, []).append((a, b))
    def reset(self): self.g.reset()


sg.g0 = R(sg.g0, "g0", first=True)
sg.g1 = R(sg.g1, "g1")
sg.bwd = [(lane, R(g, "bwd_%s%d" % (lane, i))) for i, (lane, g) in enumerate(sg.bwd)]
orig_apply = tr._exchange_and_apply
def apply():
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); orig_apply(); b.record()
    marks[-1]["apply"] = [(a, b)]
tr._exchange_and_apply = apply
N = 60
torch.cuda.synchronize()
t0 = time.perf_counter()
prev = None
for _ in range(N):
    cur = step()
    if prev is not None: prev.result()
    prev = cur
prev.result()
torch.cuda.synchronize()
print("step %.3f ms (with %d timing events per step)" % (1e3 * (time.perf_counter() - t0) / N, 2 * sum(len(v) for v in marks[0].values())))
ref = [m["g1"][0][0] for m in marks]
keys = list(marks[10].keys())
for k in keys:
    s = np.mean([ref[i].elapsed_time(marks[i][k][0][0]) for i in range(10, N)]) * 1e3
    e = np.mean([ref[i].elapsed_time(marks[i][k][0][1]) for i in range(10, N)]) * 1e3
    print("  %-12s start %8.1f  end %8.1f us (from this step's G1 start)" % (k, s, e))
s = np.mean([ref[i].elapsed_time(marks[i + 1]["g0"][0][0]) for i in range(10, N - 1)]) * 1e3
e = np.mean([ref[i].elapsed_time(marks[i + 1]["g0"][0][1]) for i in range(10, N - 1)]) * 1e3
print("  next G0      start %8.1f  end %8.1f us" % (s, e))
s = np.mean([ref[i].elapsed_time(ref[i + 1]) for i in range(10, N - 1)]) * 1e3
print("  next G1      start %8.1f us" % s)
