#!/usr/bin/env python3
"""Host time of the pieces of a replayed training step (dev): wraps the graph replays and the eager tail with perf_counter."""
import os, sys, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from faster_rcnn_amd import resnet, train, ops
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
acc = collections.defaultdict(float)


def wrap(obj, name, tag):
    f = getattr(obj, name)
    def g(*a, **k):
        t = time.perf_counter()
        r = f(*a, **k)
        acc[tag] += time.perf_counter() - t
        return r
    setattr(obj, name, g)


tr = rpn._trainer
for name in ("_stage", "_exchange_and_apply", "_send_losses", "_finish_update", "_step_graph", "_replay_step"):
    wrap(tr, name, name)
for dev_in in (False, True):
    step = (lambda: rpn.train_on_batch(xd, [ycd, ybd], defer=True)) if dev_in else (lambda: rpn.train_on_batch(x, [y_class, y_bbreg], defer=True))
    prev = None
    for _ in range(20):
        cur = step()
        if prev is not None: prev.result()
        prev = cur
    prev.result()
    sg = next(iter(tr._graphs["graphs"].values()), None)
    if sg is not None and not getattr(sg, "_wrapped", False):
        class R:
            def __init__(self, g, tag): self.g, self.tag = g, tag
            def replay(self):
                t = time.perf_counter(); self.g.replay(); acc[self.tag] += time.perf_counter() - t
            def reset(self): self.g.reset()
        for gname in ("g0", "g1"):
            if getattr(sg, gname) is not None:
                setattr(sg, gname, R(getattr(sg, gname), gname))
        sg.bwd = [(lane, R(g, "bwd_" + lane)) for lane, g in sg.bwd]
        print("backward pieces:", [lane for lane, _ in sg.bwd])
        sg._wrapped = True
    acc.clear()
    torch.cuda.synchronize()
    N = 100
    t0 = time.perf_counter()
    prev = None
    th = 0.0
    for _ in range(N):
        t = time.perf_counter()
        cur = step()
        th += time.perf_counter() - t
        t = time.perf_counter()
        if prev is not None: prev.result()
        acc["wait_result"] += time.perf_counter() - t
        prev = cur
    prev.result()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    print("device inputs" if dev_in else "host inputs", DT, "step %.3f ms; host in train_on_batch %.3f ms" % (1e3 * el / N, 1e3 * th / N))
    for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
        print("   %-22s %7.1f us" % (k, 1e6 * v / N))
