import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from tests.test_train_gpu import image, rpn_targets
from faster_rcnn_amd import resnet, train
from faster_rcnn_amd.weights import synthetic_resnet
from oracle import keras_train_ref as kt
A = 9
w0 = synthetic_resnet(50, anchors_per_loc=A, seed=7)
old = {k: [np.array(a, dtype=np.float64) for a in v] for k, v in w0.items()}
x = image(112, 144)
rows, cols = resnet.get_conv_rows_cols(112, 144)
y_class, y_bbreg = rpn_targets(rows, cols, A)
base = resnet.resnet50_base(weights={k: [a.copy() for a in v] for k, v in w0.items()})
rpn = resnet.resnet50_rpn(base, anchors_per_loc=A)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1
l2 = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-4
tr = train.RpnTrainer(rpn, l2=l2)
kind = sys.argv[3] if len(sys.argv) > 3 else "sgd"
tr.compile(train.SGD(lr=1e-3, momentum=0.9) if kind == "sgd" else train.Adam(lr=1e-3))
ref_opt = kt.Optim(kind, 1e-3)
ref_w = w0
for s in range(steps):
    print(tr.train_on_batch(x, [y_class, y_bbreg]))
    ref_w, rl, _ = kt.rpn_train_step(ref_w, x, y_class, y_bbreg, A, ref_opt, l2=l2)
    print(rl)
tr.sync_weights()
names = kt.conv_layer_names(50, [4]) + ["rpn_conv1", "rpn_out_cls", "rpn_out_bbreg"]
for n in names:
    for i, (o, g, w) in enumerate(zip(old[n], rpn.weights[n], ref_w[n])):
        dg, dw = np.asarray(g, np.float64) - o, np.asarray(w, np.float64) - o
        scale = np.abs(dw).max()
        adj = np.maximum(np.abs(dg - dw) - 2 * 2.0 ** -23 * np.abs(o), 0).max() / scale
        print("%-18s %d maxupd %.3e relerr %.2e adj %.2e corr %.6f" % (n, i, scale, np.abs(dg - dw).max() / scale, adj, float((dg * dw).sum() / np.sqrt((dg * dg).sum() * (dw * dw).sum()))))
n = sys.argv[4] if len(sys.argv) > 4 else "res4d_branch2c"
o, g, w = old[n][0], np.asarray(rpn.weights[n][0], np.float64), np.asarray(ref_w[n][0], np.float64)
err = np.abs((g - o) - (w - o))
idx = np.argsort(err.ravel())[::-1][:8]
for i in idx:
    print("w0 %.6e  got_d %.6e want_d %.6e err %.3e  idx %s" % (o.ravel()[i], (g - o).ravel()[i], (w - o).ravel()[i], err.ravel()[i], np.unravel_index(i, o.shape)))
print("err vs |w| corr", np.corrcoef(err.ravel(), np.abs(o).ravel())[0, 1], "mean err", err.mean(), "per-cout max err", err.reshape(-1, err.shape[-1]).max(0)[:8])
