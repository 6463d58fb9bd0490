import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from faster_rcnn_amd import ops
from oracle import np_ref
g = np.load("tests/golden/dets_legacy.npz")
rois = g["rois"].astype(np.float32); n = len(rois)
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
out = ops.detections(dev(rois), dev(np.array([n], np.int32)), dev(g["out_cls"][:n]), dev(g["out_reg"][:n]), 64, 20, 0.0, 16.0, 1.6)
nd = int(out["n_dets"].item())
got_roi = out["det_roi"].cpu().numpy()[:nd]
# oracle kept roi rows
cls = g["out_cls"][:n].argmax(1); conf = g["out_cls"][:n].max(1)
valid = cls != 20
boxes = {}
for r in range(n):
    if valid[r]:
        t = g["out_reg"][r, 4*cls[r]:4*cls[r]+4] / np_ref.BBREG_MULTIPLIERS
        boxes[r] = [16*v for v in np_ref.transform_legacy(rois[r], t)]
want = []
for c in sorted(set(cls[valid])):
    rows = [r for r in range(n) if valid[r] and cls[r] == c]
    b = np.array([boxes[r] for r in rows]); p = conf[rows]
    _, _, pick = np_ref.nms(b, p, 0.5, 2000)
    want += [rows[i] for i in pick]
extra = sorted(set(got_roi) - set(want)); missing = sorted(set(want) - set(got_roi))
print("nd", nd, "want", len(want), "extra", extra, "missing", missing, "nvalid", valid.sum())
order = np.argsort(-conf[valid], kind="stable"); vrows = np.nonzero(valid)[0][order]
pos = {r: i for i, r in enumerate(vrows)}
for e in extra:
    c = cls[e]
    better = [r for r in want if cls[r] == c and conf[r] > conf[e]]
    for r in better:
        a, b = np.array(boxes[r]), np.array(boxes[e])
        w = max(0, min(a[2], b[2]) - max(a[0], b[0]) + 1); h = max(0, min(a[3], b[3]) - max(a[1], b[1]) + 1)
        ov = w*h / ((a[2]-a[0]+1)*(a[3]-a[1]+1) + (b[2]-b[0]+1)*(b[3]-b[1]+1) - w*h)
        if ov > 0.5: print("extra", e, "sorted pos", pos[e], "should be suppressed by", r, "pos", pos[r], "ov", ov)
