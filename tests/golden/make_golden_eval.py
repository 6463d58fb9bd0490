#!/usr/bin/env python3
"""Golden vectors for eval_dets.voc_eval (eval_dets.py:37-127) from the IMPORTED reference (build container only):
the VOC_test fixture (image 000005: 5 objects, 3 chairs of which 1 'difficult') against a synthetic detection file
per class.  Writes tests/golden/eval_dets.npz = the detection lines and the (rec, prec, ap) the reference returns.
    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_eval.py
"""
import contextlib
import io
import os
import sys
import types

import numpy as np

REF = "/root/reference/faster_rcnn"
OUT = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.modules["cv2"] = types.ModuleType("cv2")
if not hasattr(np, "bool"):
    np.bool = bool
sys.path.insert(0, REF)
import eval_dets  # noqa: E402
from data.voc_data_helpers import extract_img_data  # noqa: E402

voc = os.path.join(OUT, "VOC_test")
imageset = os.path.join(voc, "ImageSets", "Main", "trainval.txt")
img = extract_img_data(voc, "000005")
rs = np.random.RandomState(5)
out = {}
for cls in sorted({b.obj_cls for b in img.gt_boxes}) + ["dog"]:
    gts = [b for b in img.gt_boxes if b.obj_cls == cls]
    lines = []
    for b in gts:                                   # a good hit, a duplicate, a near miss (IoU around the threshold)
        x1, y1, x2, y2 = [int(v) + 1 for v in b.corners]        # detection files carry +1 coordinates (voc_dets.py:126)
        lines.append("000005 %.4f %d %d %d %d" % (0.5 + 0.4 * rs.rand(), x1 + 2, y1 - 1, x2 + 3, y2 + 2))
        lines.append("000005 %.4f %d %d %d %d" % (0.3 + 0.2 * rs.rand(), x1, y1, x2, y2))
        w = x2 - x1
        lines.append("000005 %.4f %d %d %d %d" % (0.2 + 0.6 * rs.rand(), x1 + w // 3, y1, x2 + w // 3, y2))
    for _ in range(4):                              # clutter
        x, y = rs.randint(1, 300), rs.randint(1, 200)
        lines.append("000005 %.4f %d %d %d %d" % (rs.rand(), x, y, x + rs.randint(20, 150), y + rs.randint(20, 150)))
    path = os.path.join("/tmp", "golden_eval_%s.txt" % cls)
    with open(path, "w") as f:
        f.write("\n".join(lines) + "\n")
    with contextlib.redirect_stdout(io.StringIO()):
        rec, prec, ap = eval_dets.voc_eval(voc, path, imageset, cls)
    out["lines_" + cls] = np.array(lines)
    out["rec_" + cls], out["prec_" + cls], out["ap_" + cls] = np.asarray(rec), np.asarray(prec), np.float64(ap)
    print(cls, "gt", len(gts), "dets", len(lines), "ap %.4f" % ap)
np.savez(os.path.join(OUT, "eval_dets.npz"), **out)
