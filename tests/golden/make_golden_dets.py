#!/usr/bin/env python3
"""Golden vectors for the detection post-process (voc_dets.get_dets, voc_dets.py:20-88).

voc_dets.py imports the Keras model builders at module level, so it is imported with EMPTY
stub modules for resnet / vgg / args_util / cv2 (get_dets itself touches none of them) and fed
a fake training manager / detector that return seeded arrays.

Run under BOTH interpreters (build container only):
    /opt/conda/bin/python3.9 tests/golden/make_golden_dets.py legacy   # numpy 1.26: value-based
        scalar promotion, i.e. the semantics of the reference's pinned numpy 1.13.3
    python tests/golden/make_golden_dets.py nep50                      # numpy 2.2: NEP-50
The two differ only in scalar float32-vs-float64 promotion inside util.transform; the product
follows the LEGACY result (the reference's own environment).
"""
import os
import sys
import types

import numpy as np

mode = sys.argv[1]
REF = "/root/reference/faster_rcnn"
OUT = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
for name in ("cv2", "resnet", "vgg", "args_util"):
    sys.modules[name] = types.ModuleType(name)
sys.modules["args_util"].base_paths_to_imgs = sys.modules["args_util"].resize_dims_from_str = None
sys.modules["args_util"].anchor_scales_from_str = None
if not hasattr(np, "bool"):
    np.bool = bool
sys.path.insert(0, REF)
import io
import contextlib

import voc_dets  # noqa: E402
from data.voc_data_helpers import VOC_CLASS_MAPPING  # noqa: E402

g = np.load(os.path.join(OUT, "numpy_half.npz"))
rois = g["prop_c2_8000_kept"]                      # (300,4) int16, conv units
rs = np.random.RandomState(11)
n_pad = 320
C = 21
logits = rs.randn(n_pad, C).astype(np.float32) * 2.0
logits[:, 20] += 1.0                                # a fair share of background rows
e = np.exp(logits - logits.max(axis=1, keepdims=True))
out_cls = (e / e.sum(axis=1, keepdims=True)).astype(np.float32)
out_reg = (rs.randn(n_pad, 4 * (C - 1)) * 1.5).astype(np.float32)
# the reference pads the last batch with copies of ITS first RoI (row 256) -> same detector outputs
out_cls[300:] = out_cls[256]
out_reg[300:] = out_reg[256]


class Mgr:
    class_mapping = VOC_CLASS_MAPPING

    def get_det_inputs(self, image):
        return "conv", rois


class Det:
    def __init__(self):
        self.calls = 0

    def predict(self, inputs):
        b = self.calls
        self.calls += 1
        return out_cls[None, 64 * b:64 * b + 64], out_reg[None, 64 * b:64 * b + 64]


# "tb": a threshold ONE f64 step above a row's own f32 score.  `confidence < det_threshold` (voc_dets.py:57) compares an
# np.float32 scalar with a Python float: in f64 under the legacy rules (the row is skipped), in f32 under NEP 50 (the
# threshold rounds to the score itself, the row stays) -- the two goldens differ in exactly that row's detection.
fg = [r for r in range(300) if np.argmax(out_cls[r]) != 20]
r_b = max(fg, key=lambda r: out_cls[r].max())       # the most confident foreground row: nothing suppresses it
thr_b = float(np.nextafter(np.float64(out_cls[r_b].max()), 1.0))
res = {}
for tag, thr, ratio in (("t0", 0.0, 1.6), ("t5", 0.5, 1.0), ("t0r", 0.0, 600 / 375), ("tb", thr_b, 1.0)):
    with contextlib.redirect_stdout(io.StringIO()):
        dets = voc_dets.get_dets(Mgr(), Det(), None, ratio, det_threshold=thr)
    res[tag + "_bbox"] = np.array([d["bbox"] for d in dets], dtype=np.int64).reshape(-1, 4)
    res[tag + "_cls"] = np.array([VOC_CLASS_MAPPING[d["cls_name"]] for d in dets], dtype=np.int32)
    res[tag + "_prob"] = np.array([d["prob"] for d in dets], dtype=np.float32)
    res[tag + "_args"] = np.array([thr, ratio])
res["rois"] = rois
res["out_cls"] = out_cls
res["out_reg"] = out_reg
np.savez_compressed(os.path.join(OUT, "dets_%s.npz" % mode), **res)
print(mode, np.__version__, {k: v.shape for k, v in res.items() if k.endswith("_bbox")})
