"""Mirror of the reference's util.py (util.py:9-253) over the HIP library.

Same names, argument meaning, return types and in-place behaviour; the array functions hand
their numpy arguments to the C ABI (device copies are plumbing) and return numpy arrays.
Scalar helpers (``calc_iou``, ``transform``, ``get_reg_params``, ``get_anchors``) are O(1)
host arithmetic and stay on the host exactly as written in the reference.
"""
import math

import numpy as np
import torch

from . import ops
from .shared_constants import (DEFAULT_ANCHOR_RATIOS, DEFAULT_ANCHOR_SCALES, RESIZE_MAX_SIZE, RESIZE_MIN_SIZE)


def calc_iou(coords1, coords2):
    """util.py:9-24 (scalar IoU, no +1)."""
    ix1, iy1 = max(coords1[0], coords2[0]), max(coords1[1], coords2[1])
    ix2, iy2 = min(coords1[2], coords2[2]), min(coords1[3], coords2[3])
    inter = 0 if (ix2 < ix1 or iy2 < iy1) else (ix2 - ix1) * (iy2 - iy1)
    if inter <= 0:
        return 0.0
    a1 = (coords1[2] - coords1[0]) * (coords1[3] - coords1[1])
    a2 = (coords2[2] - coords2[0]) * (coords2[3] - coords2[1])
    return inter * 1.0 / (a1 + a2 - inter)


def transform(anchor_coords, reg_targets):
    """util.py:55-74: scalar f64 box decode (math.exp, no rounding)."""
    x1, y1, x2, y2 = anchor_coords
    cxa, cya = (x1 + x2) / 2, (y1 + y2) / 2
    wa, ha = x2 - x1, y2 - y1
    tx, ty, tw, th = reg_targets
    cx, cy = tx * wa + cxa, ty * ha + cya
    w, h = math.exp(tw) * wa, math.exp(th) * ha
    x, y = cx - w / 2, cy - h / 2
    return x, y, x + w, y + h


def transform_np_inplace(coords, reg_targets):
    """util.py:111-142.  Mutates ``coords`` (f32 (n,4)) and returns it."""
    assert coords.dtype == np.float32 and coords.flags.c_contiguous
    c = torch.from_numpy(coords).cuda()
    d = torch.from_numpy(np.ascontiguousarray(reg_targets, dtype=np.float32)).cuda()
    ops.transform_inplace(c, d)
    coords[...] = c.cpu().numpy()
    return coords


def cross_ious(boxes1, boxes2):
    """util.py:146-177: (m,n) f32 IoU matrix; boxes1 f32 or int16, boxes2 f32."""
    b1 = np.ascontiguousarray(boxes1)
    if b1.dtype != np.int16:
        b1 = b1.astype(np.float32)
    b2 = np.ascontiguousarray(boxes2, dtype=np.float32).reshape(-1, 4)
    return ops.cross_ious(torch.from_numpy(b1.reshape(-1, 4)).cuda(), torch.from_numpy(b2).cuda()).cpu().numpy()


def get_reg_params(anchor_coords, bbox_coords):
    """util.py:180-206 (scalar; numpy scalar promotion exactly as the reference's expression)."""
    bbox_x1, bbox_y1, bbox_x2, bbox_y2 = bbox_coords
    anchor_x1, anchor_y1, anchor_x2, anchor_y2 = anchor_coords
    bcx, bcy = (bbox_x2 + bbox_x1) / 2.0, (bbox_y2 + bbox_y1) / 2.0
    bw, bh = bbox_x2 - bbox_x1, bbox_y2 - bbox_y1
    acx, acy = (anchor_x2 + anchor_x1) / 2.0, (anchor_y2 + anchor_y1) / 2.0
    aw, ah = anchor_x2 - anchor_x1, anchor_y2 - anchor_y1
    return (bcx - acx) / aw, (bcy - acy) / ah, np.log(bw / aw), np.log(bh / ah)


def resize_imgs(imgs, min_size=RESIZE_MIN_SIZE, max_size=RESIZE_MAX_SIZE):
    """util.py:209-226."""
    out, ratios = [], []
    for img in imgs:
        r, ratio = img.resize_within_bounds(min_size=min_size, max_size=max_size)
        out.append(r)
        ratios.append(ratio)
    return out, ratios


def get_bbox_coords(gt_boxes):
    """util.py:229-238: list of GroundTruthBox -> (n,4) f32."""
    res = np.zeros((len(gt_boxes), 4), dtype=np.float32)
    for i, b in enumerate(gt_boxes):
        res[i] = b.corners
    return res


def get_anchors(anchor_scales=DEFAULT_ANCHOR_SCALES, anchor_ratios=DEFAULT_ANCHOR_RATIOS):
    """util.py:242-253: (A,2) int [height, width]."""
    naive = np.array([[s * h, s * w] for s in anchor_scales for h, w in anchor_ratios])
    ratios = np.array([math.sqrt(s * h * s * w) / s for s in anchor_scales for h, w in anchor_ratios])
    return (naive // ratios[:, None]).astype(int)
