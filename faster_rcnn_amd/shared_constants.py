"""Numeric constants of the hot path, with the reference line each one comes from (shared_constants.py:5-18).

They are baked into kernels and tests alike: the box-regression scaling, the anchor family (scales x aspect ratios ->
integer (height, width) pairs), the resize bounds and the reference's training defaults.
"""
import math

import numpy as np

# --- box regression: targets are (tx, ty, tw, th) scaled by these before the loss, un-scaled at decode (:5)
BBREG_MULTIPLIERS = np.asarray((10, 10, 5, 5), dtype=np.float32)

# --- anchors (:7-12): every scale paired with every (height, width) aspect ratio, area kept ~ scale^2
DEFAULT_ANCHOR_SCALES = np.asarray((16, 32, 64, 128, 256, 512))
DEFAULT_ANCHOR_RATIOS = np.asarray(((1, 1), (1, 2), (2, 1)))


def anchor_table(scales, ratios):
    """(len(scales) * len(ratios), 2) integer [height, width]: [s*rh, s*rw] floor-divided by sqrt(s*rh * s*rw) / s,
    scale-major -- the arithmetic (and therefore the rounding) of shared_constants.py:9-11 / util.get_anchors."""
    rows = []
    for s in scales:
        for rh, rw in ratios:
            shrink = math.sqrt(s * rh * s * rw) / s
            rows.append((np.float64(s * rh) // shrink, np.float64(s * rw) // shrink))
    return np.asarray(rows).astype(int)


DEFAULT_ANCHORS = anchor_table(DEFAULT_ANCHOR_SCALES, DEFAULT_ANCHOR_RATIOS)
DEFAULT_ANCHORS_PER_LOC = len(DEFAULT_ANCHORS)

# --- input geometry (:16-18): shorter side to 600 unless the longer would pass 1000; RoIs per detector batch
RESIZE_MIN_SIZE, RESIZE_MAX_SIZE = 600, 1000
NUM_ROIS = 64

# --- training defaults (:13-15)
DEFAULT_NUM_ITERATIONS = 10
DEFAULT_LEARN_RATE = 1e-3
DEFAULT_MOMENTUM = 0.9
