"""Constants of the reference's shared_constants.py (shared_constants.py:5-18)."""
import math

import numpy as np

BBREG_MULTIPLIERS = np.array([10, 10, 5, 5], dtype=np.float32)
DEFAULT_ANCHOR_SCALES = np.array([16, 32, 64, 128, 256, 512])
DEFAULT_ANCHOR_RATIOS = np.array([[1, 1], [1, 2], [2, 1]])


def _anchors(scales, ratios):
    naive = np.array([[s * h, s * w] for s in scales for h, w in ratios])
    rat = np.array([math.sqrt(s * h * s * w) / s for s in scales for h, w in ratios])
    return (naive // rat[:, None]).astype(int)


DEFAULT_ANCHORS = _anchors(DEFAULT_ANCHOR_SCALES, DEFAULT_ANCHOR_RATIOS)
DEFAULT_ANCHORS_PER_LOC = len(DEFAULT_ANCHORS)
DEFAULT_NUM_ITERATIONS = 10
DEFAULT_LEARN_RATE = 1e-3
DEFAULT_MOMENTUM = 0.9
RESIZE_MIN_SIZE = 600
RESIZE_MAX_SIZE = 1000
NUM_ROIS = 64
