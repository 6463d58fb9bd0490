"""Mirror of the reference's rpn_util.py (rpn_util.py:11-350): RPN training targets.

``RpnTrainingManager`` keeps the reference's constructor, ``batched_image`` and ``rpn_y_true``
contract.  Anchor generation, IoU, positive/negative/out-of-bounds assignment and the
regression targets run in ONE C-ABI call (frcnn_rpn_assign); the batch sampling stays on the
host because it consumes the global Python ``random`` stream exactly like the reference
(rpn_util.py:324-350) -- moving it would change which anchors get sampled.
"""
import random

import numpy as np

from . import ops
from .shared_constants import DEFAULT_ANCHORS
from .util import get_bbox_coords

POS_OVERLAP = 0.7
NEG_OVERLAP = 0.3
SAMPLE_SIZE = 256
MAX_POS_SAMPLES = 128


class RpnTrainingManager:
    def __init__(self, calc_conv_dims, stride, preprocess_func, anchor_dims=DEFAULT_ANCHORS):
        self._cache = {}
        self.calc_conv_dims = calc_conv_dims
        self.stride = stride
        self.preprocess_func = preprocess_func
        self.anchor_dims = anchor_dims

    def batched_image(self, image):
        return np.expand_dims(self.preprocess_func(image.data), axis=0)

    def _process(self, image):
        """rpn_util.py:54-103.  The cache entry holds numpy arrays like the reference's."""
        conv_rows, conv_cols = self.calc_conv_dims(image.height, image.width)
        gt = get_bbox_coords(image.gt_boxes)
        can_use, is_pos, bbreg, _ = ops.rpn_assign(conv_rows, conv_cols, self.anchor_dims, self.stride, gt, image.width, image.height)
        self._cache[image.cache_key] = {
            "can_use": can_use.cpu().numpy().astype(bool),
            "is_pos": is_pos.cpu().numpy().astype(bool),
            "bbreg_targets": bbreg.cpu().numpy(),
        }

    def rpn_y_true(self, image):
        """rpn_util.py:106-140: (y_class (1,R,C,2A) bool, y_bbreg (1,R,C,8A) f32).  Like the
        reference the cache entry is dropped right after use (:121-123)."""
        if image.cache_key not in self._cache:
            self._process(image)
        results = self._cache.pop(image.cache_key)
        can_use = _apply_sampling(results["is_pos"], results["can_use"])
        conv_rows, conv_cols = self.calc_conv_dims(image.height, image.width)
        A = len(self.anchor_dims)
        is_pos = results["is_pos"].reshape(conv_rows, conv_cols, A)
        can_use = can_use.reshape(conv_rows, conv_cols, A)
        y_class = np.concatenate([can_use, is_pos], axis=2)
        bbreg_can_use = np.repeat(np.logical_and(is_pos, can_use), 4, axis=2)
        bbreg_targets = results["bbreg_targets"].reshape(conv_rows, conv_cols, 4 * A)
        y_bbreg = np.concatenate([bbreg_can_use, bbreg_targets], axis=2)
        return np.expand_dims(y_class, axis=0), np.expand_dims(y_bbreg, axis=0)


def _idx_to_conv(idx, conv_width, anchors_per_loc):
    """rpn_util.py:143-156."""
    divisor = conv_width * anchors_per_loc
    y, rem = idx // divisor, idx % divisor
    return y, rem // anchors_per_loc, rem % anchors_per_loc


def _get_conv_center(conv_x, conv_y, stride):
    """rpn_util.py:169-181."""
    return int(stride * (conv_x + 0.5)), int(stride * (conv_y + 0.5))


def _get_all_anchor_coords(conv_rows, conv_cols, anchor_dims, stride):
    """rpn_util.py:276-298: (N,4) f32 anchors in image pixels."""
    return ops.anchors_image(conv_rows, conv_cols, anchor_dims, stride).cpu().numpy()


def _get_out_of_bounds_idxs(anchor_coords, img_width, img_height):
    """rpn_util.py:302-310."""
    a = anchor_coords
    return np.where((a[:, 0] < 0) | (a[:, 1] < 0) | (a[:, 2] >= img_width) | (a[:, 3] >= img_height))[0]


def _apply_sampling(is_pos, can_use):
    """rpn_util.py:324-350 (host; global ``random`` stream; mutates can_use)."""
    pos_locs = np.where(np.logical_and(is_pos == 1, can_use == 1))[0]
    neg_locs = np.where(np.logical_and(is_pos == 0, can_use == 1))[0]
    num_pos, num_neg = len(pos_locs), len(neg_locs)
    if num_pos > MAX_POS_SAMPLES:
        locs_off = random.sample(range(num_pos), num_pos - MAX_POS_SAMPLES)
        can_use[pos_locs[locs_off]] = 0
        num_pos = MAX_POS_SAMPLES
    if num_neg + num_pos > SAMPLE_SIZE:
        locs_off = random.sample(range(num_neg), num_neg + num_pos - SAMPLE_SIZE)
        can_use[neg_locs[locs_off]] = 0
    return can_use
