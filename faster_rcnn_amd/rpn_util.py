"""Mirror of the reference's rpn_util.py (rpn_util.py:11-350): RPN training targets.

``RpnTrainingManager`` keeps the reference's constructor, ``batched_image`` and ``rpn_y_true``
contract.  Anchor generation, IoU, positive/negative/out-of-bounds assignment and the
regression targets run in ONE C-ABI call (frcnn_rpn_assign); the batch sampling stays on the
host because it consumes the global Python ``random`` stream exactly like the reference
(rpn_util.py:324-350) -- moving it would change which anchors get sampled.
"""
import ctypes
import random
from math import ceil as _ceil, log as _log

import numpy as np

from . import _lib, ops
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

    # ------------------------------------------------------------------ device-resident fast path (train_util.train_rpn)
    # The same two calls with everything but the host RNG left on the device: ``prefetch`` is the part of batched_image +
    # _process that needs no random draw (upload of the decoded frame, resize / preprocess, frcnn_rpn_assign, the ascending
    # positive / negative lists) and may run a step AHEAD on the manager's own stream; ``rpn_inputs_dev`` reads the two counts
    # (8 bytes), makes the reference's two ``random.sample`` draws in the reference's order (global ``random`` stream, _sample_off)
    # and packs y_class / y_bbreg on the device.  What comes back are the float32 device tensors train_on_batch's feed would
    # have built from batched_image(image) and rpn_y_true(image): 1.2 MB of masks and targets, 1 MB of packed targets and 14 MB
    # of float64 image never cross PCIe.  tests/test_train_loop_gpu.py: same bits, same weights after N iterations.
    def _own_stream(self):
        from . import feed
        return feed.manager_stream()                          # (one per process: see feed.manager_stream)

    def decode_ahead(self, image):
        """Start the JPEG decode of an image a later ``prefetch`` will want, on feed's background thread (train_util names it two iterations ahead)."""
        from . import feed
        if feed.device_preprocess(self.preprocess_func):
            feed.decode_ahead(image)

    def prefetch(self, image):
        import torch
        from . import feed
        dev = self.__dict__.setdefault("_dev", {})
        if image.cache_key in dev:
            return
        st = self._own_stream()
        with torch.cuda.stream(st):
            x = feed.device_image(image, self.preprocess_func)
            conv_rows, conv_cols = self.calc_conv_dims(image.height, image.width)
            gt = get_bbox_coords(image.gt_boxes)
            can_use, is_pos, bbreg, _ = ops.rpn_assign(conv_rows, conv_cols, self.anchor_dims, self.stride, gt, image.width, image.height)
            pos_locs, neg_locs, counts = ops.rpn_sample_lists(can_use, is_pos)
            counts_host, ev = feed.ring().host_slot((2,), torch.int32)
            counts_host.copy_(counts, non_blocking=True)
            ev.record()
        dev[image.cache_key] = (x, can_use, is_pos, bbreg, pos_locs, neg_locs, counts_host, ev, conv_rows * conv_cols)

    def rpn_inputs_dev(self, image):
        """-> (x (1,H,W,3), y_class (cells,2A), y_bbreg (cells,8A)) float32 device tensors, marked with the event the step waits for."""
        import torch
        from . import feed
        self.prefetch(image)
        x, can_use, is_pos, bbreg, pos_locs, neg_locs, counts_host, ev, cells = self._dev.pop(image.cache_key)
        ev.synchronize()
        num_pos, num_neg = (int(v) for v in counts_host.tolist())
        off_pos, off_neg = _sample_off(num_pos, num_neg)           # the reference's two draws, in its order (rpn_util.py:336-348)
        with torch.cuda.stream(self._own_stream()):
            up = lambda a: None if a is None else feed.upload(a)
            yc, yb = ops.rpn_pack_targets(can_use, is_pos, bbreg, cells, len(self.anchor_dims), pos_locs, num_pos, up(off_pos), neg_locs, num_neg, up(off_neg))
            return feed.Ready.mark(x, yc, yb)

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


FAST_SAMPLE = True          # False: the interpreter's own random.sample (what sample_range is tested against)


def sample_range(n, k):
    """``random.sample(range(n), k)`` -- the same list, the global ``random`` stream left in the same state -- computed by
    frcnn_host_mt_sample_range on the interpreter's own generator state.  The reference draws 20 000-60 000 positions per
    image this way (rpn_util.py:343-348: every usable negative but ~128 is switched OFF), 7-80 ms of interpreter time against a
    2 ms training step; the C replay takes ~0.3 ms.  -> int32 numpy array of length k."""
    if not FAST_SAMPLE or type(random._inst) is not random.Random or k == 0 or n >= 2 ** 31:
        return np.asarray(random.sample(range(n), k), dtype=np.int32)
    if not 0 <= k <= n:
        raise ValueError("Sample larger than population or is negative")
    setsize = 21                                             # Lib/random.py sample(): the interpreter's own arithmetic decides the branch
    if k > 5:
        setsize += 4 ** _ceil(_log(k * 3, 4))
    version, internal, gauss = random.getstate()
    state = np.array(internal[:624], dtype=np.uint32)
    index = ctypes.c_int32(internal[624])
    out = np.empty(k, dtype=np.int32)
    _lib.call("frcnn_host_mt_sample_range", state.ctypes.data, ctypes.addressof(index), int(n), int(k), 1 if n <= setsize else 0, out.ctypes.data)
    random.setstate((version, tuple(state.tolist()) + (index.value,), gauss))
    return out


def _sample_off(num_pos, num_neg):
    """The two draws of rpn_util.py:336-348 given only the COUNTS: positions (in the ascending pos / neg lists) to switch off."""
    off_pos = off_neg = None
    if num_pos > MAX_POS_SAMPLES:
        off_pos = sample_range(num_pos, num_pos - MAX_POS_SAMPLES)
        num_pos = MAX_POS_SAMPLES
    if num_neg + num_pos > SAMPLE_SIZE:
        off_neg = sample_range(num_neg, num_neg + num_pos - SAMPLE_SIZE)
    return off_pos, off_neg


def _apply_sampling(is_pos, can_use):
    """rpn_util.py:324-350 (host; global ``random`` stream; mutates can_use)."""
    pos_locs = np.where(np.logical_and(is_pos == 1, can_use == 1))[0]
    neg_locs = np.where(np.logical_and(is_pos == 0, can_use == 1))[0]
    off_pos, off_neg = _sample_off(len(pos_locs), len(neg_locs))
    if off_pos is not None:
        can_use[pos_locs[off_pos]] = 0
    if off_neg is not None:
        can_use[neg_locs[off_neg]] = 0
    return can_use
