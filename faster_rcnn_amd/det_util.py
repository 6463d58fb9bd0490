"""Mirror of the reference's det_util.py (det_util.py:7-380): proposals, NMS, detector targets.

Same public surface: ``DetTrainingManager(rpn_model, class_mapping, preprocess_func, num_rois,
stride, anchor_dims)`` with ``get_training_input`` / ``get_det_inputs`` and the free function
``nms(boxes, probs, overlap_thresh, max_boxes)``.  Inside, the RPN forward, the decode, the
score ordering, the NMS and the RoI->truth matching stay on the device between C-ABI calls; only
the results the reference hands to its callers come back as numpy.  RoI batch sampling uses
``np.random`` on the host like the reference (det_util.py:260-306).
"""
import numpy as np
import torch

from . import nets, ops
from .shared_constants import BBREG_MULTIPLIERS, DEFAULT_ANCHORS
from .util import get_bbox_coords

CLASSIFIER_MIN_OVERLAP = 0.1
CLASSIFIER_POS_OVERLAP = 0.5
PROBABLE_THRESHOLD = 0.05


class DetTrainingManager:
    def __init__(self, rpn_model, class_mapping, preprocess_func, num_rois=64, stride=16, anchor_dims=DEFAULT_ANCHORS):
        self.rpn_model = rpn_model
        self.class_mapping = class_mapping
        self.preprocess_func = preprocess_func
        self.num_rois = num_rois
        self.stride = stride
        self.anchor_dims = anchor_dims
        self._cache = {}
        self.conv_only = True if len(rpn_model.output) == 3 else False
        self._stream = None

    def _own_stream(self, probe=True):
        """The manager's device work (RPN forward, decode, ordering, NMS, RoI -> truth, the copies back to numpy) runs on a
        stream of its own.  In the detector loops (train_util._train_detector) it is called while the PREVIOUS detector step
        is still enqueued on the caller's stream; on that stream the `.cpu()` at the end of this work would make the host wait
        for the whole step before it could stage the next image.  The RPN model here is never the model being
        trained (steps 2 / 4 freeze it: train_det_step2.py / step4), and a detector trainer writes only its own master
        and packed buffers, so nothing this stream reads is written by the step beside it."""
        if self._stream is None:
            from . import feed
            # the PROBED stream (feed.manager_stream: ~70 ms once, keeps the trainer's streams alive) only where a training step runs
            # beside this work; an inference-only process (voc_dets through get_det_inputs) takes a plain stream of its own
            self._stream = feed.manager_stream() if probe else torch.cuda.Stream()
            self._stream_probed = probe
        elif probe and not getattr(self, "_stream_probed", False):
            from . import feed
            self._stream.synchronize()
            self._stream, self._stream_probed = feed.manager_stream(), True
        if getattr(self.rpn_model, "_trainer", None) is not None and getattr(self.rpn_model, "_dirty", False):
            self._stream.wait_stream(torch.cuda.current_stream())       # (a trained RPN model: order behind its last step)
        return torch.cuda.stream(self._stream)

    def batched_image(self, image):
        return np.expand_dims(self.preprocess_func(image.data), axis=0)

    # ------------------------------------------------------------------ device path
    def _proposals_dev(self, image, pre_nms, max_boxes, x=None):
        """RPN forward + decode + valid filter + descending score order + int16 cast + NMS
        (det_util.py:44-77 / 145-156), all on the device.  ``x``: the preprocessed image already on the device (fast path).
        Returns (rois int16 device (n,4), conv feature map device or None)."""
        if x is None:
            x = nets.to_device_image(self.batched_image(image))
        cls, reg, feat = self.rpn_model.forward_dev(x)
        rois_all, valid = ops.decode_proposals(reg, np.asarray(self.anchor_dims) // self.stride)
        scores = cls.reshape(-1)
        K = min(pre_nms, ops.NMS_MAX_BOXES)
        order, n = ops.topk_order(scores, valid, K)
        cand, _ = ops.gather_candidates(rois_all, scores, order, n, K)
        keep, n_keep = ops.nms_sorted(cand, n, 0.7, max_boxes)
        nk = int(n_keep.item())
        rois = cand[keep[:nk].long()]
        return rois, (feat if self.conv_only else None)

    def _process(self, image):
        with self._own_stream():
            rois, feat = self._proposals_dev(image, 12000, 2000)
            filtered_rois, y_class_num, y_transform = _rois_to_truth(rois, image, self.class_mapping, stride=self.stride)
            cache_obj = {"rois": filtered_rois, "y_class_num": y_class_num, "y_transform": y_transform}
            if feat is not None:
                cache_obj["conv_out"] = feat.cpu().numpy()
        self._cache[image.cache_key] = cache_obj

    # ------------------------------------------------------------------ device-resident fast path (train_util._train_detector)
    # get_training_input with the LARGE tensors left on the device.  The reference's loop (train_util.py:100-118) builds the float64
    # image on the host for the RPN pass (det_util.py:36), and again for the detector step (:127) -- or, with conv_only (step 4),
    # copies the 9.8 MB conv4 map to numpy (:56) and train_on_batch uploads it again.  Here the decoded uint8 frame is uploaded ONCE,
    # resized / preprocessed on the device (feed.device_image: the same float32 bits) and serves both the RPN pass and the step;
    # the conv map never leaves the device.  The small arrays (<= 2 000 RoIs, their one-hot classes and targets) stay host numpy:
    # the reference's np.random sampling picks among them (det_util.py:260-306) exactly as before.  ``prefetch`` holds everything
    # that needs no random draw and may run one image AHEAD, beside the current detector step, on the manager's own stream.
    def decode_ahead(self, image):
        """The RPN loop's decode-ahead thread (rpn_util.RpnTrainingManager.decode_ahead) does not pay here: a detector step is 1.8 ms of
        device time, the inline decode hides in its slack (1.785 ms per iteration from files, 1.793 from memory), and a second thread taking
        the interpreter lock for its header parsing and byte copies costs the loop's thread 0.2 ms (2.007).  FRCNN_DET_DECODE_AHEAD=1: on."""
        import os
        from . import feed
        if os.environ.get("FRCNN_DET_DECODE_AHEAD", "0") != "0" and feed.device_preprocess(self.preprocess_func):
            feed.decode_ahead(image)

    def prefetch(self, image):
        from . import feed
        key = image.cache_key
        pre = self.__dict__.setdefault("_pre", {})
        if key in pre:
            return
        with self._own_stream():
            have = key in self._cache
            x = None if (have and self.conv_only) else feed.device_image(image, self.preprocess_func)
            feat = None
            if not have:
                rois, feat = self._proposals_dev(image, 12000, 2000, x=x)
                filtered_rois, y_class_num, y_transform = _rois_to_truth(rois, image, self.class_mapping, stride=self.stride)
                self._cache[key] = {"rois": filtered_rois, "y_class_num": y_class_num, "y_transform": y_transform}
            first = feat if self.conv_only else x
            if first is not None:
                feed.Ready.mark(first)
            pre[key] = first

    def get_training_input_dev(self, image):
        """get_training_input whose first element is a float32 DEVICE tensor ((1,H,W,3) image, or the (1,R,C,Cf) conv map when
        conv_only); rois / y_class_num / y_transform are the reference's numpy arrays."""
        self.prefetch(image)
        first = self._pre.pop(image.cache_key)
        results = self._cache[image.cache_key]
        if self.conv_only and first is None:                 # (a conv_only entry is consumed by its first use, det_util.py:129-130)
            conv = results.get("conv_out")
            first = None if conv is None else nets.to_device_image(conv)
        if len(results["rois"]) == 0:
            return None, None, None, None                    # (the entry stays cached, as in the reference: det_util.py:103-104)
        rois, y_class_num, y_transform = results["rois"], results["y_class_num"], results["y_transform"]
        sampled_idxs = _get_det_samples(y_class_num[:, -1] == 0, self.num_rois)
        rois, y_class_num, y_transform = rois[sampled_idxs], y_class_num[sampled_idxs], y_transform[sampled_idxs]
        if self.conv_only:
            del self._cache[image.cache_key]
        if first.dim() == 3:
            ready = getattr(first, "_ready", None)
            first = first.unsqueeze(0)
            if ready is not None:
                first._ready = ready                         # (a new tensor object: the event the step's streams wait for travels with it)
        return first, np.expand_dims(rois, axis=0), np.expand_dims(y_class_num, axis=0), np.expand_dims(y_transform, axis=0)

    def get_training_input(self, image):
        """det_util.py:90-133."""
        if image.cache_key not in self._cache:
            self._process(image)
        results = self._cache[image.cache_key]
        if len(results["rois"]) == 0:
            return None, None, None, None
        rois, y_class_num, y_transform = results["rois"], results["y_class_num"], results["y_transform"]
        found_object = y_class_num[:, -1] == 0
        sampled_idxs = _get_det_samples(found_object, self.num_rois)
        rois, y_class_num, y_transform = rois[sampled_idxs], y_class_num[sampled_idxs], y_transform[sampled_idxs]
        first_input = results["conv_out"] if self.conv_only else self.batched_image(image)
        if self.conv_only:
            del self._cache[image.cache_key]
        return first_input, np.expand_dims(rois, axis=0), np.expand_dims(y_class_num, axis=0), np.expand_dims(y_transform, axis=0)

    def get_det_inputs(self, image):
        """det_util.py:136-158: (conv feature map (1,R,C,Cf) or None, nms_rois (n,4) int16)."""
        with self._own_stream(probe=False):
            rois, feat = self._proposals_dev(image, 8000, 300)
            # (a bf16 base hands its map over as float32 -- numpy has no bf16, the widening is exact and the detector narrows it back)
            return (feat.float().cpu().numpy() if feat is not None else None), rois.cpu().numpy()


def _get_anchor_coords(conv_rows, conv_cols, anchor_dims, multiplier=1):
    """det_util.py:162-175."""
    return ops.anchors_conv(conv_rows, conv_cols, np.asarray(anchor_dims) * multiplier).cpu().numpy()


def _sanitize_boxes_inplace(conv_cols, conv_rows, coords):
    """det_util.py:179-192 (host; the device decode fuses this step, see _get_rois)."""
    coords[:, 2] = np.maximum(coords[:, 0] + 1, coords[:, 2])
    coords[:, 3] = np.maximum(coords[:, 1] + 1, coords[:, 3])
    coords[:, 0] = np.maximum(0, coords[:, 0])
    coords[:, 1] = np.maximum(0, coords[:, 1])
    coords[:, 2] = np.minimum(conv_cols - 1, coords[:, 2])
    coords[:, 3] = np.minimum(conv_rows - 1, coords[:, 3])
    return coords


def _get_valid_box_idxs(boxes):
    """det_util.py:196-205."""
    return np.where((boxes[:, 2] > boxes[:, 0]) & (boxes[:, 3] > boxes[:, 1]))[0]


def nms(boxes, probs, overlap_thresh=0.7, max_boxes=300):
    """det_util.nms (det_util.py:209-256).  boxes: int16 or float (n,4); probs (n,).
    Returns (boxes[pick], probs[pick]) in pick order; ``[]`` for empty input like the reference
    (:220-221).  Ties in ``probs`` (unspecified in the reference) resolve to the lower index."""
    if len(boxes) == 0:
        return []
    boxes = np.asarray(boxes)
    probs = np.asarray(probs)
    if len(boxes) > ops.NMS_MAX_BOXES:
        raise ValueError("nms: at most %d boxes per call" % ops.NMS_MAX_BOXES)
    n = len(boxes)
    order, cnt = ops.topk_order(torch.from_numpy(np.ascontiguousarray(probs, dtype=np.float32)).cuda(), None, n)
    order_h = order.cpu().numpy()
    sorted_boxes = boxes[order_h]
    if sorted_boxes.dtype != np.int16:
        sorted_boxes = sorted_boxes.astype(np.float64)
    keep, n_keep = ops.nms_sorted(torch.from_numpy(np.ascontiguousarray(sorted_boxes)).cuda(), cnt, overlap_thresh, max_boxes)
    pick = order_h[keep.cpu().numpy()[:int(n_keep.item())]]
    return boxes[pick], probs[pick]


def _get_det_samples(is_pos, num_desired_rois):
    """det_util.py:260-306 (host, np.random)."""
    desired_pos = num_desired_rois // 4
    pos_samples = np.where(is_pos)[0]
    neg_samples = np.where(np.logical_not(is_pos))[0]
    if len(pos_samples) == 0:
        selected_pos = []
    elif len(pos_samples) < desired_pos:
        selected_pos = pos_samples.tolist()
    else:
        selected_pos = np.random.choice(pos_samples, desired_pos, replace=False).tolist()
    desired_neg = num_desired_rois - len(selected_pos)
    if len(neg_samples) == 0:
        selected_neg = []
    elif len(neg_samples) < desired_neg:
        selected_neg = np.random.choice(neg_samples, desired_neg, replace=True).tolist()
    else:
        selected_neg = np.random.choice(neg_samples, desired_neg, replace=False).tolist()
    if len(selected_neg) == 0 and len(pos_samples) > 0:
        num_copies = desired_neg // len(pos_samples) + 1
        selected_neg = np.tile(pos_samples, num_copies)[:desired_neg].tolist()
    return selected_pos + selected_neg


def _rois_to_truth(rois, image, class_mapping, stride=16):
    """det_util.py:310-366.  rois: int16 (n,4) numpy or device tensor.
    Returns (eligible rois (E,4) int16, one-hot classes (E,C) int32, [labels | targets] (E,8(C-1)) f32)."""
    gt_boxes = [gt_box.resize(1 / stride) for gt_box in image.gt_boxes]
    C = len(class_mapping)
    rois_d = rois if isinstance(rois, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(rois, dtype=np.int16)).cuda()
    rois_h = rois_d.cpu().numpy()
    if len(gt_boxes) == 0 or len(rois_h) == 0:
        return rois_h[:0], np.zeros((0, C), np.int32), np.zeros((0, 8 * (C - 1)), np.float32)
    gt64 = np.array([b.corners for b in gt_boxes], dtype=np.float64)
    gt32 = get_bbox_coords(gt_boxes)
    gt_cls = np.array([class_mapping[b.obj_cls] for b in gt_boxes], dtype=np.int32)
    elig, cls, tg = ops.roi_targets(rois_d, torch.from_numpy(gt32).cuda(), torch.from_numpy(gt64).cuda(),
                                    torch.from_numpy(gt_cls).cuda(), class_mapping["bg"])
    elig, cls, tg = elig.cpu().numpy().astype(bool), cls.cpu().numpy()[...], tg.cpu().numpy()
    e_rois, e_cls, e_tg = rois_h[elig], cls[elig], tg[elig]
    E = len(e_rois)
    onehot = np.zeros((E, C), dtype=np.int32)
    onehot[np.arange(E), e_cls] = 1
    labels = np.zeros((E, 4 * (C - 1)), dtype=np.float32)
    targs = np.zeros((E, 4 * (C - 1)), dtype=np.float32)
    pos = np.nonzero(e_cls != class_mapping["bg"])[0]
    for k in range(4):
        labels[pos, 4 * e_cls[pos] + k] = 1
        targs[pos, 4 * e_cls[pos] + k] = e_tg[pos, k]
    return e_rois, onehot, np.concatenate([labels, targs], axis=1)


def _get_rois(regr_out, anchor_dims, stride):
    """det_util.py:370-380: RPN regression output (1,R,C,4A) -> sanitised proposals (N,4) f32."""
    reg = regr_out if isinstance(regr_out, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(regr_out, dtype=np.float32)).cuda()
    rois, _ = ops.decode_proposals(reg, np.asarray(anchor_dims) // stride)
    return rois.cpu().numpy()
