"""numpy restatement of the reference's host-side (numpy) half of the hot path.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Every function cites the
reference lines it restates (paths relative to /root/reference/faster_rcnn/).
PINNED: tests/test_oracle_golden.py checks every function here against golden
vectors captured from the imported reference (tests/golden/make_golden.py).

The restatement is written array-at-a-time (broadcast over the GT axis, boolean
masks instead of index lists) so it is an independent statement of the
algorithm, but every arithmetic step keeps the reference's dtype and operation
order so results are bit-identical.
"""
import math
import random

import numpy as np

BBREG_MULTIPLIERS = np.array([10, 10, 5, 5], dtype=np.float32)   # shared_constants.py:5
ANCHOR_RATIOS = ((1, 1), (1, 2), (2, 1))                          # shared_constants.py:8 (h, w)
POS_OVERLAP, NEG_OVERLAP = 0.7, 0.3                               # rpn_util.py:11-12
SAMPLE_SIZE, MAX_POS_SAMPLES = 256, 128                           # rpn_util.py:14-15
CLASSIFIER_MIN_OVERLAP, CLASSIFIER_POS_OVERLAP = 0.1, 0.5         # det_util.py:7-8


# --------------------------------------------------------------------------- anchors
def get_anchors(scales, ratios=ANCHOR_RATIOS):
    """util.py:242-253.  (A,2) int64 rows [height, width], scale-major."""
    rows = []
    for s in scales:
        for rh, rw in ratios:
            ratio = math.sqrt(s * rh * s * rw) / s
            rows.append([(s * rh) // ratio, (s * rw) // ratio])
    return np.array(rows).astype(int)


def conv_dims_resnet(height, width):
    """resnet.py:78-93: +6 pad then four k/2 VALID reductions with k = 7,3,1,1."""
    out = []
    for d in (height, width):
        d += 6
        for k in (7, 3, 1, 1):
            d = (d - k) // 2 + 1
        out.append(d)
    return out


def conv_dims_vgg(height, width):
    """vgg.py:60-61."""
    return height // 16, width // 16


def anchors_image(rows, cols, anchor_hw, stride):
    """rpn_util.py:276-298 (+160-166, 184-189).  (N,4) f32 [x1,y1,x2,y2] in image
    pixels; flat index i = (y*cols + x)*A + a; centre = int32(stride*(x+.5))."""
    anchor_hw = np.asarray(anchor_hw)
    A = len(anchor_hw)
    yy, xx, aa = np.meshgrid(np.arange(rows), np.arange(cols), np.arange(A), indexing="ij")
    cx = (stride * (xx.reshape(-1) + 0.5)).astype("int32")
    cy = (stride * (yy.reshape(-1) + 0.5)).astype("int32")
    h = anchor_hw[aa.reshape(-1), 0]
    w = anchor_hw[aa.reshape(-1), 1]
    out = np.zeros((rows * cols * A, 4), dtype=np.float32)
    out[:, 0] = cx - w // 2
    out[:, 1] = cy - h // 2
    out[:, 2] = out[:, 0] + w
    out[:, 3] = out[:, 1] + h
    return out


def oob_mask(anchors, img_w, img_h):
    """rpn_util.py:302-310 as a boolean mask."""
    return (anchors[:, 0] < 0) | (anchors[:, 1] < 0) | (anchors[:, 2] >= img_w) | (anchors[:, 3] >= img_h)


def anchors_conv(rows, cols, anchor_hw_conv):
    """det_util.py:162-175.  (rows, cols, A, 4) f32 in conv-cell units, centre = cell index."""
    anchor_hw_conv = np.asarray(anchor_hw_conv)
    out = np.zeros((rows, cols, len(anchor_hw_conv), 4), dtype=np.float32)
    xs = np.arange(cols)[None, :]
    ys = np.arange(rows)[:, None]
    for a, (h, w) in enumerate(anchor_hw_conv):
        out[:, :, a, 0] = xs - w // 2
        out[:, :, a, 1] = ys - h // 2
        out[:, :, a, 2] = out[:, :, a, 0] + w
        out[:, :, a, 3] = out[:, :, a, 1] + h
    return out


# --------------------------------------------------------------------------- IoU
def cross_ious(boxes1, boxes2):
    """util.py:146-177.  (M,G) f32, no +1 convention.  Operation order per element:
    area1*, area2*, max/min, max(0,.), w*h, (area1+area2)-inter, inter/union."""
    boxes1 = np.asarray(boxes1)
    boxes2 = np.asarray(boxes2)
    res = np.zeros((len(boxes1), len(boxes2)), dtype=np.float32)
    if len(boxes2) == 0 or len(boxes1) == 0:
        return res
    a1 = (boxes1[:, 2] - boxes1[:, 0]) * (boxes1[:, 3] - boxes1[:, 1])
    a2 = (boxes2[:, 2] - boxes2[:, 0]) * (boxes2[:, 3] - boxes2[:, 1])
    ix1 = np.maximum(boxes1[:, None, 0], boxes2[None, :, 0])
    iy1 = np.maximum(boxes1[:, None, 1], boxes2[None, :, 1])
    ix2 = np.minimum(boxes1[:, None, 2], boxes2[None, :, 2])
    iy2 = np.minimum(boxes1[:, None, 3], boxes2[None, :, 3])
    iw = np.maximum(0, ix2 - ix1)
    ih = np.maximum(0, iy2 - iy1)
    inter = iw * ih
    union = a1[:, None] + a2[None, :] - inter
    with np.errstate(divide="ignore", invalid="ignore"):
        res[:, :] = inter / union
    return res


def reg_params_f64(anchor_xyxy_int, gt_xyxy, gt_is_f32=True):
    """util.py:180-206 as evaluated by its two callers.  Anchor/RoI coords are integers
    (np.int64 at rpn_util.py:91, np.int16 at det_util.py:349) so their centre is f64.
    * rpn_util.py:91 passes a row of the f32 GT array: the GT centre SUM and the GT
      width/height are formed in f32 (halving is exact), then promoted to f64.
    * det_util.py:349 passes ``gt_box.corners`` (python floats -> f64): all f64."""
    a = np.asarray(anchor_xyxy_int).astype(np.int64)
    if gt_is_f32:
        g = np.asarray(gt_xyxy, dtype=np.float32)
    else:
        g = np.asarray(gt_xyxy, dtype=np.float64)
    gcx = ((g[..., 2] + g[..., 0]) / g.dtype.type(2.0)).astype(np.float64)
    gcy = ((g[..., 3] + g[..., 1]) / g.dtype.type(2.0)).astype(np.float64)
    gw = (g[..., 2] - g[..., 0]).astype(np.float64)
    gh = (g[..., 3] - g[..., 1]).astype(np.float64)
    acx = (a[..., 2] + a[..., 0]) / 2.0
    acy = (a[..., 3] + a[..., 1]) / 2.0
    aw = (a[..., 2] - a[..., 0]).astype(np.float64)
    ah = (a[..., 3] - a[..., 1]).astype(np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        tx = (gcx - acx) / aw
        ty = (gcy - acy) / ah
        tw = np.log(gw / aw)
        th = np.log(gh / ah)
    return np.stack([tx, ty, tw, th], axis=-1)


# --------------------------------------------------------------------------- RPN targets
def rpn_assign(gt_f32, rows, cols, anchor_hw, stride, img_w, img_h):
    """rpn_util.py:54-103 (pre-sampling).  Returns can_use (N,) bool, is_pos (N,) bool,
    bbreg_targets (N,4) f32, argmax_gt (N,) int64."""
    gt = np.asarray(gt_f32, dtype=np.float32).reshape(-1, 4)
    anchors = anchors_image(rows, cols, anchor_hw, stride)
    n = len(anchors)
    can_use = np.zeros(n, dtype=bool)
    is_pos = np.zeros(n, dtype=bool)
    bbreg = np.zeros((n, 4), dtype=np.float32)
    ious = cross_ious(anchors, gt)
    max_by_anchor = ious.max(axis=1)
    arg_by_anchor = ious.argmax(axis=1)
    max_by_gt = ious.max(axis=0)
    arg_by_gt = ious.argmax(axis=0)
    is_pos |= max_by_anchor > POS_OVERLAP
    is_pos[arg_by_gt[max_by_gt > 0.0]] = True
    can_use |= is_pos
    pos = np.nonzero(is_pos)[0]
    if len(pos):
        a_int = anchors[pos].astype(np.int64)        # anchors are integer valued
        t = reg_params_f64(a_int, gt[arg_by_anchor[pos]])
        bbreg[pos] = BBREG_MULTIPLIERS * t            # f32 * f64 -> f64, stored to f32
    neg = (~is_pos) & (max_by_anchor < NEG_OVERLAP)
    can_use |= neg
    can_use[oob_mask(anchors, img_w, img_h)] = False
    return can_use, is_pos, bbreg, arg_by_anchor


def apply_sampling(is_pos, can_use):
    """rpn_util.py:324-350.  Uses the Python ``random`` module exactly as the reference
    does (host-side RNG parity); mutates and returns can_use."""
    pos_locs = np.nonzero(is_pos & can_use)[0]
    neg_locs = np.nonzero(~is_pos & can_use)[0]
    n_pos, n_neg = len(pos_locs), len(neg_locs)
    if n_pos > MAX_POS_SAMPLES:
        off = random.sample(range(n_pos), n_pos - MAX_POS_SAMPLES)
        can_use[pos_locs[off]] = 0
        n_pos = MAX_POS_SAMPLES
    if n_neg + n_pos > SAMPLE_SIZE:
        off = random.sample(range(n_neg), n_neg + n_pos - SAMPLE_SIZE)
        can_use[neg_locs[off]] = 0
    return can_use


def rpn_pack(can_use, is_pos, bbreg, rows, cols, A):
    """rpn_util.py:126-140.  y_class (1,R,C,2A) bool, y_bbreg (1,R,C,8A) f32."""
    cu = can_use.reshape(rows, cols, A)
    ip = is_pos.reshape(rows, cols, A)
    y_class = np.concatenate([cu, ip], axis=2)[None]
    sel = np.repeat(cu & ip, 4, axis=2)
    y_bbreg = np.concatenate([sel, bbreg.reshape(rows, cols, 4 * A)], axis=2)[None]
    return y_class, y_bbreg


# --------------------------------------------------------------------------- proposals
def decode_f32(coords, deltas):
    """util.py:111-142 (transform_np_inplace) on a copy.  All f32, np.round half-even."""
    c = np.array(coords, dtype=np.float32, copy=True)
    d = np.asarray(deltas)
    c[:, 2] -= c[:, 0]
    c[:, 3] -= c[:, 1]
    c[:, 0] += c[:, 2] / 2
    c[:, 1] += c[:, 3] / 2
    c[:, 0] += d[:, 0] * c[:, 2]
    c[:, 1] += d[:, 1] * c[:, 3]
    c[:, 2] *= np.exp(d[:, 2])
    c[:, 3] *= np.exp(d[:, 3])
    c[:, 0] -= c[:, 2] / 2
    c[:, 1] -= c[:, 3] / 2
    np.round(c, out=c)
    c[:, 2] += c[:, 0]
    c[:, 3] += c[:, 1]
    return c


def decode_preround(coords, deltas):
    """Same as decode_f32 but stops before np.round: used by tests to find coordinates
    whose fractional part sits on a .5 rounding boundary (libm-sensitive)."""
    c = np.array(coords, dtype=np.float32, copy=True)
    d = np.asarray(deltas)
    c[:, 2] -= c[:, 0]
    c[:, 3] -= c[:, 1]
    c[:, 0] += c[:, 2] / 2
    c[:, 1] += c[:, 3] / 2
    c[:, 0] += d[:, 0] * c[:, 2]
    c[:, 1] += d[:, 1] * c[:, 3]
    c[:, 2] *= np.exp(d[:, 2])
    c[:, 3] *= np.exp(d[:, 3])
    c[:, 0] -= c[:, 2] / 2
    c[:, 1] -= c[:, 3] / 2
    return c


def sanitize(coords, conv_cols, conv_rows):
    """det_util.py:179-192 on a copy (order of the clamps matters)."""
    c = np.array(coords, copy=True)
    c[:, 2] = np.maximum(c[:, 0] + 1, c[:, 2])
    c[:, 3] = np.maximum(c[:, 1] + 1, c[:, 3])
    c[:, 0] = np.maximum(0, c[:, 0])
    c[:, 1] = np.maximum(0, c[:, 1])
    c[:, 2] = np.minimum(conv_cols - 1, c[:, 2])
    c[:, 3] = np.minimum(conv_rows - 1, c[:, 3])
    return c


def valid_mask(boxes):
    """det_util.py:196-205 as a mask."""
    return (boxes[:, 2] > boxes[:, 0]) & (boxes[:, 3] > boxes[:, 1])


def get_rois(regr_out, anchor_hw, stride):
    """det_util.py:370-380.  regr_out (1,R,C,4A) f32 -> (N,4) f32 sanitized proposals."""
    rows, cols = regr_out.shape[1:3]
    anc = anchors_conv(rows, cols, np.asarray(anchor_hw) // stride).reshape(-1, 4)
    deltas = regr_out[0].reshape(-1, 4) / BBREG_MULTIPLIERS
    return sanitize(decode_f32(anc, deltas), cols, rows)


def score_order(probs, k):
    """det_util.py:71-74 / 151-154 with the build's tie rule: descending
    score, ascending original index on ties."""
    return np.argsort(-np.asarray(probs), kind="stable")[:k]


def nms(boxes, probs, overlap_thresh=0.7, max_boxes=300):
    """det_util.py:209-256.  Greedy NMS, +1 pixel convention, keep overlap <= thresh,
    stop at max_boxes.  Returns (boxes[pick], probs[pick], pick)."""
    boxes = np.asarray(boxes)
    probs = np.asarray(probs)
    if len(boxes) == 0:
        return [], [], []
    x1, y1, x2, y2 = boxes[:, 0], boxes[:, 1], boxes[:, 2], boxes[:, 3]
    area = (x2 - x1 + 1) * (y2 - y1 + 1)
    remaining = np.argsort(probs)[::-1]          # best first
    pick = []
    while len(remaining):
        i = remaining[0]
        rest = remaining[1:]
        pick.append(i)
        w = np.maximum(0, np.minimum(x2[i], x2[rest]) - np.maximum(x1[i], x1[rest]) + 1)
        h = np.maximum(0, np.minimum(y2[i], y2[rest]) - np.maximum(y1[i], y1[rest]) + 1)
        inter = w * h
        overlap = inter / (area[i] + area[rest] - inter)
        remaining = rest[overlap <= overlap_thresh]
        if len(pick) >= max_boxes:
            break
    pick = np.array(pick, dtype=np.int64)
    return boxes[pick], probs[pick], pick


def proposals(regr_out, cls_out, anchor_hw, stride, pre_nms_top_n, max_boxes, thresh=0.7):
    """det_util.py:63-77 (train: 12000/2000) and :145-156 (test: 8000/300).
    Returns int16 rois (n,4), their probs, and the sorted int16 candidates fed to NMS."""
    rois = get_rois(regr_out, anchor_hw, stride)
    probs = cls_out.reshape(-1)
    v = np.nonzero(valid_mask(rois))[0]
    rois, probs = rois[v], probs[v]
    order = score_order(probs, pre_nms_top_n)
    cand = rois[order].astype("int16")
    cprobs = probs[order]
    kept, kprobs, pick = nms(cand, cprobs, overlap_thresh=thresh, max_boxes=max_boxes)
    return kept, kprobs, cand, cprobs, pick


# --------------------------------------------------------------------------- detector targets
def rois_to_truth(rois_i16, gt_xyxy_img, gt_cls_idx, num_classes, stride=16):
    """det_util.py:310-366.  gt_xyxy_img: GT corners in image pixels (python/np numbers),
    scaled by 1/stride as Box.resize does (shapes.py Box.resize: each coord * ratio, kept
    as float) then stored f32 by get_bbox_coords (util.py:229-238).
    Returns eligible rois (E,4) int16, one-hot (E,C) int32, bbreg (E, 8(C-1)) f32."""
    ratio = 1 / stride
    gt64 = np.array([[c * ratio for c in box] for box in gt_xyxy_img], dtype=np.float64).reshape(-1, 4)
    gt = gt64.astype(np.float32)                 # get_bbox_coords stores f32 (util.py:235)
    rois = np.asarray(rois_i16)
    ious = cross_ious(rois, gt)
    max_iou = ious.max(axis=1)
    arg_gt = ious.argmax(axis=1)
    elig = max_iou >= CLASSIFIER_MIN_OVERLAP
    pos = (max_iou >= CLASSIFIER_POS_OVERLAP)[elig]
    e_rois = rois[elig]
    e_gt = arg_gt[elig]
    C = num_classes
    onehot = np.zeros((len(e_rois), C), dtype=np.int32)
    cls = np.where(pos, np.asarray(gt_cls_idx)[e_gt] if len(gt) else 0, C - 1)
    onehot[np.arange(len(e_rois)), cls] = 1
    labels = np.zeros((len(e_rois), 4 * (C - 1)), dtype=np.float32)
    targs = np.zeros((len(e_rois), 4 * (C - 1)), dtype=np.float32)
    pi = np.nonzero(pos)[0]
    if len(pi):
        t = reg_params_f64(e_rois[pi], gt64[e_gt[pi]], gt_is_f32=False).astype(np.float32)  # stored f32 (det_util.py:350)
        t *= BBREG_MULTIPLIERS                                           # then scaled in f32 (:351)
        for k in range(4):
            labels[pi, 4 * cls[pi] + k] = 1
            targs[pi, 4 * cls[pi] + k] = t[:, k]
    return e_rois, onehot, np.concatenate([labels, targs], axis=1)


def det_samples(is_pos, num_rois):
    """det_util.py:260-306.  Uses np.random exactly as the reference (host RNG parity)."""
    want_pos = num_rois // 4
    pos = np.nonzero(is_pos)[0]
    neg = np.nonzero(~np.asarray(is_pos))[0]
    if len(pos) == 0:
        sel_pos = []
    elif len(pos) < want_pos:
        sel_pos = pos.tolist()
    else:
        sel_pos = np.random.choice(pos, want_pos, replace=False).tolist()
    want_neg = num_rois - len(sel_pos)
    if len(neg) == 0:
        sel_neg = []
    elif len(neg) < want_neg:
        sel_neg = np.random.choice(neg, want_neg, replace=True).tolist()
    else:
        sel_neg = np.random.choice(neg, want_neg, replace=False).tolist()
    if len(sel_neg) == 0 and len(pos) > 0:
        copies = want_neg // len(pos) + 1
        sel_neg = np.tile(pos, copies)[:want_neg].tolist()
    return sel_pos + sel_neg


# --------------------------------------------------------------------------- detections
def transform_f64(anchor, reg):
    """util.py:55-74.  Scalar f64 decode with math.exp, no rounding."""
    x1, y1, x2, y2 = anchor
    cxa, cya = (x1 + x2) / 2, (y1 + y2) / 2
    wa, ha = x2 - x1, y2 - y1
    tx, ty, tw, th = reg
    cx = tx * wa + cxa
    cy = ty * ha + cya
    w = math.exp(tw) * wa
    h = math.exp(th) * ha
    x = cx - w / 2
    y = cy - h / 2
    return x, y, x + w, y + h


def transform_legacy(roi_f32, t_f32):
    """util.transform (util.py:55-74) as voc_dets.py:63-65 evaluates it -- np.float32 scalars for
    the RoI and the deltas -- under the reference's pinned numpy 1.13 (legacy value-based scalar
    promotion): f32+f32 stays f32, f32/int and python-float*f32 become f64.  Written with explicit
    casts so the result does not depend on the numpy version running the oracle."""
    x1, y1, x2, y2 = (np.float32(v) for v in roi_f32)
    tx, ty, tw, th = (np.float32(v) for v in t_f32)
    cxa, cya = np.float64(np.float32(x1 + x2)) / 2, np.float64(np.float32(y1 + y2)) / 2
    wa, ha = np.float32(x2 - x1), np.float32(y2 - y1)
    cx = np.float64(np.float32(tx * wa)) + cxa
    cy = np.float64(np.float32(ty * ha)) + cya
    w = math.exp(float(tw)) * np.float64(wa)
    h = math.exp(float(th)) * np.float64(ha)
    x, y = cx - w / 2, cy - h / 2
    return x, y, x + w, y + h


def detections(rois, out_cls, out_reg, bg_idx, resize_ratio, stride=16, det_threshold=0.0,
               num_rois=64, nms_thresh=0.5):
    """voc_dets.py:20-88 given the detector outputs for every scored RoI row.

    rois: (n,4) kept proposals.  out_cls (n_pad, C) / out_reg (n_pad, 4(C-1)): detector
    outputs for the padded RoI list (each last batch padded with copies of its first RoI,
    voc_dets.py:42-46) -- ``pad_rois`` builds that list; unpadded outputs (n rows) are accepted
    too (the padding rows only ever produce duplicates that the NMS removes).
    Returns a list of (cls_idx, prob f32, bbox int64[4]) in the reference's emission
    order: classes in first-seen order, boxes in NMS pick order."""
    padded = pad_rois(rois, num_rois)[:len(out_cls)].astype(np.float32)
    by_cls = {}
    for r in range(len(padded)):
        c = int(np.argmax(out_cls[r]))
        conf = out_cls[r, c]
        # np.float32 scalar < Python float: an f64 comparison under the reference's pinned numpy 1.13 (legacy promotion;
        # np.float32(0.7) < 0.7 is True there, False under NEP 50)
        if c == bg_idx or float(conf) < float(det_threshold):
            continue
        t = out_reg[r, 4 * c:4 * c + 4] / BBREG_MULTIPLIERS
        p = transform_legacy(padded[r], t)
        by_cls.setdefault(c, ([], []))
        by_cls[c][0].append([stride * p[0], stride * p[1], stride * p[2], stride * p[3]])
        by_cls[c][1].append(conf)
    dets = []
    for c, (bb, pp) in by_cls.items():
        kb, kp, _ = nms(np.array(bb), np.array(pp), overlap_thresh=nms_thresh, max_boxes=2000)
        for b, p in zip(kb, kp):
            dets.append((c, p, np.array([int(round(v / resize_ratio)) for v in b])))
    return dets


def pad_rois(rois, num_rois=64):
    """voc_dets.py:31-47: split into batches of num_rois; the last batch is padded with
    copies of ITS first RoI.  Returns the concatenated padded list."""
    rois = np.asarray(rois)
    n = len(rois)
    if n % num_rois == 0:
        return rois
    last = (n // num_rois) * num_rois
    extra = np.tile(rois[last], (num_rois - (n - last), 1))
    return np.concatenate([rois, extra])
