"""Device-tensor level wrappers over the C ABI (include/frcnn_hip.h).

torch tensors are only the container type (device memory + the current HIP stream); all
arithmetic happens inside libfrcnn_hip.so.  Every function takes/returns CUDA(=HIP) tensors
and is asynchronous on ``torch.cuda.current_stream()``.
"""
import ctypes
import os

import numpy as np
import torch

from . import _lib

NMS_MAX_BOXES = 12288


_GPU_SEEN = False


def _require_gpu():
    # (torch.cuda.is_available() re-reads the environment on every call: 2 us x ~170 calls per training step)
    global _GPU_SEEN
    if not _GPU_SEEN:
        if not torch.cuda.is_available():
            raise _lib.FrcnnError("no HIP device visible: the faster_rcnn_amd ops need an MI355X (no CPU fallback)")
        _GPU_SEEN = True


_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_RAW_DEVICE = getattr(torch._C, "_cuda_getDevice", None)


def _stream():
    """The current torch stream's HIP handle.  torch.cuda.current_stream() builds a Stream object through three layers of
    Python (8 us); a training step asks ~90 times, and its mixed-precision form is bound by the host's launch rate."""
    if _RAW_STREAM is not None and _RAW_DEVICE is not None:
        return ctypes.c_void_p(_RAW_STREAM(_RAW_DEVICE()))
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    """Device pointer of a tensor as the plain integer ctypes' c_void_p argtype accepts (None = NULL).  (Building a c_void_p
    object per argument cost ~0.2 us x ~470 pointers per training step, whose mixed-precision form is bound by this host path.)"""
    if t is None:
        return None
    assert t.is_cuda and t.is_contiguous(), "device-contiguous tensor required"
    return t.data_ptr()


def _anchor_arg(anchor_hw):
    a = np.ascontiguousarray(np.asarray(anchor_hw), dtype=np.int32)
    assert a.ndim == 2 and a.shape[1] == 2
    return a, a.ctypes.data_as(ctypes.c_void_p), a.shape[0]


def _dev(x, dtype):
    """numpy / tensor -> contiguous device tensor of dtype."""
    if isinstance(x, torch.Tensor):
        return x.to(device="cuda", dtype=dtype).contiguous()
    return torch.from_numpy(np.ascontiguousarray(x)).to(device="cuda", dtype=dtype)


def _ws(nbytes):
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device="cuda")


def preprocess_u8(img_u8, mean_bgr, out=None):
    """(H,W,3) uint8 BGR image (numpy or device tensor) -> (1,H,W,3) f32 device tensor = float64(img) - mean, cast to f32
    (resnet.preprocess followed by the network's f32 input cast, bit for bit).  ``out``: write into this tensor
    (a captured graph's static input) instead of a new one."""
    _require_gpu()
    t = img_u8 if isinstance(img_u8, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(img_u8, dtype=np.uint8))
    t = t.to(device="cuda").contiguous()
    assert t.dtype == torch.uint8 and t.dim() == 3 and t.shape[2] == 3
    if out is None:
        out = torch.empty((1,) + tuple(t.shape), dtype=torch.float32, device="cuda")
    else:
        assert out.dtype == torch.float32 and out.is_contiguous() and out.numel() == t.numel()
    mean = (ctypes.c_double * 3)(*[float(v) for v in mean_bgr])
    _lib.call("frcnn_preprocess_u8", _p(t), t.shape[0] * t.shape[1], mean, _p(out), _stream())
    return out


def preprocess_u8_canvas(img_u8, mean_bgr, out, offset=None):
    """``preprocess_u8`` into a canvas: img_u8 (h,w,3) uint8 device tensor, out (1,hc,wc,3) / (hc,wc,3) f32; the image sits at ``offset``
    (rows, cols) -- default (h & 1, w & 1), which makes conv1's SAME padding on an even canvas the image's own -- zeros outside
    (frcnn_preprocess_u8_canvas)."""
    _require_gpu()
    assert img_u8.is_cuda and img_u8.dtype == torch.uint8 and img_u8.dim() == 3 and img_u8.shape[2] == 3 and img_u8.is_contiguous()
    hc, wc = int(out.shape[-3]), int(out.shape[-2])
    assert out.dtype == torch.float32 and out.is_contiguous() and out.numel() == hc * wc * 3
    h, w = int(img_u8.shape[0]), int(img_u8.shape[1])
    oy, ox = (h & 1, w & 1) if offset is None else (int(offset[0]), int(offset[1]))
    mean = (ctypes.c_double * 3)(*[float(v) for v in mean_bgr])
    _lib.call("frcnn_preprocess_u8_canvas", _p(img_u8), h, w, hc, wc, oy, ox, mean, _p(out), _stream())
    return out


def zero_outside(x, true_hw):
    """x (n,hc,wc,C) f32 or bf16 canvas tensor, IN PLACE: zero every cell at or beyond image i's true extent true_hw[i] = (rows, cols)
    (device int32 (n,2); frcnn_zero_outside).  Zeros cannot raise a magnitude bound: x keeps its record."""
    _require_gpu()
    assert x.dim() == 4 and x.is_contiguous() and true_hw.dtype == torch.int32 and true_hw.is_cuda and true_hw.numel() >= 2 * x.shape[0]
    n, hc, wc, c = (int(v) for v in x.shape)
    _lib.call("frcnn_zero_outside", _p(x), n, hc, wc, c * x.element_size(), _p(true_hw), _stream())
    return x


def decode_proposals_canvas(regr, anchor_hw_conv, true_rc):
    """``decode_proposals`` on canvas-shaped RPN outputs: true_rc = device int32 (2,) = the image's true (rows, cols)."""
    _require_gpu()
    keep, ap, A = _anchor_arg(anchor_hw_conv)
    regr = regr.reshape(regr.shape[-3], regr.shape[-2], regr.shape[-1]).contiguous()
    rows, cols = regr.shape[0], regr.shape[1]
    assert regr.shape[2] == 4 * A and true_rc.dtype == torch.int32 and true_rc.is_cuda and true_rc.numel() >= 2
    n = rows * cols * A
    rois = torch.empty((n, 4), dtype=torch.float32, device="cuda")
    valid = torch.empty(n, dtype=torch.uint8, device="cuda")
    _lib.call("frcnn_decode_proposals_canvas", _p(regr), rows, cols, ap, A, _p(true_rc), _p(rois), _p(valid), _stream())
    return rois, valid


_TAPS = {}


def resize_cubic_taps(dst, src):
    """One axis' INTER_CUBIC tap table [dst][8] int32 = {4 clamped source indices, 4 fixed-point weights} (host array;
    frcnn_resize_cubic_taps: OpenCV's f32 arithmetic).  Cached: a dataset has a handful of (dst, src) pairs."""
    key = (int(dst), int(src))
    tab = _TAPS.get(key)
    if tab is None:
        tab = np.empty((key[0], 8), dtype=np.int32)
        _lib.call("frcnn_resize_cubic_taps", key[0], key[1], tab.ctypes.data_as(ctypes.c_void_p))
        if len(_TAPS) > 512:
            _TAPS.clear()
        _TAPS[key] = tab
    return tab


def resize_cubic_u8(src, dst_h, dst_w, flip=False, tabs=None, out=None):
    """cv2.resize(src, (dst_w, dst_h), interpolation=cv2.INTER_CUBIC) [+ horizontal flip] of a (h,w,3) uint8 device image
    (shapes.Image.data, shapes.py:19-29) -> (dst_h, dst_w, 3) uint8 device tensor.  ``flip``: bool, or the C ABI's bit set (1 = horizontal
    flip, 2 = the source is RGB: write B, G, R).  ``tabs``: (tab_x, tab_y) DEVICE int32
    tables to reuse (a captured pass keeps its own); ``out``: write into this tensor."""
    _require_gpu()
    assert src.is_cuda and src.dtype == torch.uint8 and src.dim() == 3 and src.shape[2] == 3 and src.is_contiguous()
    sh, sw = int(src.shape[0]), int(src.shape[1])
    if tabs is None:
        tabs = (torch.from_numpy(resize_cubic_taps(dst_w, sw)).cuda(), torch.from_numpy(resize_cubic_taps(dst_h, sh)).cuda())
    if out is None:
        out = torch.empty((dst_h, dst_w, 3), dtype=torch.uint8, device="cuda")
    _lib.call("frcnn_resize_cubic_u8", _p(src), sh, sw, _p(tabs[0]), _p(tabs[1]), int(dst_h), int(dst_w), int(flip), _p(out), _stream())
    return out


# ----------------------------------------------------------------------------- anchors
def anchors_image(rows, cols, anchor_hw, stride):
    _require_gpu()
    keep, ap, A = _anchor_arg(anchor_hw)
    out = torch.empty((rows * cols * A, 4), dtype=torch.float32, device="cuda")
    _lib.call("frcnn_anchors_image", rows, cols, ap, A, stride, _p(out), _stream())
    return out


def anchors_conv(rows, cols, anchor_hw_conv):
    _require_gpu()
    keep, ap, A = _anchor_arg(anchor_hw_conv)
    out = torch.empty((rows, cols, A, 4), dtype=torch.float32, device="cuda")
    _lib.call("frcnn_anchors_conv", rows, cols, ap, A, _p(out), _stream())
    return out


# ----------------------------------------------------------------------------- IoU
def cross_ious(boxes1, boxes2):
    """boxes1: (M,4) f32 or int16 device tensor; boxes2: (G,4) f32.  -> (M,G) f32."""
    _require_gpu()
    M, G = boxes1.shape[0], boxes2.shape[0]
    out = torch.zeros((M, G), dtype=torch.float32, device="cuda")
    b2 = boxes2.to(torch.float32).contiguous()
    if boxes1.dtype == torch.int16:
        _lib.call("frcnn_cross_ious_i16", _p(boxes1.contiguous()), M, _p(b2), G, _p(out), _stream())
    else:
        _lib.call("frcnn_cross_ious_f32", _p(boxes1.to(torch.float32).contiguous()), M, _p(b2), G, _p(out), _stream())
    return out


# ----------------------------------------------------------------------------- RPN targets
def rpn_assign(rows, cols, anchor_hw, stride, gt, img_w, img_h):
    """-> can_use (N,) u8, is_pos (N,) u8, bbreg (N,4) f32, argmax_gt (N,) i32 (device)."""
    _require_gpu()
    keep, ap, A = _anchor_arg(anchor_hw)
    gt = _dev(gt, torch.float32).reshape(-1, 4)
    G = gt.shape[0]
    n = rows * cols * A
    can_use = torch.empty(n, dtype=torch.uint8, device="cuda")
    is_pos = torch.empty(n, dtype=torch.uint8, device="cuda")
    bbreg = torch.empty((n, 4), dtype=torch.float32, device="cuda")
    argmax = torch.empty(n, dtype=torch.int32, device="cuda")
    nbytes = _lib.load().frcnn_rpn_assign_workspace_bytes(rows, cols, A, G)
    ws = _ws(nbytes)
    _lib.call("frcnn_rpn_assign", rows, cols, ap, A, stride, _p(gt) if G else ctypes.c_void_p(0), G, int(img_w), int(img_h),
              _p(can_use), _p(is_pos), _p(bbreg), _p(argmax), _p(ws), ws.numel(), _stream())
    return can_use, is_pos, bbreg, argmax


def rpn_sample_lists(can_use, is_pos):
    """-> (pos_locs (N,) i32, neg_locs (N,) i32, counts (2,) i32): np.where(is_pos & can_use) / np.where(~is_pos & can_use) in
    ascending order, filled from the front, and their lengths (frcnn_rpn_sample_lists)."""
    _require_gpu()
    n = can_use.numel()
    pos = torch.empty(n, dtype=torch.int32, device="cuda")
    neg = torch.empty(n, dtype=torch.int32, device="cuda")
    counts = torch.empty(2, dtype=torch.int32, device="cuda")
    _lib.call("frcnn_rpn_sample_lists", _p(can_use), _p(is_pos), n, _p(pos), _p(neg), _p(counts), _stream())
    return pos, neg, counts


def rpn_pack_targets(can_use, is_pos, bbreg, cells, A, pos_locs, n_pos, off_pos, neg_locs, n_neg, off_neg):
    """Switch off the sampled positions (device int32 lists or None) IN can_use, then -> (y_class (cells,2A) f32, y_bbreg
    (cells,8A) f32) laid out like rpn_y_true's outputs (frcnn_rpn_pack_targets)."""
    _require_gpu()
    yc = torch.empty((cells, 2 * A), dtype=torch.float32, device="cuda")
    yb = torch.empty((cells, 8 * A), dtype=torch.float32, device="cuda")
    _lib.call("frcnn_rpn_pack_targets", _p(can_use), _p(is_pos), _p(bbreg), cells, A,
              _p(pos_locs), int(n_pos), _p(off_pos), 0 if off_pos is None else off_pos.numel(),
              _p(neg_locs), int(n_neg), _p(off_neg), 0 if off_neg is None else off_neg.numel(), _p(yc), _p(yb), _stream())
    return yc, yb


# ----------------------------------------------------------------------------- proposals
def decode_proposals(regr, anchor_hw_conv):
    """regr: (1,R,C,4A) or (R,C,4A) f32 device tensor -> rois (N,4) f32, valid (N,) u8."""
    _require_gpu()
    keep, ap, A = _anchor_arg(anchor_hw_conv)
    regr = regr.reshape(regr.shape[-3], regr.shape[-2], regr.shape[-1]).contiguous()
    rows, cols = regr.shape[0], regr.shape[1]
    assert regr.shape[2] == 4 * A
    n = rows * cols * A
    rois = torch.empty((n, 4), dtype=torch.float32, device="cuda")
    valid = torch.empty(n, dtype=torch.uint8, device="cuda")
    _lib.call("frcnn_decode_proposals", _p(regr), rows, cols, ap, A, _p(rois), _p(valid), _stream())
    return rois, valid


def transform_inplace(coords, deltas):
    _require_gpu()
    assert coords.dtype == torch.float32 and deltas.dtype == torch.float32
    _lib.call("frcnn_transform_inplace", _p(coords), _p(deltas.contiguous()), coords.shape[0], _stream())
    return coords


def topk_order(scores, valid, K):
    """-> order (K,) i32 (entries >= n are -1), n (1,) i32 device."""
    _require_gpu()
    N = scores.numel()
    order = torch.empty(K, dtype=torch.int32, device="cuda")
    n_out = torch.empty(1, dtype=torch.int32, device="cuda")
    ws = _ws(_lib.load().frcnn_topk_workspace_bytes(N))
    _lib.call("frcnn_topk_order", _p(scores.contiguous()), _p(valid), N, K, _p(order), _p(n_out), _p(ws), ws.numel(), _stream())
    return order, n_out


def gather_candidates(rois, scores, order, n, K):
    _require_gpu()
    cand = torch.empty((K, 4), dtype=torch.int16, device="cuda")
    cs = torch.empty(K, dtype=torch.float32, device="cuda")
    _lib.call("frcnn_gather_candidates", _p(rois), _p(scores.contiguous()), _p(order), _p(n), K, _p(cand), _p(cs), _stream())
    return cand, cs


def nms_sorted(boxes, n, thresh, max_boxes):
    """boxes: (K,4) int16 or float64, already in descending-score order; n: (1,) i32 device.
    -> keep (max_boxes,) i32 positions, n_keep (1,) i32."""
    _require_gpu()
    K = boxes.shape[0]
    keep = torch.empty((max_boxes,), dtype=torch.int32, device="cuda")       # the call writes every slot (-1 past n_keep)
    n_keep = torch.empty(1, dtype=torch.int32, device="cuda")
    ws = _ws(_lib.load().frcnn_nms_workspace_bytes(K))
    name = {torch.int16: "frcnn_nms_i16", torch.float64: "frcnn_nms_f64"}[boxes.dtype]
    _lib.call(name, _p(boxes.contiguous()), _p(n), K, float(thresh), int(max_boxes), _p(keep), _p(n_keep), _p(ws), ws.numel(), _stream())
    return keep, n_keep


def gather_rois(cand, keep, n_keep, batch, out_rows, out=None):
    """``out``: write into this (out_rows,4) f32 tensor (a slice of a batch's RoI list) instead of a new one."""
    _require_gpu()
    if out is None:
        out = torch.empty((out_rows, 4), dtype=torch.float32, device="cuda")
    else:
        assert out.dtype == torch.float32 and tuple(out.shape) == (out_rows, 4) and out.is_contiguous()
    _lib.call("frcnn_gather_rois", _p(cand), _p(keep), _p(n_keep), batch, out_rows, _p(out), _stream())
    return out


# ----------------------------------------------------------------------------- detector targets
def roi_targets(rois_i16, gt_f32, gt_f64, gt_cls, bg_idx):
    _require_gpu()
    E, G = rois_i16.shape[0], gt_f32.shape[0]
    elig = torch.zeros(E, dtype=torch.uint8, device="cuda")
    cls = torch.full((E,), bg_idx, dtype=torch.int32, device="cuda")
    tg = torch.zeros((E, 4), dtype=torch.float32, device="cuda")
    _lib.call("frcnn_roi_targets", _p(rois_i16.contiguous()), E, _p(gt_f32.contiguous()), _p(gt_f64.contiguous()),
              _p(gt_cls.contiguous()), G, bg_idx, _p(elig), _p(cls), _p(tg), _stream())
    return elig, cls, tg


# ----------------------------------------------------------------------------- RoI crop/resize
def roi_crop_resize(feat, rois, pool, fill=None, relu=False, layout=0, planes_out=False, n_per_img=0):
    """feat: (R,C,Cf) f32 (or (1,R,C,Cf)); rois: (n,4) f32 -> (n,pool,pool,Cf) f32, or (pool,pool,n,Cf) with layout=1.
    fill (Cf,) = value of an invalid RoI (default zeros); relu clamps the output (frcnn_roi_crop_resize_fwd_ex).
    ``planes_out``: the result as a PlaneTensor for an f16x3 convolution behind it (frcnn_roi_crop_resize_fwd_planes); needs the
    map's magnitude record (its producer's), otherwise the f32 tensor comes back.
    ``n_per_img`` > 0: feat (B,R,C,Cf) holds one map per image and RoI r crops image r // n_per_img's (frcnn_roi_crop_resize_fwd_batch)."""
    _require_gpu()
    src = feat
    if n_per_img > 0:
        assert feat.dim() == 4 and feat.is_contiguous() and rois.reshape(-1, 4).shape[0] <= feat.shape[0] * n_per_img
        rows, cols, C = feat.shape[1:]
    else:
        feat = feat.reshape(feat.shape[-3], feat.shape[-2], feat.shape[-1]).contiguous()
        rows, cols, C = feat.shape
    rois = rois.reshape(-1, 4).to(torch.float32).contiguous()
    n = rois.shape[0]
    oshape = (pool, pool, n, C) if layout else (n, pool, pool, C)
    if planes_out and _tracking() and getattr(src, "_amax", None) is not None and C % 4 == 0 and n > 0:
        floor = 0.0
        if fill is not None:
            floor = getattr(fill, "_absmax", None)
            if floor is None:
                floor = float(fill.abs().max().item())
        out = PlaneTensor(oshape)
        amax_carry(out, src, floor, exponent_out=out.exponent)    # the bound and, from it, the planes' scale: known before the launch
        yp = _lib.H3Planes(planes=out.planes.data_ptr(), exponent=out.exponent.data_ptr(), status=out._amax.data_ptr() + 4)
        if n_per_img > 0:
            _lib.call("frcnn_roi_crop_resize_fwd_batch", _p(feat), rows, cols, C, _p(rois), n, n_per_img, pool, _p(fill), 1 if relu else 0, layout, None, ctypes.byref(yp), _stream())
        else:
            _lib.call("frcnn_roi_crop_resize_fwd_planes", _p(feat), rows, cols, C, _p(rois), n, pool, _p(fill), 1 if relu else 0, layout, ctypes.byref(yp), _stream())
        return out
    out = torch.empty(oshape, dtype=torch.float32, device="cuda")
    if n_per_img > 0:
        _lib.call("frcnn_roi_crop_resize_fwd_batch", _p(feat), rows, cols, C, _p(rois), n, n_per_img, pool, _p(fill), 1 if relu else 0, layout, _p(out), None, _stream())
    else:
        _lib.call("frcnn_roi_crop_resize_fwd_ex", _p(feat), rows, cols, C, _p(rois), n, pool, _p(fill), 1 if relu else 0, layout, _p(out), _stream())
    if _tracking() and getattr(src, "_amax", None) is not None:
        # a bilinear sample is a convex combination of map values; a rejected RoI yields the fill vector
        floor = 0.0
        if fill is not None:
            floor = getattr(fill, "_absmax", None)
            if floor is None:                                    # (PackedConv records it at lowering; a foreign vector costs one host sync)
                floor = float(fill.abs().max().item())
        amax_carry(out, src, floor)
    return out


def avgpool_pos_major(x):
    """(h,w,n,c) f32 position-major -> (n,c): AveragePooling2D over the whole h x w window."""
    _require_gpu()
    h, w, n, c = x.shape
    out = torch.empty((n, c), dtype=torch.float32, device="cuda")
    _lib.call("frcnn_avgpool_pos_major", _p(x.contiguous()), h * w, n, c, _p(out), _stream())
    return out


def roi_crop_resize_bwd(dout, rois, rows, cols):
    _require_gpu()
    n, pool, _, C = dout.shape
    dfeat = torch.empty((rows, cols, C), dtype=torch.float32, device="cuda")
    _lib.call("frcnn_roi_crop_resize_bwd", _p(dout.contiguous()), rows, cols, C, _p(rois.reshape(-1, 4).contiguous()), n, pool, _p(dfeat), _stream())
    return dfeat


# ----------------------------------------------------------------------------- conv engine
ACT = {None: 0, "linear": 0, "none": 0, "relu": 1, "sigmoid": 2}


def same_pad(size, k, stride):
    """TF/Keras padding='same': (out, pad_before)."""
    out = -(-size // stride)
    total = max((out - 1) * stride + k - size, 0)
    return out, total // 2


def valid_out(size, k, stride):
    return (size - k) // stride + 1


# The f16x3 engine scales a whole filter by ONE power of two.  fp16 keeps full precision over 2^-12 .. 2^15 after scaling and loses bits
# gradually below (the low plane and fp16's subnormals stretch that: at 2^-30 of the largest value ~19 bits are left, at 2^-44 about
# six).  A filter whose reduction channels differ by more than 2^H3_MAX_SPREAD_LOG2 in magnitude (max |w| per input channel, largest
# against smallest non-zero) may pair its smallest-weight channels with the LARGEST activations (channel scales that a checkpoint's
# normalisation moved between weights and activations), and then the lost bits show in the sum.  Measured
# (tests/test_h3_fences_gpu.py): at a compensated spread of 2^30 the engine still reads 2.0e-7 of sum |x w| against the native kernel's
# 4.0e-7; the bar is kept well inside that, and a layer beyond it runs on the exact bf16 split.  With the weights inside the bar, ANY
# spread of the activations alone costs nothing (5.6e-7 against the native 1.3e-6 at 2^30).
H3_MAX_SPREAD_LOG2 = 24.0


def _channel_spread_log2(w_hwio, axis):
    """log2(largest / smallest non-zero) of max|w| per reduction channel (``axis`` of the HWIO filter); one host read at lowering."""
    dims = tuple(i for i in range(w_hwio.dim()) if i != axis)
    cm = w_hwio.abs().amax(dim=dims).double()
    nz = cm[cm > 0]
    if nz.numel() < 2:
        return 0.0
    return float(torch.log2(nz.max() / nz.min()).item())


class PackedConv:
    """A convolution's device-resident parameters: filter packed to [cout][packed_k], and the
    per-channel scale/shift the epilogue applies (bias and BatchNorm/Scale folded)."""

    def __init__(self, w_hwio, scale=None, shift=None):
        _require_gpu()
        w = _dev(w_hwio, torch.float32)
        self.kh, self.kw, self.cin, self.cout = (int(v) for v in w.shape)
        kp = _lib.load().frcnn_conv_packed_k(self.kh, self.kw, self.cin)
        self.w = torch.empty((self.cout, kp), dtype=torch.float32, device="cuda")
        _lib.call("frcnn_pack_conv_weights", _p(w), self.kh, self.kw, self.cin, self.cout, _p(self.w), _stream())
        self.scale = None if scale is None else _dev(scale, torch.float32)
        self.shift = None if shift is None else _dev(shift, torch.float32)
        if self.shift is not None and not isinstance(shift, torch.Tensor):
            self.shift._absmax = float(np.abs(np.asarray(shift, dtype=np.float64)).max())      # host copy of max|shift| (amax_carry's floor)
        self._h3_spread_log2 = _channel_spread_log2(w, 2)

    def x6_planes(self):
        """The filter as three bf16 planes [3][cout][packed_k] for the split-bf16 engine (frcnn_conv2d_fwd_x6), derived
        once from the packed f32 filter (6 bytes per weight beside the 4 of the f32 form)."""
        planes = getattr(self, "_x6", None)
        if planes is None or getattr(self, "_x6_src", None) is not self.w:
            planes = torch.empty((3,) + tuple(self.w.shape), dtype=torch.bfloat16, device="cuda")
            _lib.call("frcnn_pack_conv_weights_x6", _p(self.w), self.w.shape[0], self.w.shape[1], _p(planes), _stream())
            self._x6, self._x6_src = planes, self.w
        return planes


    def h3_planes(self):
        """The filter as a 16-byte header (max|w|) + two fp16 planes [2][cout][packed_k] for the f16x3 engine
        (frcnn_conv2d_fwd_h3), derived once from the packed f32 filter (4 bytes per weight)."""
        planes = getattr(self, "_h3", None)
        if planes is None or getattr(self, "_h3_src", None) is not self.w:
            planes = torch.empty(_lib.load().frcnn_conv_h3_planes_bytes(self.w.shape[0], self.w.shape[1]), dtype=torch.uint8, device="cuda")
            _lib.call("frcnn_pack_conv_weights_h3", _p(self.w), self.w.shape[0], self.w.shape[1], _p(planes), _stream())
            self._h3, self._h3_src = planes, self.w
        return planes


    def h3_bound(self):
        """(bound_c, bound_d) of frcnn_conv2d_fwd_h3_planes: |y| <= bound_c * max|x| + bound_d for ANY input -- the largest
        |scale[c]| * sum |w[.,.,.,c]| over the output channels and the largest |shift[c]| (host floats, computed once per filter)."""
        b = getattr(self, "_h3_bound", None)
        if b is None or getattr(self, "_h3_bound_src", None) is not self.w:
            l1 = self.w.abs().sum(dim=1)
            if self.scale is not None:
                l1 = l1 * self.scale.abs()
            b = (float(l1.max().item()), 0.0 if self.shift is None else float(self.shift.abs().max().item()))
            self._h3_bound, self._h3_bound_src = b, self.w
        return b


class PlaneTensor:
    """An activation tensor the way the f16x3 engine multiplies it: two fp16 planes (2, *shape) -- hi = f16(v * 2^e), lo = f16((v * 2^e
    - hi) * 2^11) -- and the device int32 ``exponent`` e, written by the launch that produced it (frcnn_conv2d_fwd_h3_planes).  Only
    convolutions read it (``conv2d`` accepts it in place of the f32 tensor); ``_amax`` is the producer's magnitude record."""
    is_planes = True

    def __init__(self, shape):
        self.shape = tuple(int(v) for v in shape)
        self.planes = torch.empty((2,) + self.shape, dtype=torch.float16, device="cuda")
        self.exponent = torch.empty(1, dtype=torch.int32, device="cuda")
        self._amax = None

    def float(self):
        """The f32 values (hi + lo * 2^-11) * 2^-e, on the device (tests / debugging: torch arithmetic)."""
        e = self.exponent.to(torch.float64)
        return ((self.planes[0].double() + self.planes[1].double() / 2048.0) * torch.pow(torch.tensor(2.0, dtype=torch.float64, device="cuda"), -e)).float()


# ---- which matrix path an fp32 convolution takes.
# "native": v_mfma_f32_32x32x2_f32 (csrc/conv_igemm.hip).  "bf16x6": the same GEMM on the bf16 matrix cores by exact
# three-way operand splitting (csrc/conv_x6.hip) for the launches where it measures faster -- large row counts, cin % 32 == 0;
# everything else stays native.  fp32-grade either way (error against fp64 at or below the native kernel's), but a
# different summation order: the two agree to f32 rounding, not bit for bit, so the library default stays "native" and a
# caller opts in for a scope (``with f32_engine("bf16x6"):`` -- bench.py and entry.DetectionEntry do, around their captures).
# "f16x3" (round 5): the same launches on the fp16 matrix cores by a two-way split with a scaled low part (csrc/conv_h3.hip):
# three matrix instructions per block of products instead of six, error against fp64 at or under the native kernel's; every
# tensor an f16x3 launch reads carries a device-resident magnitude record (``AmaxArena``, ``amax_of``) from which the launch
# derives the power of two that brings the tensor into fp16's range.
F32_ENGINE = "native"
# where the split engine measures faster than the native kernels (MI355X, configs[1] shapes, each launch alone on the chip;
# scripts/conv_shapes.py 0,74,77,71,76): the head's 14 700-row GEMMs 355 / 189 / 167 us against 530 / 275 / 269; with 64x64 tiles
# almost every trunk layer too -- stage 2 (37 101 rows) 3x3 30.9 vs 36.3 us, 1x1 23.4 vs 27.2 / 19.0 vs 20.9, stage 3's wide 1x1
# 19.6 vs 22.0 / 28.2 vs 32.6, stage 4's 1024-column 1x1 18.9 vs 20.9 / 28.7 vs 33.6 -- as long as the launch has >= 256 tiles of
# 64x64; under that (stage 4's 256-column layers: 26.8 vs 21.1 us) the native split-K launches win, except where the engine's own
# split-K form applies (rpn_conv1 176 vs 202, stage 4's 3x3 32.7 vs 34.9).  Layers with fewer than 64 columns stay native.
X6_MIN_TILES = 256              # 64x64 tiles of the output
X6_MIN_COUT = 64


class f32_engine:
    def __init__(self, name):
        assert name in ("native", "bf16x6", "f16x3")
        self.name = name

    def __enter__(self):
        global F32_ENGINE
        self.prev, F32_ENGINE = F32_ENGINE, self.name

    def __exit__(self, *exc):
        global F32_ENGINE
        F32_ENGINE = self.prev


_ENGINE_CODE = {"native": 0, "bf16x6": 1, "f16x3": 2}
_ENGINE_TAG = {0: None, 1: "x6", 2: "h3"}


def _split_engine(d, pc, tile):
    """Which split engine a forward launch takes: "x6" (bf16, six products), "h3" (fp16, three products) or None (native).
    The policy itself lives in the library (frcnn_conv2d_engine: explicit tile codes 71..78 / 81..88 pick their engine, otherwise
    the engine of the ``F32_ENGINE`` scope takes the launches where it measures faster -- >= X6_MIN_TILES output tiles of 64x64
    and >= X6_MIN_COUT columns, or its split-K form when split-K launches are allowed), so a host in another language gets the
    same choices; memoised on the cached descriptor."""
    key = ("engine", F32_ENGINE, _CONV_WS is not NO_SPLIT_K)
    memo = getattr(d, "_ws", None)
    if memo is not None and d.tile == tile and d.cout == pc.cout and d.cin == pc.cin:
        got = memo.get(key)
        if got is None:
            got = memo[key] = _lib.load().frcnn_conv2d_engine(ctypes.byref(d), _ENGINE_CODE[F32_ENGINE], 1 if key[2] else 0)
            if got < 0:
                _lib.check(got, "frcnn_conv2d_engine")
    else:
        q = _lib.ConvDesc.from_buffer_copy(d)                 # (a caller asking about another tile code / filter than the descriptor's)
        q.tile, q.cin, q.cout = tile, pc.cin, pc.cout
        got = _lib.load().frcnn_conv2d_engine(ctypes.byref(q), _ENGINE_CODE[F32_ENGINE], 1 if key[2] else 0)
        if got < 0:
            _lib.check(got, "frcnn_conv2d_engine")
    if got == 2 and not (81 <= tile <= 88) and getattr(pc, "_h3_spread_log2", 0.0) > H3_MAX_SPREAD_LOG2:
        return "x6"                                           # an ill-scaled filter (H3_MAX_SPREAD_LOG2): the exact split (an explicit 8x tile code stands)
    return _ENGINE_TAG[got]


def _use_x6(d, pc, tile):
    return _split_engine(d, pc, tile) == "x6"


# ---- magnitude records of the f16x3 engine (include/frcnn_hip.h, "Magnitude records").  A tensor that an f16x3 launch may read
# carries ``t._amax``: a device record holding an upper bound of max|t|, written by the launch that produced the tensor.
class AmaxArena:
    """The records of ONE forward pass: a fixed pool handed out in call order, cleared by a kernel at the start of the pass, so a
    captured pass re-uses the same addresses on every replay and contains no memset node.

    ``gen`` counts the passes begun: a record handed out carries the generation it belongs to, and ``amax_of`` refuses one of an earlier
    pass (its words have been cleared, or belong to another tensor now) -- the tensor is measured again instead (a cached
    conv map, an ``out=`` buffer).  ``status()`` gathers the records' sticky status words (FRCNN_H3_*: a value above its record's
    bound, a clamped value, a non-finite record) into ONE device word to be read with the pass's outputs."""

    def __init__(self, n=192):
        _require_gpu()
        self.floats = _lib.load().frcnn_amax_record_floats()
        self.buf = torch.zeros((n, self.floats), dtype=torch.float32, device="cuda")
        self.status_word = torch.zeros(1, dtype=torch.int32, device="cuda")
        self.i = self.high_water = self.gen = 0

    def begin(self):
        """Clear the records the passes so far have used (all of them were zero at allocation; a captured pass bakes in the count of
        its warm-up passes, which walk the same launches): ~60 records = 240 KB for a ResNet-50 pass instead of the whole pool."""
        self.high_water = max(self.high_water, self.i)
        self.i = 0
        self.gen += 1
        if self.high_water:
            _lib.call("frcnn_amax_clear", _p(self.buf), self.high_water, _stream())

    def take(self):
        assert self.i < self.buf.shape[0], "AmaxArena: more tracked tensors in a pass than records"
        r = self.buf[self.i]
        r._arena_gen = (self, self.gen)
        self.i += 1
        return r

    def status(self, out=None):
        """-> device int32 word (``out`` or the arena's own): OR of the status words of the records this pass has used so far."""
        out = self.status_word if out is None else out
        n = max(self.i, 1)
        _lib.call("frcnn_amax_status", _p(self.buf), n, _p(out), _stream())
        return out


def h3_status_text(bits):
    """Readable form of a FRCNN_H3_* status word (entry.DetectionEntry raises with it)."""
    names = [(n, b) for n, b in (("a value above its magnitude record's bound", _lib.H3_UNDER), ("values clamped to fp16's range", _lib.H3_SATURATED),
                                  ("a non-finite magnitude record (Inf in a tensor)", _lib.H3_NONFINITE)) if bits & b]
    return "; ".join(n for n, _ in names) or "clean"


_AMAX_ARENA = None
AMAX_MEASURED = 0           # tensors whose bound had to be measured by a pass of their own (no producer record): a perf counter


class amax_arena:
    """``with amax_arena(arena):`` f16x3 launches inside take their records from ``arena`` (None: one fresh allocation each)."""

    def __init__(self, arena):
        self.arena = arena

    def __enter__(self):
        global _AMAX_ARENA
        self.prev, _AMAX_ARENA = _AMAX_ARENA, self.arena
        return self.arena

    def __exit__(self, *exc):
        global _AMAX_ARENA
        _AMAX_ARENA = self.prev


def amax_begin():
    """Start of a forward pass: clear the active arena's records (no arena: nothing to do)."""
    if _AMAX_ARENA is not None:
        _AMAX_ARENA.begin()


def _amax_new(optional=False):
    """A cleared record: from the active arena, or (eager launches) a fresh allocation.  A capture without an arena cannot have one
    (a zero-filling allocation would put a memset node into the graph): an error for an f16x3 launch, None for a native layer that
    would merely have left its record for a layer behind it (``optional``)."""
    if _AMAX_ARENA is not None:
        return _AMAX_ARENA.take()
    if torch.cuda.is_current_stream_capturing():
        if optional:
            return None
        raise _lib.FrcnnError("f16x3 launches inside a capture need an ops.amax_arena (pipeline.InferencePipeline.capture makes one)")
    return torch.zeros(_lib.load().frcnn_amax_record_floats(), dtype=torch.float32, device="cuda")


def _tracking():
    return F32_ENGINE == "f16x3"


def amax_of(x):
    """The magnitude record of tensor ``x``: the one its producer attached, or a MEASURED one (one pass over x).  A measured record is
    not kept on the tensor: a tensor nobody produced is somebody's input buffer, rewritten between passes (a captured pass's static
    input), and its record lives in an arena that the next pass clears."""
    rec = getattr(x, "_amax", None)
    if rec is not None:
        stamp = getattr(rec, "_arena_gen", None)
        if stamp is not None and stamp[0].gen != stamp[1]:      # a record of an EARLIER pass of its arena: cleared or re-used since
            rec = None
            try:
                x._amax = None
            except AttributeError:
                pass
    if rec is None:
        global AMAX_MEASURED
        AMAX_MEASURED += 1
        rec = _amax_new()
        _lib.call("frcnn_amax_f32", _p(x), x.numel(), _p(rec), _stream())
    return rec


def amax_carry(dst, src, floor=0.0, exponent_out=None):
    """``dst`` was derived from ``src`` by a map that cannot exceed max(|src|, floor) (a view, max-pooling, ReLU, the bilinear RoI
    resampling with a fill vector): it inherits the bound.  floor > 0 (or a request for the bound's plane exponent) needs a record
    of its own (frcnn_amax_merge)."""
    rec = getattr(src, "_amax", None)
    stamp = getattr(rec, "_arena_gen", None) if rec is not None else None
    if stamp is not None and stamp[0].gen != stamp[1]:
        rec = None                                               # (a record of an earlier pass: see amax_of)
    if rec is None:
        if exponent_out is not None:
            rec = amax_of(src)                                   # planes need their scale: measure
        else:
            return dst
    if floor > 0.0 or exponent_out is not None:
        merged = _amax_new()
        _lib.call("frcnn_amax_merge", _p(merged), _p(rec), float(floor), _p(exponent_out), _stream())
        rec = merged
    dst._amax = rec
    return dst


def conv_accepts_planes(x_shape, pc, stride=1, padding="valid", act=None, layout=0, tile=0):
    """Would ``conv2d`` of a tensor of this shape read fp16 planes (the f16x3 engine on its 256x128 tile, un-split)?  Producers ask
    before they hand a PlaneTensor on."""
    if F32_ENGINE != "f16x3" or not isinstance(pc, PackedConv):
        return False
    d = _conv_desc(tuple(x_shape), pc.kh, pc.kw, pc.cout, stride, padding, ACT[act], layout, tile or AUTO_TILE)
    return _planes_ok(d, pc, _split_engine(d, pc, tile or AUTO_TILE))


H3_KERNEL_NAMES = {81: "k_conv_igemm_h3<2,1,2,4>", 82: "k_conv_igemm_h3_db<2,2,4,2>", 83: "k_conv_igemm_h3<2,2,2,2>", 84: "k_conv_igemm_h3<1,1,2,2>",
                   85: "k_conv_igemm_h3_db<2,1,4,4>", 86: "k_conv_igemm_h3_db<2,1,4,4>", 87: "k_conv_igemm_h3<2,1,2,2>"}


def _h3_name(d, n1=0):
    return H3_KERNEL_NAMES[_lib.load().frcnn_conv2d_h3_config(ctypes.byref(d), n1)]


X6_KERNEL_NAMES = {71: "k_conv_igemm_x6<2,1,2,4>", 72: "k_conv_igemm_x6<2,2,4,2>", 73: "k_conv_igemm_x6<2,2,2,2>", 74: "k_conv_igemm_x6<1,1,2,2>",
                   75: "k_conv_igemm_x6<1,1,2,4>", 76: "k_conv_igemm_x6_db", 77: "k_conv_igemm_x6<2,1,2,2>"}


def _x6_name(d, n1=0):
    return X6_KERNEL_NAMES[_lib.load().frcnn_conv2d_x6_config(ctypes.byref(d), n1)]


# When set to a list, every conv2d launch appends {kernel, flops, shape, relaunch()} so bench.py can
# re-issue each distinct launch back to back between one HIP-event pair on the launch stream.
CONV_PROFILE = None
CONV_KERNEL_NAMES = {1: "k_conv_igemm_f32<2,2,false>", 2: "k_conv_igemm_f32<1,1,*>", 3: "k_conv_igemm_f32<2,1,*>",
                     4: "k_conv_igemm_f32<4,2,false>", 11: "k_conv_igemm_f32_v2<2,2>", 12: "k_conv_igemm_f32_v2<1,1>",
                     13: "k_conv_igemm_f32_v2<2,1>", 14: "k_conv_igemm_f32_v2<4,2>",
                     21: "k_conv_igemm_f32_v2<2,2,1>", 22: "k_conv_igemm_f32_v2<1,1,1>",
                     23: "k_conv_igemm_f32_v2<1,1,2>", 24: "k_conv_igemm_f32_v2<1,2,2>", 25: "k_conv_igemm_f32_v2<2,1,2>", 26: "k_conv_igemm_f32_v2<2,2,2>",
                     30: "k_conv_igemm_f32_v2<1,1,2> stem", 32: "k_conv_igemm_f32_v2<1,1,1> stem", 61: "k_conv_igemm_f32_sk<2,2>", 62: "k_conv_igemm_f32_sk<1,1>", 41: "k_conv_igemm_f32_v2<1,2,1,4,2>", 42: "k_conv_igemm_f32_v2<2,1,1,2,4>", 43: "k_conv_igemm_f32_v2<1,1,1,4,2>"}


class ConvWorkspace:
    """Split-K workspace (arrival tickets + f32 partial tiles) of frcnn_conv2d_fwd_ws.

    The C ABI wants the ticket words zero on entry and leaves them zero, so one buffer zeroed at
    allocation serves every conv launched on ONE stream (or replayed from ONE hipGraph).  Streams or
    graphs that may run concurrently need one ``ConvWorkspace`` each (pipeline.py owns one per graph).
    """

    def __init__(self, nbytes=0):
        self.buf = None
        if nbytes:
            self.get(nbytes)

    def get(self, nbytes):
        if self.buf is None or self.buf.numel() < nbytes:
            self.buf = torch.zeros(int(nbytes), dtype=torch.uint8, device="cuda")
        return self.buf


NO_SPLIT_K = object()       # ``with conv_workspace(NO_SPLIT_K):`` = plain launches only
_CONV_WS = None             # the workspace conv launches use right now (None: per-stream default)
_STREAM_WS = {}             # eager launches: HIP stream handle -> ConvWorkspace (same-stream launches serialise)


AUTO_TILE = 0               # tile code substituted for tile=0: 0 = launch alone on the chip, 50 = beside other streams' launches


class tile_policy:
    """``with tile_policy(throughput=True):`` conv launches inside pick their tile for a chip shared with other
    images' launches (frcnn_conv_desc.tile = 50) instead of for a launch that has the chip to itself."""

    def __init__(self, throughput):
        self.code = 50 if throughput else 0

    def __enter__(self):
        global AUTO_TILE
        self.prev, AUTO_TILE = AUTO_TILE, self.code

    def __exit__(self, *exc):
        global AUTO_TILE
        AUTO_TILE = self.prev


class conv_workspace:
    """``with conv_workspace(ws):`` routes the split-K launches inside to ``ws``."""

    def __init__(self, ws):
        self.ws = ws

    def __enter__(self):
        global _CONV_WS
        self.prev, _CONV_WS = _CONV_WS, self.ws
        return self.ws

    def __exit__(self, *exc):
        global _CONV_WS
        _CONV_WS = self.prev


def _split_k_ws(need):
    """The split-K workspace tensor a launch needing `need` bytes should use right now, or None (plain launch)."""
    if not need or _CONV_WS is NO_SPLIT_K:
        return None
    holder = _CONV_WS
    if holder is None and not torch.cuda.is_current_stream_capturing():
        # a capture without an explicit workspace takes the plain launch: graphs replay concurrently
        holder = _STREAM_WS.setdefault(torch.cuda.current_stream().cuda_stream, ConvWorkspace())
    return None if holder is None else holder.get(need)


def _conv_launch(d, x, w, scale, shift, residual, mask, out, y_amax=None):
    """frcnn_conv2d_fwd_ws with the right workspace (``y_amax``: frcnn_conv2d_fwd_ws_amax, which also folds max|y| into that
    record); returns (entry point, ctypes argument tuple) for re-launches, and the workspace."""
    ws = _split_k_ws(_ws_need(d, "frcnn_conv2d_workspace_bytes"))
    head = (ctypes.byref(d), _p(x), _p(w), _p(scale), _p(shift), _p(residual), _p(mask), _p(out))
    tail = (_p(ws), ws.numel() if ws is not None else 0)
    fn, args = ("frcnn_conv2d_fwd_ws", head + tail) if y_amax is None else ("frcnn_conv2d_fwd_ws_amax", head + (_p(y_amax),) + tail)
    _lib.call(fn, *args, _stream())
    return (fn, args), ws


def _conv2d_h3_planes(d, x, pc, residual, oshape, planes_out, act):
    """frcnn_conv2d_fwd_h3_planes: x a PlaneTensor or an f32 tensor, the result a PlaneTensor (planes_out) or an f32 tensor."""
    x_in = isinstance(x, PlaneTensor)
    xp = _lib.H3Planes(planes=x.planes.data_ptr(), exponent=x.exponent.data_ptr(), status=None) if x_in else None
    y = PlaneTensor(oshape) if planes_out else torch.empty(oshape, dtype=torch.float32, device="cuda")
    ya = _amax_new()
    yp = _lib.H3Planes(planes=y.planes.data_ptr(), exponent=y.exponent.data_ptr(), status=ya.data_ptr() + 4) if planes_out else None
    bc, bd = pc.h3_bound() if planes_out else (0.0, 0.0)
    xa = amax_of(x)                                              # (a measured record is a temporary: it must outlive the launch and its re-launches)
    ra = amax_of(residual) if (residual is not None and planes_out) else None
    r_in = isinstance(residual, PlaneTensor)                     # a block's output handed on as planes: the shortcut reads them back
    rp = _lib.H3Planes(planes=residual.planes.data_ptr(), exponent=residual.exponent.data_ptr(), status=None) if r_in else None
    args = (ctypes.byref(d), None if x_in else _p(x), ctypes.byref(xp) if x_in else None, _p(xa), _p(pc.h3_planes()), _p(pc.scale), _p(pc.shift),
            None if r_in else _p(residual), ctypes.byref(rp) if r_in else None, _p(ra), None if planes_out else _p(y), _p(ya),
            ctypes.byref(yp) if planes_out else None, bc, bd)
    _lib.call("frcnn_conv2d_fwd_h3_planes_res", *args, _stream())
    y._amax = ya
    if CONV_PROFILE is not None:
        keep = (d, x, pc, residual, y, ya, xp, yp, xa, ra, rp)
        # (the instantiation that reads planes is a kernel of its own; writing planes is a run-time branch of either's epilogue)
        CONV_PROFILE.append({"kernel": _h3_planes_in_name(d, pc) if x_in else _h3_name(d), "planes_out": bool(planes_out),
                             "flops": 2.0 * d.n * d.ho * d.wo * pc.cout * pc.kh * pc.kw * pc.cin,
                             "shape": (d.n * d.ho * d.wo, pc.cout, pc.kh * pc.kw * pc.cin, d.stride),
                             "relaunch": lambda args=args, keep=keep: _lib.call("frcnn_conv2d_fwd_h3_planes_res", *args, _stream())})
    return y


def _h3_planes_in_name(d, pc):
    """The kernel a plane-input launch runs: the plane-reading instantiation of the 256x128 double buffer (which walks long reductions on
    its three-stage ring, csrc/conv_h3.hip h3_ring_tile: one kernel, two loops)."""
    return _h3_name(d).replace(">", ",planes>")


def _planes_ok(d, pc, eng):
    """A launch may read / write fp16 planes: the f16x3 engine on its double-buffered 256x128 tile, un-split, channel counts the 16- and
    8-byte pieces divide."""
    return (eng == "h3" and _lib.load().frcnn_conv2d_h3_config(ctypes.byref(d), 0) in (86, 85, 82) and pc.cout % 4 == 0 and pc.cin % 8 == 0
            and (_CONV_WS is NO_SPLIT_K or _ws_need(d, "frcnn_conv2d_h3_workspace_bytes") == 0))


def conv2d(x, pc, stride=1, padding="valid", act=None, residual=None, out=None, tile=0, layout=0, planes_out=False):
    """x: (n,h,w,cin) f32 NHWC device tensor; pc: PackedConv -> (n,ho,wo,cout).
    layout=1: position-major tensors, x (h,w,n,cin) -> (ho,wo,n,cout) (frcnn_conv_desc.layout).
    ``planes_out``: a REQUEST to hand the result on as a PlaneTensor (the caller knows its only readers are f16x3 convolutions of the
    detector head's size); honoured when this launch runs on the engine's 256x128 tile, otherwise the f32 tensor comes back as usual.
    x may be a PlaneTensor (then the launch must be such a one: anything else raises)."""
    _require_gpu()
    x_planes = isinstance(x, PlaneTensor)
    assert (x_planes or (x.dtype == torch.float32 and x.is_contiguous())) and x.shape[-1] == pc.cin, (x.shape, pc.cin)
    d = _conv_desc(tuple(x.shape), pc.kh, pc.kw, pc.cout, stride, padding, ACT[act], layout, tile or AUTO_TILE)
    n, ho, wo = d.n, d.ho, d.wo
    oshape = (ho, wo, n, pc.cout) if layout else (n, ho, wo, pc.cout)
    res_planes = isinstance(residual, PlaneTensor)
    if x_planes or planes_out or res_planes:
        ok = out is None and _planes_ok(d, pc, _split_engine(d, pc, tile or AUTO_TILE))
        if (x_planes or res_planes) and not ok:
            raise _lib.FrcnnError("conv2d: a PlaneTensor input / residual needs an f16x3 launch on the 256x128 tile (frcnn_conv2d_fwd_h3_planes)")
        if ok:
            if residual is not None:
                assert tuple(residual.shape) == tuple(oshape) and (res_planes or residual.is_contiguous())
            return _conv2d_h3_planes(d, x, pc, residual, oshape, planes_out, act)
    if out is None:
        out = torch.empty(oshape, dtype=torch.float32, device="cuda")
    else:
        assert out.shape == oshape and out.is_contiguous()
    if residual is not None:
        assert residual.shape == out.shape and residual.is_contiguous()
    eng = _split_engine(d, pc, tile or AUTO_TILE)
    if eng == "h3":
        ws = _split_k_ws(_ws_need(d, "frcnn_conv2d_h3_workspace_bytes"))
        ya, xa = _amax_new(), amax_of(x)
        args = (ctypes.byref(d), _p(x), _p(xa), _p(pc.h3_planes()), _p(pc.scale), _p(pc.shift), _p(residual), None, _p(out), _p(ya),
                _p(ws), ws.numel() if ws is not None else 0)
        _lib.call("frcnn_conv2d_fwd_h3", *args, _stream())
        out._amax = ya
        if CONV_PROFILE is not None:
            keep = (d, x, pc, residual, out, ws, ya, xa)
            CONV_PROFILE.append({"kernel": "k_conv_igemm_h3<1,1,2,2> split-K" if ws is not None else _h3_name(d),
                                 "flops": 2.0 * n * ho * wo * pc.cout * pc.kh * pc.kw * pc.cin,
                                 "shape": (n * ho * wo, pc.cout, pc.kh * pc.kw * pc.cin, stride),
                                 "relaunch": lambda args=args, keep=keep: _lib.call("frcnn_conv2d_fwd_h3", *args, _stream())})
        return out
    if eng == "x6":
        ws = _split_k_ws(_ws_need(d, "frcnn_conv2d_x6_workspace_bytes"))
        args = (ctypes.byref(d), _p(x), _p(pc.x6_planes()), _p(pc.scale), _p(pc.shift), _p(residual), None, _p(out),
                _p(ws), ws.numel() if ws is not None else 0)
        _lib.call("frcnn_conv2d_fwd_x6", *args, _stream())
        if getattr(out, "_amax", None) is not None:
            out._amax = None
        if CONV_PROFILE is not None:
            keep = (d, x, pc, residual, out, ws)
            CONV_PROFILE.append({"kernel": "k_conv_igemm_x6<1,1,2,2> split-K" if ws is not None else _x6_name(d),
                                 "flops": 2.0 * n * ho * wo * pc.cout * pc.kh * pc.kw * pc.cin,
                                 "shape": (n * ho * wo, pc.cout, pc.kh * pc.kw * pc.cin, stride),
                                 "relaunch": lambda args=args, keep=keep: _lib.call("frcnn_conv2d_fwd_x6", *args, _stream())})
        return out
    ya = _amax_new(optional=True) if (_tracking() and pc.cout >= 32) else None      # a native layer in an f16x3 pass leaves max|y| for the layer behind it
    (fn, args), ws = _conv_launch(d, x, pc.w, pc.scale, pc.shift, residual, None, out, ya)
    if ya is not None or getattr(out, "_amax", None) is not None:
        out._amax = ya                                           # (an `out=` buffer must not keep the record of what it held before)
    if CONV_PROFILE is not None:
        flops = 2.0 * n * ho * wo * pc.cout * pc.kh * pc.kw * pc.cin
        cfg = _lib.load().frcnn_conv2d_config(ctypes.byref(d))
        if cfg in (61, 62) and ws is None:
            cfg -= 40                                            # balanced form needs a workspace: the plain tile ran
        kname = CONV_KERNEL_NAMES.get(cfg, "?") + (" split-K" if ws is not None and cfg not in (61, 62) else "")
        keep = (d, x, pc, residual, out, ws, ya)
        CONV_PROFILE.append({"kernel": kname, "flops": flops, "shape": (n * ho * wo, pc.cout, pc.kh * pc.kw * pc.cin, stride),
                             "relaunch": lambda args=args, keep=keep, fn=fn: _lib.call(fn, *args, _stream())})
    return out


def conv2d_dual(x, pc, n1, stride=1, padding="valid", act1=None, act2=None, layout=0, tile=0):
    """Two layers on the same input in one launch (frcnn_conv2d_fwd_dual): ``pc`` packs the two filters concatenated along
    the output axis, the first ``n1`` output channels are layer 1.  -> (y1 (n,ho,wo,n1), y2 (n,ho,wo,cout-n1))."""
    _require_gpu()
    assert x.dtype == torch.float32 and x.is_contiguous() and x.shape[-1] == pc.cin and 0 < n1 < pc.cout
    d = _conv_desc(tuple(x.shape), pc.kh, pc.kw, pc.cout, stride, padding, 0, layout, tile or AUTO_TILE)
    lead = (d.ho, d.wo, d.n) if layout else (d.n, d.ho, d.wo)
    y1 = torch.empty(lead + (n1,), dtype=torch.float32, device="cuda")
    y2 = torch.empty(lead + (pc.cout - n1,), dtype=torch.float32, device="cuda")
    eng = _split_engine(d, pc, tile or AUTO_TILE)
    if eng == "h3":
        a1, a2, xa = _amax_new(), _amax_new(), amax_of(x)
        args = (ctypes.byref(d), _p(x), _p(xa), _p(pc.h3_planes()), _p(pc.scale), _p(pc.shift), _p(y1), n1, ACT[act1], _p(a1),
                _p(y2), ACT[act2], _p(a2))
        _lib.call("frcnn_conv2d_fwd_dual_h3", *args, _stream())
        y1._amax, y2._amax = a1, a2
        if CONV_PROFILE is not None:
            keep = (d, x, pc, y1, y2, a1, a2, xa)
            CONV_PROFILE.append({"kernel": _h3_name(d, n1), "flops": 2.0 * d.n * d.ho * d.wo * pc.cout * pc.kh * pc.kw * pc.cin,
                                 "shape": (d.n * d.ho * d.wo, pc.cout, pc.kh * pc.kw * pc.cin, stride),
                                 "relaunch": lambda args=args, keep=keep: _lib.call("frcnn_conv2d_fwd_dual_h3", *args, _stream())})
        return y1, y2
    if eng == "x6":
        args = (ctypes.byref(d), _p(x), _p(pc.x6_planes()), _p(pc.scale), _p(pc.shift), _p(y1), n1, ACT[act1], _p(y2), ACT[act2])
        _lib.call("frcnn_conv2d_fwd_dual_x6", *args, _stream())
        if CONV_PROFILE is not None:
            keep = (d, x, pc, y1, y2)
            CONV_PROFILE.append({"kernel": _x6_name(d, n1), "flops": 2.0 * d.n * d.ho * d.wo * pc.cout * pc.kh * pc.kw * pc.cin,
                                 "shape": (d.n * d.ho * d.wo, pc.cout, pc.kh * pc.kw * pc.cin, stride),
                                 "relaunch": lambda args=args, keep=keep: _lib.call("frcnn_conv2d_fwd_dual_x6", *args, _stream())})
        return y1, y2
    ws = _split_k_ws(_ws_need(d, "frcnn_conv2d_dual_workspace_bytes"))
    args = (ctypes.byref(d), _p(x), _p(pc.w), _p(pc.scale), _p(pc.shift), _p(y1), n1, ACT[act1], _p(y2), ACT[act2],
            _p(ws), ws.numel() if ws is not None else 0)
    _lib.call("frcnn_conv2d_fwd_dual", *args, _stream())
    if CONV_PROFILE is not None:
        flops = 2.0 * d.n * d.ho * d.wo * pc.cout * pc.kh * pc.kw * pc.cin
        cfg = _lib.load().frcnn_conv2d_dual_config(ctypes.byref(d), 1 if ws is not None else 0)     # the remapped code the launch ran
        kname = CONV_KERNEL_NAMES.get(cfg, "?") + (" split-K" if ws is not None else "")
        keep = (d, x, pc, y1, y2, ws)
        CONV_PROFILE.append({"kernel": kname, "flops": flops, "shape": (d.n * d.ho * d.wo, pc.cout, pc.kh * pc.kw * pc.cin, stride),
                             "relaunch": lambda args=args, keep=keep: _lib.call("frcnn_conv2d_fwd_dual", *args, _stream())})
    return y1, y2


def pool2d(x, k, stride, is_max=True, planes_out=False):
    """``planes_out``: a REQUEST for the result as a PlaneTensor (the caller knows its only reader is an f16x3 convolution that reads planes:
    VGG's block<n>_conv1 behind a max-pool); honoured under the f16x3 engine with a magnitude record on x, channels in fours."""
    _require_gpu()
    n, h, w, c = x.shape
    ho, wo = valid_out(h, k, stride), valid_out(w, k, stride)
    if planes_out and F32_ENGINE == "f16x3" and _tracking() and c % 4 == 0:
        out = PlaneTensor((n, ho, wo, c))
        amax_carry(out, x, 0.0, exponent_out=out.exponent)       # a window's maximum / mean cannot exceed the largest |input|: the scale is known before the launch
        yp = _lib.H3Planes(planes=out.planes.data_ptr(), exponent=out.exponent.data_ptr(), status=out._amax.data_ptr() + 4)
        _lib.call("frcnn_pool2d_fwd_planes", _p(x.contiguous()), n, h, w, c, k, stride, 1 if is_max else 0, ctypes.byref(yp), _stream())
        return out
    out = torch.empty((n, ho, wo, c), dtype=torch.float32, device="cuda")
    _lib.call("frcnn_pool2d_fwd", _p(x.contiguous()), n, h, w, c, k, stride, 1 if is_max else 0, _p(out), _stream())
    return amax_carry(out, x)                                    # max / mean of a window never exceeds the largest |input|


def softmax_rows(x, cols=None):
    _require_gpu()
    rows, ld = x.shape
    cols = ld if cols is None else cols
    out = torch.empty((rows, cols), dtype=torch.float32, device="cuda")
    _lib.call("frcnn_softmax_rows", _p(x.contiguous()), rows, cols, ld, _p(out), cols, _stream())
    return out


def dense_heads_split(y, num_classes):
    """(rows, C + 4(C-1)) output of the merged dense GEMM -> (softmax class probabilities (rows, C), regressions
    (rows, 4(C-1))), both dense, in one launch."""
    _require_gpu()
    rows, ld = y.shape
    tail = ld - num_classes
    cls = torch.empty((rows, num_classes), dtype=torch.float32, device="cuda")
    reg = torch.empty((rows, tail), dtype=torch.float32, device="cuda")
    _lib.call("frcnn_dense_heads_split", _p(y.contiguous()), rows, num_classes, tail, ld, _p(cls), _p(reg), _stream())
    return cls, reg


# ----------------------------------------------------------------------------- detections
def detections(rois, n_rois, out_cls, out_reg, roi_batch, bg_idx, det_threshold, stride, resize_ratio, nms_thresh=0.5):
    """voc_dets.get_dets post-process on the device -> dict of device tensors."""
    _require_gpu()
    rows, C = out_cls.shape
    # ONE allocation, the five outputs are views into it: a client that wants the detections on the host moves
    # `det_packed` with a single copy (five small D2H copies cost ~0.25 ms of stream time per image; split_detections()
    # carves the same views out of the host copy).  Layout in 4-byte words: [n_dets, pad x3 | bbox 4R | cls R | prob R | roi R]
    packed = torch.empty(4 + 7 * rows, dtype=torch.int32, device="cuda")      # the call writes every row (-1 / 0 past n_dets)
    n_dets, det_bbox, det_cls, det_prob, det_roi = split_detections(packed, rows)
    _lib.call("frcnn_detections", _p(rois), _p(n_rois), rows, _p(out_cls.contiguous()), _p(out_reg.contiguous()), C, int(bg_idx),
              float(det_threshold), float(stride), float(resize_ratio), float(nms_thresh),
              _p(det_cls), _p(det_prob), _p(det_bbox), _p(det_roi), _p(n_dets), _stream())
    return {"det_cls": det_cls, "det_prob": det_prob, "det_bbox": det_bbox, "det_roi": det_roi, "n_dets": n_dets, "det_packed": packed}


def detections_dyn(rois, n_rois, out_cls, out_reg, roi_batch, bg_idx, stride, dyn, nms_thresh=0.5):
    """`detections` for a captured pass that serves every image of one size: resize_ratio and det_threshold are read from
    the device tensor ``dyn`` (2 x f64), the reference's padded RoI list is scored when ``roi_batch`` > 0, and word 1 of
    `det_packed` carries the number of proposals the NMS kept (frcnn_detections_dyn)."""
    _require_gpu()
    rows, C = out_cls.shape
    assert dyn.dtype == torch.float64 and dyn.numel() >= 2 and dyn.is_cuda
    packed = torch.empty(4 + 7 * rows, dtype=torch.int32, device="cuda")
    n_dets, det_bbox, det_cls, det_prob, det_roi = split_detections(packed, rows)
    _lib.call("frcnn_detections_dyn", _p(rois), _p(n_rois), int(roi_batch), rows, _p(out_cls.contiguous()), _p(out_reg.contiguous()), C,
              int(bg_idx), float(stride), float(nms_thresh), dyn.data_ptr(),
              _p(det_cls), _p(det_prob), _p(det_bbox), _p(det_roi), _p(packed), _stream())
    return {"det_cls": det_cls, "det_prob": det_prob, "det_bbox": det_bbox, "det_roi": det_roi, "n_dets": n_dets, "det_packed": packed}


def split_detections(packed, rows=None):
    """Views (n_dets, det_bbox, det_cls, det_prob, det_roi) into a `det_packed` buffer (device tensor or its host copy)."""
    rows = (packed.numel() - 4) // 7 if rows is None else rows
    return (packed[0:1], packed[4:4 + 4 * rows].view(rows, 4), packed[4 + 4 * rows:4 + 5 * rows],
            packed[4 + 5 * rows:4 + 6 * rows].view(torch.float32), packed[4 + 6 * rows:4 + 7 * rows])


# ----------------------------------------------------------------------------- conv backward
_DESC_CACHE = {}


def _conv_desc(x_shape, kh, kw, cout, stride, padding, act=0, layout=0, tile=0):
    """frcnn_conv_desc of a launch.  Descriptors are CACHED by their defining tuple and shared between calls (a training step
    builds ~90 of them from the same few dozen shapes; constructing the 17-field ctypes structure and asking the library for
    its workspace size cost ~4 us per launch of a host path that bounds the mixed-precision step): callers must not write
    to the returned object.  ``d._ws`` memoises the workspace-size queries made with it."""
    key = (x_shape, kh, kw, cout, stride, padding, act, layout, tile)
    d = _DESC_CACHE.get(key)
    if d is not None:
        return d
    if layout:
        h, w, n, cin = x_shape
    else:
        n, h, w, cin = x_shape
    if padding == "same":
        ho, pt = same_pad(h, kh, stride)
        wo, pl = same_pad(w, kw, stride)
    else:
        ho, wo, pt, pl = valid_out(h, kh, stride), valid_out(w, kw, stride), 0, 0
    d = _lib.ConvDesc(n=n, h=h, w=w, cin=cin, cout=cout, kh=kh, kw=kw, stride=stride, pad_top=pt, pad_left=pl,
                      ho=ho, wo=wo, act=act, ldy=0, ldres=0, tile=tile, layout=layout)
    d._ws = {}
    if len(_DESC_CACHE) > 4096:
        _DESC_CACHE.clear()
    _DESC_CACHE[key] = d
    return d


def _ws_need(d, query):
    """Workspace bytes of descriptor ``d`` under the library query ``query`` (memoised on cached descriptors)."""
    memo = getattr(d, "_ws", None)
    if memo is None:
        return getattr(_lib.load(), query)(ctypes.byref(d))
    need = memo.get(query)
    if need is None:
        need = memo[query] = getattr(_lib.load(), query)(ctypes.byref(d))
    return need


class PackedDgrad:
    """Filter of the input-gradient convolution of a stride-1 layer (transposed, flipped, with the
    forward epilogue scale folded in).  w_hwio: device or host (kh,kw,cin,cout); scale: (cout,) or None."""

    def __init__(self, w_hwio, scale=None):
        _require_gpu()
        w = _dev(w_hwio, torch.float32)
        self.kh, self.kw, cin, cout = (int(v) for v in w.shape)
        self.cin, self.cout = cout, cin                      # geometry of the dgrad convolution
        kp = _lib.load().frcnn_conv_packed_k(self.kh, self.kw, cout)
        self.w = torch.empty((cin, kp), dtype=torch.float32, device="cuda")
        sc = None if scale is None else _dev(scale, torch.float32)
        _lib.call("frcnn_pack_conv_weights_dgrad", _p(w), _p(sc), self.kh, self.kw, cin, cout, _p(self.w), _stream())
        self.scale = self.shift = None
        self._h3_spread_log2 = _channel_spread_log2(w if sc is None else w * sc, 3)   # (the transposed convolution reduces over the forward OUTPUT channels)

    x6_planes = PackedConv.x6_planes                         # the same [rows][packed k] f32 layout: the same three-plane form
    h3_planes = PackedConv.h3_planes                         # ... and the same header + two fp16 planes


def conv2d_dgrad(gy, pd, padding="valid", residual=None, mask=None, out=None):
    """Gradient w.r.t. the input of a stride-1 conv.  gy: (n,ho,wo,cout_fwd) gradient w.r.t. the layer's
    post-BN pre-activation output; residual: gradient arriving over an identity shortcut; mask: forward
    activation (n,h,w,cin_fwd) whose ReLU sits in front of this layer's input (None = no ReLU)."""
    _require_gpu()
    n, ho, wo, _ = gy.shape
    # forward 'same' (stride 1, odd kernel) pads (k-1)/2 on both sides -> so does the transposed conv
    pt = (pd.kh - 1) // 2 if padding == "same" else 0
    pl = (pd.kw - 1) // 2 if padding == "same" else 0
    assert padding == "same" or (pd.kh == 1 and pd.kw == 1), "dgrad supports 1x1 valid and odd 'same' kernels"
    if out is None:
        out = torch.empty((n, ho, wo, pd.cout), dtype=torch.float32, device="cuda")
    d = _conv_desc((n, ho, wo, pd.cin), pd.kh, pd.kw, pd.cout, 1, padding, 0, 0, 0)      # (see conv2d_dgrad_bf16)
    assert (d.pad_top, d.pad_left, d.ho, d.wo) == (pt, pl, ho, wo)
    gy = gy.contiguous()
    eng = _split_engine(d, pd, 0)                               # the split engines' policy, as for a forward launch of this geometry
    if eng == "x6":
        ws = _split_k_ws(_ws_need(d, "frcnn_conv2d_x6_workspace_bytes"))
        _lib.call("frcnn_conv2d_fwd_x6", ctypes.byref(d), _p(gy), _p(pd.x6_planes()), None, None, _p(residual), _p(mask), _p(out),
                  _p(ws), ws.numel() if ws is not None else 0, _stream())
        return out
    if eng == "h3":                                             # the gradient's magnitude record: its producer's (the dgrad launch behind it) or measured
        ws = _split_k_ws(_ws_need(d, "frcnn_conv2d_h3_workspace_bytes"))
        ya, ga = _amax_new(), amax_of(gy)
        _lib.call("frcnn_conv2d_fwd_h3", ctypes.byref(d), _p(gy), _p(ga), _p(pd.h3_planes()), None, None, _p(residual), _p(mask), _p(out), _p(ya),
                  _p(ws), ws.numel() if ws is not None else 0, _stream())
        out._amax = ya                                          # (residual and mask are applied before the epilogue takes the maximum)
        return out
    ya = _amax_new(optional=True) if (_tracking() and pd.cout >= 32) else None
    _conv_launch(d, gy, pd.w, None, None, residual, mask, out, ya)
    if ya is not None or getattr(out, "_amax", None) is not None:
        out._amax = ya
    return out


# Weight gradients of f32 layers: "native" = v_mfma_f32_32x32x2_f32; "bf16x6" = both operands split exactly into three bf16 pieces
# inside the kernel, six partial products on the bf16 matrix cores (conv_igemm.hip wgrad_body_x6_big; layers with cin, cout >= 128,
# the others stay native).  Same slabs and fixed-order reduction: runs are bitwise reproducible either way, the two engines agree
# to f32 rounding.  The library default stays "native"; train.py opts in.
WGRAD_ENGINE = os.environ.get("FRCNN_WGRAD_ENGINE", "native")


def _wgrad_tile(dtype):
    return 71 if WGRAD_ENGINE == "bf16x6" and dtype == torch.float32 else 0


def conv2d_wgrad(x, g, kh, kw, stride=1, padding="valid", scale=None, dw=None, dbias=None, want_bias=True):
    """-> (dw (kh,kw,cin,cout), dbias (cout,) or None); g: (n,ho,wo,cout)."""
    _require_gpu()
    cout = g.shape[-1]
    d = _conv_desc(tuple(x.shape), kh, kw, cout, stride, padding, tile=_wgrad_tile(x.dtype))
    assert (d.ho, d.wo) == (g.shape[1], g.shape[2])
    if dw is None:
        dw = torch.empty((kh, kw, x.shape[-1], cout), dtype=torch.float32, device="cuda")
    if dbias is None and want_bias:
        dbias = torch.empty(cout, dtype=torch.float32, device="cuda")
    ws = _ws(_lib.load().frcnn_conv2d_wgrad_workspace_bytes(ctypes.byref(d)))
    _lib.call("frcnn_conv2d_wgrad", ctypes.byref(d), _p(x.contiguous()), _p(g.contiguous()), _p(scale), _p(dw), _p(dbias if want_bias else None),
              _p(ws), ws.numel(), _stream())
    return dw, (dbias if want_bias else None)


def conv2d_wgrad_batch(jobs):
    """Weight gradients of many layers in one go (frcnn_conv2d_wgrad_batch).  jobs: list of
    (x, g, kh, kw, stride, padding, scale or None, dw) -- x / g both f32 or both bf16, dw an f32 (kh,kw,cin,cout) tensor the
    result is written into.  Bit-identical to conv2d_wgrad / conv2d_wgrad_bf16 per layer (without the bias gradient)."""
    _require_gpu()
    if not jobs:
        return
    arr = (_lib.WgradJob * len(jobs))()
    keep = []
    for j, (x, g, kh, kw, stride, padding, scale, dw) in zip(arr, jobs):
        assert x.dtype == g.dtype and x.dtype in (torch.float32, torch.bfloat16) and dw.dtype == torch.float32 and dw.is_contiguous()
        x, g = x.contiguous(), g.contiguous()
        keep.append((x, g))
        d = _conv_desc(tuple(x.shape), kh, kw, g.shape[-1], stride, padding, tile=_wgrad_tile(x.dtype))
        assert (d.ho, d.wo) == (g.shape[1], g.shape[2]) and tuple(dw.shape[-4:]) == (kh, kw, x.shape[-1], g.shape[-1])
        j.d = d
        j.x, j.g, j.dw = x.data_ptr(), g.data_ptr(), dw.data_ptr()
        j.scale = None if scale is None else scale.data_ptr()
        j.in_bf16 = 1 if x.dtype == torch.bfloat16 else 0
    ws = _ws(_lib.load().frcnn_conv2d_wgrad_batch_workspace_bytes(arr, len(jobs)))
    _lib.call("frcnn_conv2d_wgrad_batch", arr, len(jobs), _p(ws), ws.numel(), _stream())
    return ws, keep                                              # (a capturing caller keeps the launch's buffers from being recycled)


# ----------------------------------------------------------------------------- bf16 conv path
class PackedConvBf16:
    """bf16-packed filter + f32 epilogue scale/shift (configs[3]: bf16 conv)."""

    def __init__(self, w_hwio, scale=None, shift=None):
        _require_gpu()
        w = _dev(w_hwio, torch.float32)
        self.kh, self.kw, self.cin, self.cout = (int(v) for v in w.shape)
        self.w = torch.empty((self.cout, self.kh * self.kw * self.cin), dtype=torch.bfloat16, device="cuda")
        _lib.call("frcnn_pack_conv_weights_bf16", _p(w), self.kh, self.kw, self.cin, self.cout, _p(self.w), _stream())
        self.scale = None if scale is None else _dev(scale, torch.float32)
        self.shift = None if shift is None else _dev(shift, torch.float32)


class PackedDgradBf16:
    """bf16 filter of the input-gradient convolution of a stride-1 layer (PackedDgrad's bf16 twin; cout % 64 == 0).
    Filled by frcnn_refresh_packed_bf16."""

    def __init__(self, w_hwio, scale=None):
        _require_gpu()
        w = _dev(w_hwio, torch.float32)
        self.kh, self.kw, cin, cout = (int(v) for v in w.shape)
        self.cin, self.cout = cout, cin                      # geometry of the dgrad convolution
        self.w = torch.empty((cin, self.kh * self.kw * cout), dtype=torch.bfloat16, device="cuda")
        self.scale = self.shift = None
        sc = None if scale is None else _dev(scale, torch.float32)
        job = _lib.PackJob(w_hwio=w.data_ptr(), packed=None, packed_dgrad=self.w.data_ptr(), bias=None,
                           scale=None if sc is None else sc.data_ptr(), shift_const=None, shift=None,
                           kh=self.kh, kw=self.kw, cin=cin, cout=cout)
        _lib.call("frcnn_refresh_packed_bf16", ctypes.byref(job), 1, _stream())
        torch.cuda.current_stream().synchronize()            # w / sc may be temporaries


def conv2d_dgrad_bf16(gy, pd, padding="valid", residual=None, mask=None):
    """bf16 twin of conv2d_dgrad: gy (n,ho,wo,cout_fwd) bf16 -> gradient w.r.t. the layer input, bf16."""
    _require_gpu()
    assert gy.dtype == torch.bfloat16
    n, ho, wo, _ = gy.shape
    pt = (pd.kh - 1) // 2 if padding == "same" else 0
    pl = (pd.kw - 1) // 2 if padding == "same" else 0
    assert padding == "same" or (pd.kh == 1 and pd.kw == 1), "dgrad supports 1x1 valid and odd 'same' kernels"
    out = torch.empty((n, ho, wo, pd.cout), dtype=torch.bfloat16, device="cuda")
    # (stride 1 with 'same' padding of an odd kernel, or 1x1 'valid': the forward descriptor of that geometry IS the dgrad one)
    d = _conv_desc((n, ho, wo, pd.cin), pd.kh, pd.kw, pd.cout, 1, padding, 0, 0, AUTO_TILE)
    assert (d.pad_top, d.pad_left, d.ho, d.wo) == (pt, pl, ho, wo)
    ws = _split_k_ws(_ws_need(d, "frcnn_conv2d_workspace_bytes_bf16"))
    _lib.call("frcnn_conv2d_fwd_bf16_masked", ctypes.byref(d), _p(gy.contiguous()), _p(pd.w), None, None, _p(residual), _p(mask), _p(out), 0,
              _p(ws), ws.numel() if ws is not None else 0, _stream())
    return out


def conv2d_wgrad_bf16(x, g, kh, kw, stride=1, padding="valid", scale=None, dw=None, dbias=None, want_bias=True):
    """bf16 activations x (n,h,w,cin) and gradients g (n,ho,wo,cout) -> f32 (dw (kh,kw,cin,cout), dbias or None)."""
    _require_gpu()
    assert x.dtype == torch.bfloat16 and g.dtype == torch.bfloat16
    cout = g.shape[-1]
    d = _conv_desc(tuple(x.shape), kh, kw, cout, stride, padding)
    assert (d.ho, d.wo) == (g.shape[1], g.shape[2])
    if dw is None:
        dw = torch.empty((kh, kw, x.shape[-1], cout), dtype=torch.float32, device="cuda")
    if dbias is None and want_bias:
        dbias = torch.empty(cout, dtype=torch.float32, device="cuda")
    ws = _ws(_lib.load().frcnn_conv2d_wgrad_workspace_bytes(ctypes.byref(d)))
    _lib.call("frcnn_conv2d_wgrad_bf16", ctypes.byref(d), _p(x.contiguous()), _p(g.contiguous()), _p(scale), _p(dw), _p(dbias if want_bias else None),
              _p(ws), ws.numel(), _stream())
    return dw, (dbias if want_bias else None)


def cast_f32(x):
    """bf16 device tensor -> f32."""
    _require_gpu()
    assert x.dtype == torch.bfloat16
    out = torch.empty(x.shape, dtype=torch.float32, device="cuda")
    _lib.call("frcnn_cast_bf16_to_f32", _p(x.contiguous()), x.numel(), _p(out), _stream())
    return out


def roi_crop_resize_bwd_bf16(dout, rois, rows, cols):
    _require_gpu()
    n, pool, _, C = dout.shape
    dfeat = torch.empty((rows, cols, C), dtype=torch.float32, device="cuda")
    _lib.call("frcnn_roi_crop_resize_bwd_bf16", _p(dout.contiguous()), rows, cols, C, _p(rois.reshape(-1, 4).contiguous()), n, pool, _p(dfeat), _stream())
    return dfeat


def conv2d_bf16(x, pc, stride=1, padding="valid", act=None, residual=None, out_f32=False, tile=0, layout=0):
    """x: (n,h,w,cin) bf16 NHWC -> (n,ho,wo,cout) bf16 (or f32 when out_f32); layout=1: (h,w,n,cin) -> (ho,wo,n,cout)."""
    _require_gpu()
    assert x.dtype == torch.bfloat16 and x.is_contiguous() and x.shape[-1] == pc.cin
    d = _conv_desc(tuple(x.shape), pc.kh, pc.kw, pc.cout, stride, padding, ACT[act], layout, tile or AUTO_TILE)
    oshape = (d.ho, d.wo, d.n, pc.cout) if layout else (d.n, d.ho, d.wo, pc.cout)
    out = torch.empty(oshape, dtype=torch.float32 if out_f32 else torch.bfloat16, device="cuda")
    if residual is not None:
        assert residual.dtype == torch.bfloat16 and residual.shape == out.shape and residual.is_contiguous()
    ws = _split_k_ws(_ws_need(d, "frcnn_conv2d_workspace_bytes_bf16"))
    args = (ctypes.byref(d), _p(x), _p(pc.w), _p(pc.scale), _p(pc.shift), _p(residual), _p(out), 1 if out_f32 else 0,
            _p(ws), ws.numel() if ws is not None else 0)
    _lib.call("frcnn_conv2d_fwd_bf16_ws", *args, _stream())
    if CONV_PROFILE is not None:
        flops = 2.0 * d.n * d.ho * d.wo * pc.cout * pc.kh * pc.kw * pc.cin
        keep = (d, x, pc, residual, out, ws)
        CONV_PROFILE.append({"kernel": "k_conv_igemm_bf16", "flops": flops, "shape": (d.n * d.ho * d.wo, pc.cout, pc.kh * pc.kw * pc.cin, stride),
                             "relaunch": lambda args=args, keep=keep: _lib.call("frcnn_conv2d_fwd_bf16_ws", *args, _stream())})
    return out


def cast_bf16(x):
    _require_gpu()
    out = torch.empty(x.shape, dtype=torch.bfloat16, device="cuda")
    _lib.call("frcnn_cast_f32_to_bf16", _p(x.contiguous()), x.numel(), _p(out), _stream())
    return out


def avgpool_bf16(x, k, layout=0):
    """(n,k,k,c) bf16 (or (k,k,n,c) with layout=1) -> (n,c) f32."""
    _require_gpu()
    n, c = (x.shape[2] if layout else x.shape[0]), x.shape[3]
    out = torch.empty((n, c), dtype=torch.float32, device="cuda")
    _lib.call("frcnn_avgpool_bf16_to_f32_ex", _p(x.contiguous()), n, k, c, layout, _p(out), _stream())
    return out


def roi_crop_resize_bf16(feat, rois, pool, fill=None, relu=False, layout=0):
    _require_gpu()
    feat = feat.reshape(feat.shape[-3], feat.shape[-2], feat.shape[-1]).contiguous()
    rows, cols, C = feat.shape
    rois = rois.reshape(-1, 4).to(torch.float32).contiguous()
    n = rois.shape[0]
    out = torch.empty((pool, pool, n, C) if layout else (n, pool, pool, C), dtype=torch.bfloat16, device="cuda")
    _lib.call("frcnn_roi_crop_resize_fwd_bf16_ex", _p(feat), rows, cols, C, _p(rois), n, pool, _p(fill), 1 if relu else 0, layout, _p(out), _stream())
    return out


def roi_crop_resize_bf16_batch(feat, rois, n_per_img, pool, fill=None, relu=False, layout=0):
    """RoiResizeConv for the RoIs of several images in one launch: feat (B,R,C,Cf) bf16, rois (B*n_per_img,4) f32 (image =
    row // n_per_img) -> (B*n,pool,pool,Cf) bf16, or (pool,pool,B*n,Cf) with layout=1."""
    _require_gpu()
    assert feat.dtype == torch.bfloat16 and feat.dim() == 4 and feat.is_contiguous()
    B, rows, cols, C = feat.shape
    rois = rois.reshape(-1, 4).to(torch.float32).contiguous()
    n = rois.shape[0]
    assert n == B * n_per_img
    out = torch.empty((pool, pool, n, C) if layout else (n, pool, pool, C), dtype=torch.bfloat16, device="cuda")
    _lib.call("frcnn_roi_crop_resize_fwd_bf16_batch", _p(feat), B, rows, cols, C, _p(rois), n_per_img, pool, _p(fill), 1 if relu else 0, layout, _p(out), _stream())
    return out


class PackedStemBf16:
    """The fused bf16 stem's parameters (frcnn_stem_bf16_fwd): conv1's 7x7x3x64 filter packed to bf16 [64][176] and the
    folded f32 scale / shift of bias + BatchNorm (+ Scale)."""

    def __init__(self, w_hwio, scale, shift):
        _require_gpu()
        w = _dev(w_hwio, torch.float32)
        assert tuple(w.shape) == (7, 7, 3, 64), "the fused stem is conv1 of the ResNets: 7x7x3 -> 64"
        self.w = torch.empty(_lib.load().frcnn_stem_bf16_packed_elems(), dtype=torch.bfloat16, device="cuda")
        _lib.call("frcnn_pack_stem_weights_bf16", _p(w), _p(self.w), _stream())
        self.scale, self.shift = _dev(scale, torch.float32), _dev(shift, torch.float32)
        torch.cuda.current_stream().synchronize()            # w may be a temporary


def stem_bf16(x, ps):
    """(n,H,W,3) f32 preprocessed images -> (n,Hp,Wp,64) bf16: conv1 + BN (+Scale) + ReLU + 3x3/2 max-pool, one launch."""
    _require_gpu()
    assert x.dtype == torch.float32 and x.is_contiguous() and x.dim() == 4 and x.shape[-1] == 3
    n, h, w, _ = x.shape
    ho, wo = (h + 1) // 2, (w + 1) // 2
    out = torch.empty((n, (ho - 3) // 2 + 1, (wo - 3) // 2 + 1, 64), dtype=torch.bfloat16, device="cuda")
    _lib.call("frcnn_stem_bf16_fwd", _p(x), n, h, w, _p(ps.w), _p(ps.scale), _p(ps.shift), _p(out), _stream())
    return out


class PackedStemH3:
    """The fused f32 stem's parameters (frcnn_stem_h3_fwd): conv1's 7x7x3x64 filter as a header + two fp16 planes [2][64][176] and the
    folded f32 scale / shift of bias + BatchNorm (+ Scale)."""

    def __init__(self, w_hwio, scale, shift):
        _require_gpu()
        w = _dev(w_hwio, torch.float32)
        assert tuple(w.shape) == (7, 7, 3, 64), "the fused stem is conv1 of the ResNets: 7x7x3 -> 64"
        self.w = torch.empty(_lib.load().frcnn_stem_h3_packed_bytes(), dtype=torch.uint8, device="cuda")
        _lib.call("frcnn_pack_stem_weights_h3", _p(w), _p(self.w), _stream())
        self.scale, self.shift = _dev(scale, torch.float32), _dev(shift, torch.float32)
        torch.cuda.current_stream().synchronize()            # w may be a temporary


def stem_h3(x, ps):
    """(n,H,W,3) f32 preprocessed images -> (n,Hp,Wp,64) f32: conv1 + BN (+Scale) + ReLU + 3x3/2 max-pool in ONE f16x3 launch; the
    result carries its magnitude record."""
    _require_gpu()
    assert x.dtype == torch.float32 and x.is_contiguous() and x.dim() == 4 and x.shape[-1] == 3
    n, h, w, _ = x.shape
    ho, wo = (h + 1) // 2, (w + 1) // 2
    out = torch.empty((n, (ho - 3) // 2 + 1, (wo - 3) // 2 + 1, 64), dtype=torch.float32, device="cuda")
    ya, xa = _amax_new(), amax_of(x)
    args = (_p(x), _p(xa), n, h, w, _p(ps.w), _p(ps.scale), _p(ps.shift), _p(out), _p(ya))
    _lib.call("frcnn_stem_h3_fwd", *args, _stream())
    out._amax = ya
    if CONV_PROFILE is not None:
        keep = (x, ps, out, ya, xa)
        CONV_PROFILE.append({"kernel": "k_stem_h3", "flops": 2.0 * n * ho * wo * 64 * 147, "shape": (n * ho * wo, 64, 147, 2),
                             "relaunch": lambda args=args, keep=keep: _lib.call("frcnn_stem_h3_fwd", *args, _stream())})
    return out
