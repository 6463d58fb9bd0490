"""torch/numpy CPU restatement of the reference's Keras half (resnet.py, vgg.py,
custom_layers.py, loss_functions.py).  TEST INFRASTRUCTURE ONLY (oracle/__init__.py).

PARITY UNPINNED: Keras 2.0.8 / TensorFlow 1.3 cannot be installed here and the reference's
golden .h5 files are missing from its checkout, so nothing below can be
checked against the reference's own outputs.  Third-party semantics ([3P] in SURVEY
Appendix A) are restated from the published TF-1.3 / Keras-2.0.8 algorithms and cited at
the reference call site that invokes them.
"""
import numpy as np

f32 = np.float32


# --------------------------------------------------------------------------- RoI crop + resize
def roi_resize(feat, rois, pool=7):
    """custom_layers.RoiResizeConv.call (custom_layers.py:35-56).
    feat (R,C,Cf) f32; rois (n,4) f32 -> (n,pool,pool,Cf) f32.
    int32 truncation of the corners (:45-48), crop [y1:y2, x1:x2] (:50), then TF-1.3
    tf.image.resize_images bilinear, align_corners=False [3P: resize_bilinear_op.cc,
    compute_interpolation_weights + compute_lerp]: scale = in/out in f32, src = i*scale,
    lo = int(src), hi = min(lo+1, in-1), lerp = src-lo; top/bottom lerp in x then y."""
    feat = np.asarray(feat, dtype=f32)
    out = np.zeros((len(rois), pool, pool, feat.shape[2]), dtype=f32)
    for r, roi in enumerate(np.asarray(rois)):
        x1, y1, x2, y2 = (int(v) for v in roi)           # K.cast(., 'int32') truncates toward zero
        crop = feat[y1:y2, x1:x2, :]
        h, w = crop.shape[:2]
        if h == 0 or w == 0:
            continue
        sy, sx = f32(h) / f32(pool), f32(w) / f32(pool)
        for py in range(pool):
            fy = f32(py) * sy
            ylo = int(fy)
            yhi = min(ylo + 1, h - 1)
            ty = f32(fy - f32(ylo))
            for px in range(pool):
                fx = f32(px) * sx
                xlo = int(fx)
                xhi = min(xlo + 1, w - 1)
                tx = f32(fx - f32(xlo))
                tl, tr, bl, br = crop[ylo, xlo], crop[ylo, xhi], crop[yhi, xlo], crop[yhi, xhi]
                top = tl + (tr - tl) * tx
                bot = bl + (br - bl) * tx
                out[r, py, px] = top + (bot - top) * ty
    return out


# --------------------------------------------------------------------------- conv / BN / pool
def _torch():
    import torch
    return torch


def same_pad(size, k, stride):
    """Keras padding='same' -> TF SAME [3P]: out = ceil(in/stride); pad_total =
    max((out-1)*stride + k - in, 0); pad_before = pad_total // 2 (extra pixel at the end)."""
    out = -(-size // stride)
    total = max((out - 1) * stride + k - size, 0)
    return out, total // 2, total - total // 2


def conv2d(x, w_hwio, bias=None, stride=1, padding="valid", dtype=None):
    """Keras Conv2D on NHWC input with HWIO kernel (resnet.py:150, 408; vgg.py:96).
    x: (n,h,w,cin) tensor/array; returns NHWC torch tensor (CPU)."""
    torch = _torch()
    import torch.nn.functional as F
    dtype = dtype or torch.float32
    x = torch.as_tensor(np.asarray(x)).to(dtype).permute(0, 3, 1, 2)
    w = torch.as_tensor(np.asarray(w_hwio)).to(dtype).permute(3, 2, 0, 1)
    if padding == "same":
        _, pt, pb = same_pad(x.shape[2], w.shape[2], stride)
        _, pl, pr = same_pad(x.shape[3], w.shape[3], stride)
        x = F.pad(x, (pl, pr, pt, pb))
    b = None if bias is None else torch.as_tensor(np.asarray(bias)).to(dtype)
    y = F.conv2d(x, w, b, stride=stride)
    return y.permute(0, 2, 3, 1).contiguous()


def batchnorm_inference(x, gamma, beta, mean, var, eps):
    """BatchNormalization(training=False) [3P tf.nn.batch_normalization]:
    inv = rsqrt(var+eps)*gamma; y = x*inv + (beta - mean*inv)."""
    torch = _torch()
    g, b, m, v = (torch.as_tensor(np.asarray(t)).to(x.dtype) for t in (gamma, beta, mean, var))
    inv = torch.rsqrt(v + eps) * g
    return x * inv + (b - m * inv)


def pool2d(x, k, stride, is_max=True):
    torch = _torch()
    import torch.nn.functional as F
    xc = x.permute(0, 3, 1, 2)
    y = F.max_pool2d(xc, k, stride) if is_max else F.avg_pool2d(xc, k, stride)
    return y.permute(0, 2, 3, 1).contiguous()
