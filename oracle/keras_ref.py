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
