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
    lo = int(src), hi = min(lo+1, in-1), lerp = src-lo; top/bottom lerp in x then y.
    ``pool`` may be (rows, cols) -- the reference only uses squares; the non-square form exists for TF's own published
    resize vector (tests/test_oracle_kats.py)."""
    feat = np.asarray(feat, dtype=f32)
    pool_h, pool_w = (pool, pool) if np.isscalar(pool) else pool
    out = np.zeros((len(rois), pool_h, pool_w, feat.shape[2]), dtype=f32)
    for r, roi in enumerate(np.asarray(rois)):
        x1, y1, x2, y2 = (int(v) for v in roi)           # K.cast(., 'int32') truncates toward zero
        crop = feat[y1:y2, x1:x2, :]
        h, w = crop.shape[:2]
        if h == 0 or w == 0:
            continue
        sy, sx = f32(h) / f32(pool_h), f32(w) / f32(pool_w)
        for py in range(pool_h):
            fy = f32(py) * sy
            ylo = int(fy)
            yhi = min(ylo + 1, h - 1)
            ty = f32(fy - f32(ylo))
            for px in range(pool_w):
                fx = f32(px) * sx
                xlo = int(fx)
                xhi = min(xlo + 1, w - 1)
                tx = f32(fx - f32(xlo))
                tl, tr, bl, br = crop[ylo, xlo], crop[ylo, xhi], crop[yhi, xlo], crop[yhi, xhi]
                top = tl + (tr - tl) * tx
                bot = bl + (br - bl) * tx
                out[r, py, px] = top + (bot - top) * ty
    return out


def roi_resize_torch(feat, rois, pool=7):
    """roi_resize on a torch tensor (R,C,Cf), differentiable w.r.t. feat; same tap/lerp arithmetic."""
    torch = _torch()
    outs = []
    for roi in np.asarray(rois):
        x1, y1, x2, y2 = (int(v) for v in roi)
        crop = feat[y1:y2, x1:x2, :]
        h, w = crop.shape[:2]
        sy, sx = f32(h) / f32(pool), f32(w) / f32(pool)
        ys = [f32(i) * sy for i in range(pool)]
        xs = [f32(i) * sx for i in range(pool)]
        ylo = [int(v) for v in ys]; xlo = [int(v) for v in xs]
        yhi = [min(v + 1, h - 1) for v in ylo]; xhi = [min(v + 1, w - 1) for v in xlo]
        ty = torch.tensor([float(f32(a - f32(b))) for a, b in zip(ys, ylo)], dtype=feat.dtype).view(pool, 1, 1)
        tx = torch.tensor([float(f32(a - f32(b))) for a, b in zip(xs, xlo)], dtype=feat.dtype).view(1, pool, 1)
        tl = crop[ylo][:, xlo]; tr = crop[ylo][:, xhi]; bl = crop[yhi][:, xlo]; br = crop[yhi][:, xhi]
        top = tl + (tr - tl) * tx
        bot = bl + (br - bl) * tx
        outs.append(top + (bot - top) * ty)
    return torch.stack(outs)


# --------------------------------------------------------------------------- conv / BN / pool
def _torch():
    import torch
    return torch


def _t(a, dtype):
    """array or tensor -> tensor of dtype (tensors keep their autograd history)."""
    torch = _torch()
    if isinstance(a, torch.Tensor):
        return a.to(dtype)
    return torch.as_tensor(np.asarray(a)).to(dtype)


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
    x = _t(x, dtype).permute(0, 3, 1, 2)
    w = _t(w_hwio, dtype).permute(3, 2, 0, 1)
    if padding == "same":
        _, pt, pb = same_pad(x.shape[2], w.shape[2], stride)
        _, pl, pr = same_pad(x.shape[3], w.shape[3], stride)
        x = F.pad(x, (pl, pr, pt, pb))
    b = None if bias is None else _t(bias, dtype)
    y = F.conv2d(x, w, b, stride=stride)
    return y.permute(0, 2, 3, 1).contiguous()


def batchnorm_inference(x, gamma, beta, mean, var, eps):
    """BatchNormalization(training=False) [3P tf.nn.batch_normalization]:
    inv = rsqrt(var+eps)*gamma; y = x*inv + (beta - mean*inv)."""
    torch = _torch()
    g, b, m, v = (_t(t, x.dtype) for t in (gamma, beta, mean, var))
    inv = torch.rsqrt(v + eps) * g
    return x * inv + (b - m * inv)


def pool2d(x, k, stride, is_max=True):
    torch = _torch()
    import torch.nn.functional as F
    xc = x.permute(0, 3, 1, 2)
    y = F.max_pool2d(xc, k, stride) if is_max else F.avg_pool2d(xc, k, stride)
    return y.permute(0, 2, 3, 1).contiguous()


# --------------------------------------------------------------------------- bf16 storage model (mixed precision)
def _bf16_round(t):
    torch = _torch()
    return t.to(torch.bfloat16).to(t.dtype)


def _make_quantisers():
    """(q, qw) for the mixed-precision model of BASELINE configs[3]/[4] ("bf16 conv", "mixed bf16"): the product keeps
    activations, activation gradients and the packed filters of its ResNet blocks in bf16 (one round-to-nearest-even
    at every store) while it accumulates, folds BatchNorm, and keeps master weights / weight gradients in f32.
      q(t)  -- a STORED activation: rounds the value going forward and the gradient coming back (the input-gradient
               kernels store bf16 too);
      qw(w) -- a packed filter: rounded going forward, gradient passed through unrounded (the weight gradient is
               accumulated in f32 from the bf16 operands and applied to the f32 master copy)."""
    torch = _torch()

    class _Q(torch.autograd.Function):
        @staticmethod
        def forward(ctx, t):
            return _bf16_round(t)

        @staticmethod
        def backward(ctx, g):
            return _bf16_round(g)

    class _QW(torch.autograd.Function):
        @staticmethod
        def forward(ctx, t):
            return _bf16_round(t)

        @staticmethod
        def backward(ctx, g):
            return g

    return _Q.apply, _QW.apply


# --------------------------------------------------------------------------- Keras graphs
class KerasGraphs:
    """Layer-by-layer restatement of the reference's model builders, evaluated with torch on
    the CPU in ``dtype`` (float32 = "the Keras CPU path"; float64 = tolerance anchor).
    ``weights`` = {keras_layer_name: [arrays in get_weights() order]}.

    ``mixed=True`` evaluates the SAME graph under the bf16 storage model above (every ``self.q`` / ``self.qw`` below is
    the identity otherwise): the comparand for the product's bf16 / mixed-precision paths, so that what is left between
    the two is accumulation order, not storage precision."""

    def __init__(self, weights, dtype=None, mixed=False, bf16_stem=None):
        torch = _torch()
        self.w = weights
        self.dtype = dtype or torch.float32
        self.mixed = mixed
        # round 3: on the bf16 paths the product's stem is a bf16 conv too (frcnn_stem_bf16_fwd: pixels and conv1's taps
        # rounded to bf16, f32 accumulate, BatchNorm / ReLU in f32, one rounding, then the pool); False = the round-1/2
        # f32 stem whose pooled output is cast once
        self.bf16_stem = mixed if bf16_stem is None else bool(bf16_stem)
        if mixed:
            self.q, self.qw = _make_quantisers()
        else:
            self.q = self.qw = lambda t: t

    @staticmethod
    def _bf16_layer(name):
        """Layers whose activations / filters the product stores in bf16: every ResNet block and rpn_conv1.  The
        3-channel stem, the RPN output layers and the dense layers stay f32 (mixed precision)."""
        return name.startswith("res") or name == "rpn_conv1"

    # -- layers
    def conv(self, x, name, stride=1, padding="valid"):
        w = self.w[name]
        k = w[0]
        if self.mixed and (self._bf16_layer(name) or (self.bf16_stem and name == "conv1")):
            k = self.qw(_t(k, self.dtype))
        return conv2d(x, k, w[1] if len(w) > 1 else None, stride, padding, self.dtype)

    def bn(self, x, name, eps):
        return batchnorm_inference(x, *self.w[name], eps=eps)

    def scale(self, x, name):
        """custom_layers.Scale.call (custom_layers.py:121-129): gamma * x + beta."""
        torch = _torch()
        g, b = (_t(t, x.dtype) for t in self.w[name])
        return g * x + b

    def conv_bn(self, x, stage, block, suffix, stride=1, padding="valid", separate_scale=False):
        tag = "%d%s_branch%s" % (stage, block, suffix)
        x = self.conv(x, "res" + tag, stride, padding)
        x = self.bn(x, "bn" + tag, 1e-5)                                   # eps (resnet.py:148, 216)
        if separate_scale:
            x = self.scale(x, "scale" + tag)
        return x

    # -- blocks (resnet.py:114-176 identity_block, :179-247 conv_block; TD twins :250-392)
    def identity_block(self, x, stage, block, separate_scale=False):
        q = self.q                                          # stored activations (identity unless mixed)
        t = q(self.conv_bn(x, stage, block, "2a", separate_scale=separate_scale).clamp(min=0))
        t = q(self.conv_bn(t, stage, block, "2b", padding="same", separate_scale=separate_scale).clamp(min=0))
        t = self.conv_bn(t, stage, block, "2c", separate_scale=separate_scale)      # the add and the ReLU happen before the store
        return q((t + x).clamp(min=0))

    def conv_block(self, x, stage, block, stride=2, separate_scale=False):
        q = self.q
        t = q(self.conv_bn(x, stage, block, "2a", stride=stride, separate_scale=separate_scale).clamp(min=0))
        t = q(self.conv_bn(t, stage, block, "2b", padding="same", separate_scale=separate_scale).clamp(min=0))
        t = self.conv_bn(t, stage, block, "2c", separate_scale=separate_scale)
        s = q(self.conv_bn(x, stage, block, "1", stride=stride, separate_scale=separate_scale))
        return q((t + s).clamp(min=0))

    # -- bases
    def resnet_base(self, x, depth=50):
        """resnet50_base (resnet.py:395-448) / resnet101_base (:551-602). x: (1,H,W,3)."""
        r101 = depth == 101
        if self.mixed and self.bf16_stem:
            x = self.q(_t(x, self.dtype))                                   # the fused bf16 stem rounds the pixels once
        x = self.conv(x, "conv1", stride=2, padding="same")                 # :408
        x = self.bn(x, "bn_conv1", 1e-3)                                    # Keras default eps (:410)
        if r101:
            x = self.scale(x, "scale_conv1")                                # :567
        x = self.q(pool2d(x.clamp(min=0), 3, 2, True))                      # :411-412 (mixed: the f32 stem's output is cast once)
        if r101:
            plan = {2: ["a", "b", "c"], 3: ["a", "b1", "b2", "b3"], 4: ["a"] + ["b%d" % i for i in range(1, 23)]}
        else:
            plan = {2: list("abc"), 3: list("abcd"), 4: list("abcdef")}
        for stage in (2, 3, 4):
            for block in plan[stage]:
                if block == "a":
                    x = self.conv_block(x, stage, block, stride=1 if stage == 2 else 2, separate_scale=r101)
                else:
                    x = self.identity_block(x, stage, block, separate_scale=r101)
        return x

    def vgg_base(self, x):
        """vgg16_base (vgg.py:91-141): 3x3 same + relu, 2x2 pools after blocks 1-4."""
        for blk, n in ((1, 2), (2, 2), (3, 3), (4, 3), (5, 3)):
            for i in range(1, n + 1):
                x = self.conv(x, "block%d_conv%d" % (blk, i), padding="same").clamp(min=0)
            if blk < 5:
                x = pool2d(x, 2, 2, True)
        return x

    # -- heads
    def rpn(self, feat):
        """resnet50_rpn (resnet.py:464-474) / vgg16_rpn (vgg.py:171-185)."""
        torch = _torch()
        t = self.q(self.conv(feat, "rpn_conv1", padding="same").clamp(min=0))
        cls = torch.sigmoid(self.conv(t, "rpn_out_cls"))
        reg = self.conv(t, "rpn_out_bbreg")
        return cls, reg

    def _dense(self, x, name):
        torch = _torch()
        k, b = (_t(t, x.dtype) for t in self.w[name])
        return x @ k + b

    def resnet_classifier_logits(self, feat, rois, num_classes, depth=50):
        """Differentiable form of resnet_classifier (torch RoI resize): returns (softmax probs, reg)."""
        torch = _torch()
        r101 = depth == 101
        x = self.q(roi_resize_torch(feat[0], rois, 7))
        x = self.conv_block(x, 5, "a", stride=1, separate_scale=r101)
        x = self.identity_block(x, 5, "b", separate_scale=r101)
        x = self.identity_block(x, 5, "c", separate_scale=r101)
        x = pool2d(x, 7, 7, False).reshape(x.shape[0], -1)
        cls = torch.softmax(self._dense(x, "dense_class_%d" % num_classes), dim=1)
        reg = self._dense(x, "dense_reg_%d" % num_classes)
        return cls, reg

    def resnet_classifier(self, feat, rois, num_classes, depth=50, return_pooled=False):
        """resnet50_classifier (resnet.py:489-548): RoiResizeConv, stage 5 with strides (1,1)
        (:508), AveragePooling2D(7), flatten, dense softmax + dense linear.
        ``return_pooled``: also the (n, 2048) input of the two dense layers."""
        torch = _torch()
        r101 = depth == 101
        crops = roi_resize(np.asarray(feat[0].to(torch.float32)), np.asarray(rois), 7)
        x = self.q(torch.as_tensor(crops).to(self.dtype))                    # RoIs as batch
        x = self.conv_block(x, 5, "a", stride=1, separate_scale=r101)
        x = self.identity_block(x, 5, "b", separate_scale=r101)
        x = self.identity_block(x, 5, "c", separate_scale=r101)
        x = pool2d(x, 7, 7, False).reshape(x.shape[0], -1)
        cls = torch.softmax(self._dense(x, "dense_class_%d" % num_classes), dim=1)
        reg = self._dense(x, "dense_reg_%d" % num_classes)
        return (cls, reg, x) if return_pooled else (cls, reg)

    def vgg_classifier(self, feat, rois, num_classes):
        """vgg16_classifier (vgg.py:226-255)."""
        torch = _torch()
        crops = roi_resize(np.asarray(feat[0].to(torch.float32)), np.asarray(rois), 7)
        x = torch.as_tensor(crops).to(self.dtype).reshape(len(crops), -1)
        x = self._dense(x, "fc1").clamp(min=0)
        x = self._dense(x, "fc2").clamp(min=0)
        cls = torch.softmax(self._dense(x, "dense_class_%d" % num_classes), dim=1)
        reg = self._dense(x, "dense_reg_%d" % num_classes)
        return cls, reg
