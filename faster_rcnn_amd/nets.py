"""Lowering of the reference's Keras graphs (resnet.py, vgg.py) onto the HIP conv engine.

Every Keras ``Conv2D -> BatchNormalization(training=False) [-> Scale] [-> add] -> Activation``
group becomes ONE fused launch (``ConvUnit``): bias, BN and Scale fold into the epilogue's
per-channel scale/shift, the shortcut add and the activation run in registers.
``TimeDistributed`` blocks run with the RoI axis as the batch axis of the same kernel.
No tracing compiler: a forward pass is a fixed sequence of C-ABI calls on one HIP stream,
which ``InferenceGraph`` (pipeline.py) captures into a hipGraph.
"""
import os as _os

import numpy as np
import torch

from . import ops
from .weights import VGG_CONVS, resnet_block_names

BN_EPS_STEM = 1e-3      # Keras BatchNormalization default epsilon (bn_conv1, resnet.py:410)
BN_EPS_BLOCK = 1e-5     # resnet.py:148, 216, 280, 349


class ConvUnit:
    """One fused conv launch and the Keras layers it stands for."""

    def __init__(self, weights, conv, bn=None, scale=None, eps=BN_EPS_BLOCK, stride=1, padding="valid", act=None, tile=0,
                 dtype="f32", out_f32=False):
        self.weights, self.conv, self.bn, self.scale_name, self.eps = weights, conv, bn, scale, eps
        self.stride, self.padding, self.act, self.tile = stride, padding, act, tile
        self.dtype, self.out_f32 = dtype, out_f32           # "bf16": bf16 operands / activations (configs[3])
        self.pc = None

    def folded(self):
        """(HWIO kernel f32, per-channel scale, shift) with bias, BatchNorm and Scale folded (numpy)."""
        w = self.weights[self.conv]
        kernel = np.asarray(w[0], dtype=np.float32)
        if kernel.ndim == 2:                               # Dense kernel (in, out) == 1x1 conv
            kernel = kernel.reshape(1, 1, *kernel.shape)
        cout = kernel.shape[3]
        bias = np.asarray(w[1], dtype=np.float64) if len(w) > 1 else np.zeros(cout)
        scale = np.ones(cout)
        shift = bias.copy()
        if self.bn is not None:
            g, b, m, v = (np.asarray(a, dtype=np.float64) for a in self.weights[self.bn])
            inv = g / np.sqrt(v + self.eps)
            scale = inv
            shift = (bias - m) * inv + b
        if self.scale_name is not None:
            g2, b2 = (np.asarray(a, dtype=np.float64) for a in self.weights[self.scale_name])
            scale = scale * g2
            shift = shift * g2 + b2
        return kernel, scale.astype(np.float32), shift.astype(np.float32)

    def lower(self):
        packer = ops.PackedConvBf16 if self.dtype == "bf16" else ops.PackedConv
        self.pc = packer(*self.folded())
        return self

    def __call__(self, x, residual=None, out=None, act="unit", layout=0, planes_out=False):
        """``act`` overrides the unit's activation for this call ("unit" = keep it); ``layout`` 1 = position-major
        tensors (h,w,n,c), see frcnn_conv_desc.layout; ``planes_out``: ops.conv2d (f32 path: the result may come back as an
        ops.PlaneTensor for the next convolution)."""
        if self.pc is None:
            self.lower()
        act = self.act if act == "unit" else act
        if self.dtype == "bf16":
            return ops.conv2d_bf16(x, self.pc, self.stride, self.padding, act, residual, self.out_f32, self.tile, layout)
        return ops.conv2d(x, self.pc, self.stride, self.padding, act, residual, out, self.tile, layout, planes_out)


class DualUnit:
    """Two ConvUnits that read the SAME input with the same kernel size, stride and padding, as ONE launch
    (frcnn_conv2d_fwd_dual): the filters concatenated along the output axis, each layer with its own folded
    BatchNorm / bias vector slice, activation and output tensor.  conv_block's ``branch2a`` + shortcut ``branch1``
    (resnet.py:218-241) and rpn_out_cls + rpn_out_bbreg (resnet.py:464-474).  f32 path only; per output element the
    arithmetic is that of the two single-layer launches (same k order), so the results are bit-identical to them
    whenever those make the same split-K choice (tests/test_conv_gpu.py)."""

    def __init__(self, first, second):
        assert first.dtype == second.dtype == "f32" and (first.stride, first.padding) == (second.stride, second.padding)
        self.a, self.b, self.pc, self.n1 = first, second, None, None
        self._src = None

    def lower(self):
        a, b = self.a, self.b
        for u in (a, b):
            if u.pc is None:
                u.lower()
        # (the single-layer lowerings are kept: training and the reference-order paths launch the layers one by one)
        wa, wb = a.pc.w, b.pc.w
        pc = ops.PackedConv.__new__(ops.PackedConv)
        pc.kh, pc.kw, pc.cin, pc.cout = a.pc.kh, a.pc.kw, a.pc.cin, a.pc.cout + b.pc.cout
        assert (a.pc.kh, a.pc.kw, a.pc.cin) == (b.pc.kh, b.pc.kw, b.pc.cin)
        pc.w = torch.cat([wa, wb], dim=0).contiguous()                  # packed rows are per output channel: [cout][Kpad]
        pc.scale = torch.cat([a.pc.scale, b.pc.scale]).contiguous()
        pc.shift = torch.cat([a.pc.shift, b.pc.shift]).contiguous()
        self.pc, self.n1, self._src = pc, a.pc.cout, (a.pc, b.pc)
        return self

    def __call__(self, x, act1="unit", act2="unit", layout=0):
        if self.pc is None or self._src[0] is not self.a.pc or self._src[1] is not self.b.pc:
            self.lower()                                                # (a re-lowered or invalidated layer invalidates the pair)
        act1 = self.a.act if act1 == "unit" else act1
        act2 = self.b.act if act2 == "unit" else act2
        return ops.conv2d_dual(x, self.pc, self.n1, self.a.stride, self.a.padding, act1, act2, layout, self.a.tile)


FUSE_PAIRS = True       # dev knob (tests): False launches every layer on its own
FUSED_BF16_STEM = True  # dev knob (tests): False = f32 stem conv + f32 pool + cast in front of a bf16 trunk
FUSED_F32_STEM = True   # dev knob (tests): False = conv1 and the max-pool as two launches also under the f16x3 engine


def _pair(first, second):
    """The DualUnit of two f32 ConvUnits (cached on the first), or None when the pair cannot share a launch."""
    if not FUSE_PAIRS or first.dtype != "f32" or second.dtype != "f32":
        return None
    d = getattr(first, "_dual", None)
    if d is None or d.b is not second:
        w = first.weights[first.conv][0]
        if np.ndim(w) != 4 or np.shape(w)[2] % 32 != 0:
            return None
        d = first._dual = DualUnit(first, second)
    return d


def _block_units(weights, stage, block, has_shortcut, stride, separate_scale, dtype="f32"):
    def unit(suffix, **kw):
        tag = "%d%s_branch%s" % (stage, block, suffix)
        return ConvUnit(weights, "res" + tag, "bn" + tag, ("scale" + tag) if separate_scale else None, BN_EPS_BLOCK, dtype=dtype, **kw)
    u = {"2a": unit("2a", stride=stride, act="relu"),
         "2b": unit("2b", padding="same", act="relu"),
         "2c": unit("2c", act="relu")}                      # relu applied after the fused shortcut add
    if has_shortcut:
        u["1"] = unit("1", stride=stride)
    return u


HEAD_PLANES = True      # dev knob (tests): False keeps every tensor of the detector head f32
# round 6, OFF (FRCNN_HEAD_BLOCK_PLANES=1 switches it on): a head block's OUTPUT travels as planes too (next block's branch2a stages them,
# its closing 1x1 reads the shortcut back from them, frcnn_conv2d_fwd_h3_planes_res): the 2048 -> 512 layers lose the split in their
# loader and walk the ring -- 436-443 -> 393-407 us per four-image launch -- but the 512 -> 2048 layers that now write and read 8-byte plane
# pieces instead of 16-byte f32 pieces take 548-557 instead of 534-539 us: 533.5 against 536.5 img/s (scripts/dev/r6_block_planes_ab.sh)
HEAD_BLOCK_PLANES = _os.environ.get("FRCNN_HEAD_BLOCK_PLANES", "0") != "0"
VGG_PLANES = _os.environ.get("FRCNN_VGG_PLANES", "1") != "0"        # VggBase: plane tensors between the convolutions of a block (round 6)
# the same hand-over inside the TRUNK's bottleneck blocks, wherever the consuming launch is one that reads planes (the 256x128 tile
# forms) -- VERDICT r5 item 3.  Mid-round, with 64x64 tiles under 256 tiles of 128x128: nothing (540.1 / 538.7 against 540.5 / 538.6
# img/s).  Under the shared-chip tile policy (256x128 from 128 tiles on: stage 3 and stage 4 of a four-image pass qualify) and with the
# ring for the 3x3 layers' long reductions: 544.2 / 543.3 -> 547.1 / 547.9 img/s, backbone in flight 0.457 -> 0.449 ms per image
# (scripts/dev/r6_trunk_planes2.sh): +0.7 %, inside the boxes' spread.  OFF by default all the same (FRCNN_TRUNK_PLANES=1: on): with it
# the plane-reading kernel -- bench.py's dominant kernel, the head's six launches per pass, frac 0.39-0.40 -- also runs ~20 small trunk
# launches per pass, and the line's `roofline` would average a different set of launches than every round before (0.34 over them all).
TRUNK_PLANES = _os.environ.get("FRCNN_TRUNK_PLANES", "0") != "0"


class Extents:
    """The true extents of the images of a CANVAS pass (round 6; ops.zero_outside): a device int32 table [level][image][rows, cols] with
    levels 0 = after the stem's pool (stage 2), 1 = stage 3, 2 = stage 4 (the conv4 map) of a ResNet base.  The table's address is
    fixed (a captured pass bakes it in); ``set(i, H, W)`` writes image i's rows of the HOST copy, ``upload()`` sends it."""
    LEVELS = 3

    def __init__(self, batch, pinned=None):
        """``pinned``: a pinned uint8 host tensor of at least LEVELS * batch * 8 bytes to use as the host copy (entry.DetectionEntry cuts it
        from its arena: a pin_memory() call per captured pass is a hipHostMalloc of several milliseconds)."""
        self.batch = int(batch)
        n = self.LEVELS * self.batch * 2
        if pinned is not None:
            self.host = pinned[:4 * n].view(torch.int32).view(self.LEVELS, self.batch, 2)
            self.host.zero_()
        else:
            self.host = torch.zeros((self.LEVELS, self.batch, 2), dtype=torch.int32).pin_memory()
        self.table = torch.zeros((self.LEVELS, self.batch, 2), dtype=torch.int32, device="cuda")

    @staticmethod
    def levels_of(height, width):
        """[(rows, cols)] per level for an image of this size: resnet.get_conv_rows_cols' chain (resnet.py:78-93), level by level."""
        out = []
        h, w = (height - 1) // 2 + 1, (width - 1) // 2 + 1      # conv1 7x7 / 2 SAME
        h, w = (h - 3) // 2 + 1, (w - 3) // 2 + 1                # max-pool 3x3 / 2 VALID
        out.append((h, w))
        for _ in range(2):                                       # res3a, res4a: 1x1 / 2 VALID
            h, w = (h - 1) // 2 + 1, (w - 1) // 2 + 1
            out.append((h, w))
        return out

    def set(self, i, height, width):
        for lvl, (h, w) in enumerate(self.levels_of(height, width)):
            self.host[lvl, i, 0], self.host[lvl, i, 1] = h, w

    def upload(self):
        self.table.copy_(self.host, non_blocking=True)

    def level(self, lvl):
        return self.table[lvl]


def run_block(u, x, layout=0, planes=False, mask=None, out_planes=False):
    """identity_block / conv_block (resnet.py:114-247) and their TimeDistributed twins (:250-392).  ``planes``: branch2a's and
    branch2b's outputs have ONE reader each, the next convolution of the block: on the f16x3 engine they are handed on as the fp16
    planes that convolution multiplies (ops.PlaneTensor), so its loader splits nothing.  ``out_planes`` (round 6): the block's OUTPUT
    as planes only -- its two readers are the next block's branch2a, which stages them unchanged, and that block's closing 1x1, which
    reads the shortcut back from them (block_takes_planes says whether the next block can); ``x`` may be such a tensor."""
    planes = planes and HEAD_PLANES
    pair = _pair(u["2a"], u["1"]) if "1" in u else None
    if pair is not None:                                    # branch2a and the shortcut conv read x: one launch
        t, shortcut = pair(x, layout=layout)
    else:
        shortcut = u["1"](x, layout=layout) if "1" in u else x
        t = u["2a"](x, layout=layout, planes_out=planes and _reads_planes(u["2b"], tuple(x.shape[:-1]) + (_cout(u["2a"]),), layout))
    if mask is not None:                                    # a canvas pass: branch2b's 3x3 must read zeros beyond each image's true extent
        t = ops.zero_outside(t, mask)
    t = u["2b"](t, layout=layout, planes_out=planes and _reads_planes(u["2c"], t.shape, layout))
    return u["2c"](t, residual=shortcut, layout=layout, planes_out=out_planes)


def block_takes_planes(u, x_shape, layout):
    """Can identity block ``u`` take its input of this shape as an ops.PlaneTensor -- branch2a staging the planes, the closing 1x1 reading
    the shortcut from them?  (Both launches on the f16x3 engine's 256x128 tile.)"""
    if not (HEAD_PLANES and HEAD_BLOCK_PLANES) or "1" in u:
        return False
    inner = tuple(x_shape[:-1]) + (_cout(u["2b"]),)
    return _reads_planes(u["2a"], x_shape, layout) and _reads_planes(u["2c"], inner, layout)


def _cout(unit):
    if unit.pc is None:
        unit.lower()
    return unit.pc.cout


def _reads_planes(unit, x_shape, layout):
    """Will ``unit`` read a tensor of this shape as fp16 planes (ops.conv_accepts_planes)?  (stride-1 layers of a head block)"""
    if unit.dtype != "f32":
        return False
    if unit.pc is None:
        unit.lower()
    return ops.conv_accepts_planes(tuple(x_shape), unit.pc, unit.stride, unit.padding, unit.act, layout, unit.tile)


class ResNetBase:
    """conv1 .. stage 4 (resnet50_base resnet.py:395-448, resnet101_base :551-602)."""
    stride = 16
    out_channels = 1024

    def __init__(self, weights, depth, dtype="f32"):
        self.weights, self.depth, self.dtype = weights, depth, dtype
        r101 = depth == 101
        self.stem = ConvUnit(weights, "conv1", "bn_conv1", "scale_conv1" if r101 else None, BN_EPS_STEM,
                             stride=2, padding="same", act="relu")
        self.blocks = []
        self.block_level = []                                # Extents level of each block's tensors (stage 2 -> 0, 3 -> 1, 4 -> 2)
        for stage, block, is_conv in resnet_block_names(depth):
            stride = 2 if (is_conv and stage > 2) else 1
            self.blocks.append(_block_units(weights, stage, block, is_conv, stride, r101, dtype))
            self.block_level.append(stage - 2)

    def units(self):
        yield self.stem
        for b in self.blocks:
            yield from b.values()

    def invalidate_fused(self, only=None):
        """models._Model.invalidate: drop the fused bf16 stem with conv1's lowering (the cache below is keyed on the weight
        arrays' id(), which a freed-and-reallocated generation of arrays can repeat: ADVICE r3)."""
        if only is None or self.stem.conv in only or self.stem.bn in only or self.stem.scale_name in only:
            self._stem_bf16 = None
            self._stem_h3 = None

    def lower_fused_stem(self):
        """The packed form frcnn_stem_bf16_fwd reads (None on an f32 base); built once, dropped by invalidate_fused."""
        if self.dtype != "bf16" or not FUSED_BF16_STEM:
            return None
        src = tuple(id(a) for name in (self.stem.conv, self.stem.bn, self.stem.scale_name) if name for a in self.weights[name])
        if getattr(self, "_stem_bf16", None) is None or self._stem_src != src:      # (re-packed when conv1 / its BatchNorm were replaced)
            self._stem_bf16, self._stem_src = ops.PackedStemBf16(*self.stem.folded()), src
        return self._stem_bf16

    def lower_fused_stem_h3(self):
        """The packed form frcnn_stem_h3_fwd reads; built once per generation of conv1's arrays, dropped by invalidate_fused."""
        src = tuple(id(a) for name in (self.stem.conv, self.stem.bn, self.stem.scale_name) if name for a in self.weights[name])
        if getattr(self, "_stem_h3", None) is None or self._stem_h3_src != src:
            self._stem_h3, self._stem_h3_src = ops.PackedStemH3(*self.stem.folded()), src
        return self._stem_h3

    def stem_pool(self, x):
        """conv1 + BN (+ Scale) + ReLU + MaxPooling2D((3,3), strides=(2,2)) (resnet.py:408-412).  bf16 path: ONE launch on
        the bf16 matrix cores with the pool and the bf16 store fused (frcnn_stem_bf16_fwd); ``FUSED_BF16_STEM = False``
        keeps the round-1/2 form (f32 conv, f32 pool, cast)."""
        if self.dtype == "bf16" and FUSED_BF16_STEM:
            return ops.stem_bf16(x, self.lower_fused_stem())
        if self.dtype == "f32" and FUSED_F32_STEM and ops.F32_ENGINE == "f16x3" and x.shape[1] >= 7 and x.shape[2] >= 7:
            return ops.stem_h3(x, self.lower_fused_stem_h3())   # the fp32 twin on the f16x3 engine (round 5): one launch, no 38 MB conv map
        x = ops.pool2d(self.stem(x), 3, 2, True)
        return ops.cast_bf16(x) if self.dtype == "bf16" else x

    def __call__(self, x, extents=None):
        """``extents`` (Extents): x is a batch of canvases whose images have those true sizes -- every tensor that a 3x3 convolution
        reads is zeroed beyond each image's extent first, the conv4 map at the end (rpn_conv1 is a 3x3 too)."""
        x = self.stem_pool(x)
        planes = TRUNK_PLANES and self.dtype == "f32" and extents is None      # (a canvas pass masks branch2a's f32 output)
        for b, lvl in zip(self.blocks, self.block_level):
            x = run_block(b, x, planes=planes, mask=None if extents is None else extents.level(lvl))
        if extents is not None:
            x = ops.zero_outside(x, extents.level(2))
        return x


class VggBase:
    """13 conv + 4 pools, no pool5 (vgg16_base vgg.py:91-141)."""
    stride = 16
    out_channels = 512

    def __init__(self, weights):
        self.weights = weights
        self.convs = [(name, ConvUnit(weights, name, padding="same", act="relu")) for name, _, _ in VGG_CONVS]

    def units(self):
        for _, u in self.convs:
            yield u

    POOLED = ("block1_conv2", "block2_conv2", "block3_conv3", "block4_conv3")

    def __call__(self, x):
        """Round 6: a convolution whose output has ONE reader, the next convolution of its block, hands it on as the fp16 planes that one
        multiplies where both launches run on the f16x3 engine's 256x128 tile (ops.PlaneTensor, as inside the ResNet head's blocks): the
        reader's loader splits nothing and, its reduction being 36-144 chunks long, walks the direct-to-LDS ring."""
        for k, (name, u) in enumerate(self.convs):
            nxt = self.convs[k + 1][1] if (k + 1 < len(self.convs) and name not in self.POOLED) else None
            planes = VGG_PLANES and nxt is not None and _reads_planes(nxt, tuple(x.shape[:-1]) + (_cout(u),), 0)
            x = u(x, planes_out=planes)
            if name in self.POOLED:
                # the pooled map's one reader is the next block's first convolution: planes for it where it reads them (round 6)
                after = self.convs[k + 1][1] if k + 1 < len(self.convs) else None
                pooled = (x.shape[0], ops.valid_out(x.shape[1], 2, 2), ops.valid_out(x.shape[2], 2, 2), x.shape[3])
                x = ops.pool2d(x, 2, 2, True, planes_out=VGG_PLANES and after is not None and _reads_planes(after, pooled, 0))
        return x


class RpnHead:
    """rpn_conv1 3x3+ReLU, rpn_out_cls 1x1 sigmoid, rpn_out_bbreg 1x1 linear (resnet.py:464-474)."""

    def __init__(self, weights, dtype="f32"):
        self.conv = ConvUnit(weights, "rpn_conv1", padding="same", act="relu", dtype=dtype)
        self.cls = ConvUnit(weights, "rpn_out_cls", act="sigmoid", dtype=dtype, out_f32=True)      # scores / deltas leave in f32
        self.reg = ConvUnit(weights, "rpn_out_bbreg", dtype=dtype, out_f32=True)

    def units(self):
        return [self.conv, self.cls, self.reg]

    def __call__(self, feat):
        t = self.conv(feat)
        pair = _pair(self.cls, self.reg)
        if pair is not None and t.dtype == torch.float32:   # sigmoid scores and linear deltas from one N = 5A launch
            return pair(t)
        return self.cls(t), self.reg(t)


class _MergedDenseUnit(ConvUnit):
    """The two dense layers as one 1x1 "conv": the kernels are concatenated from the LIVE weight dict whenever the unit is
    (re-)lowered, so load_weights / set_weights on either layer reach the launch like any other layer's."""

    def __init__(self, weights, num_classes):
        super().__init__(weights, "dense")
        self.names = ("dense_class_%d" % num_classes, "dense_reg_%d" % num_classes)

    def folded(self):
        (kc, bc), (kr, br) = (self.weights[n] for n in self.names)
        merged = {"dense": [np.concatenate([kc, kr], axis=1), np.concatenate([bc, br])]}
        return ConvUnit(merged, "dense").folded()


class _MergedDense:
    """dense_class_C (softmax) and dense_reg_C (linear) share their input, so they run as ONE
    GEMM with the two kernels concatenated along the output axis (resnet.py:522-533)."""

    def __init__(self, weights, num_classes):
        self.C = num_classes
        self.unit = _MergedDenseUnit(weights, num_classes)

    def __call__(self, x2d):
        n, cin = x2d.shape
        y = self.unit(x2d.reshape(n, 1, 1, cin)).reshape(n, -1)
        return ops.dense_heads_split(y, self.C)                 # softmax | regressions, one launch, no torch kernels


BATCHED_F32_HEAD_LAYOUT = int(_os.environ.get("FRCNN_BATCHED_HEAD_LAYOUT", "0"))       # dev knob: 1 = position-major crops in the batched fp32 pass too
BATCHED_BF16_HEAD_LAYOUT = int(_os.environ.get("FRCNN_BATCHED_HEAD_LAYOUT_BF16", "1"))  # dev knob: 0 = [roi][7][7][c] crops in the batched bf16 pass


class ResNetHead:
    """RoiResizeConv -> stage 5 (TimeDistributed) -> AveragePooling2D(7) -> dense x2
    (resnet50_classifier resnet.py:489-548, resnet101_classifier :631-686).

    ``hoist`` (default): RoiResizeConv is a per-channel LINEAR resampling whose weights sum to 1, and the two
    layers that consume its output -- res5a_branch2a and res5a_branch1, both 1x1, strides (1,1) (resnet.py:508),
    each followed by an inference-mode BatchNorm (an affine map) -- are per-pixel linear, so they commute with
    it exactly in real arithmetic: conv(resize(crop(F))) == resize(crop(conv(F))).  The head therefore applies
    them ONCE to the conv4 map (2 394 pixels) instead of to every RoI crop (300 x 49 = 14 700 pixels) and
    resamples their outputs; the ReLU after branch2a moves behind the resampling.  That removes 64.6 of the
    537.4 GFLOP per image (12 %); in fp32 the result differs from the reference order by rounding only
    (the parity tests hold it to the same 1e-4 bar)."""
    pool = 7

    def __init__(self, weights, depth, num_classes, dtype="f32", hoist=True, pos_major=True):
        r101 = depth == 101
        self.dtype, self.hoist = dtype, hoist
        # RoI crops kept as [7][7][roi][c]: a 128-row conv tile then covers one or two output positions and the 3x3
        # layers skip the filter taps that only meet zero padding (14 % of their chunks; bit-identical results)
        self.layout = 1 if (pos_major and _os.environ.get("FRCNN_HEAD_POS_MAJOR", "1") != "0") else 0      # (dev knob: 0 = [roi][7][7][c] crops)
        self.blocks = [_block_units(weights, 5, b, b == "a", 1, r101, dtype) for b in "abc"]
        self.dense = _MergedDense(weights, num_classes)

    def units(self):
        for b in self.blocks:
            yield from b.values()
        yield self.dense.unit

    def _first_block_hoisted(self, feat, rois, resize, layout=None):
        a = self.blocks[0]
        if feat.dim() == 4 and feat.shape[0] > 1:               # a batch of maps (forward_batched): `resize` knows which RoI crops which
            fmap = feat
        else:
            fmap = ops.amax_carry(feat.reshape(1, feat.shape[-3], feat.shape[-2], feat.shape[-1]), feat)    # (a view keeps its magnitude record)
        pair = _pair(a["2a"], a["1"])
        if pair is not None:
            u, v = pair(fmap, act1=None)                    # one launch: conv + BN of both on the map
        else:
            u = a["2a"](fmap, act=None)                     # conv + BN on the map; its ReLU follows the resampling
            v = a["1"](fmap)                                # shortcut conv + BN
        # an invalid (empty) RoI crops to zeros in the reference order, which these layers map to their BN shift
        L = self.layout if layout is None else layout
        planes = self.dtype == "f32" and HEAD_PLANES
        n = rois.reshape(-1, 4).shape[0]
        crop_shape = (self.pool, self.pool, n, u.shape[-1]) if L else (n, self.pool, self.pool, u.shape[-1])
        if planes and _reads_planes(a["2b"], crop_shape, L):     # the crops go to res5a_branch2b as the planes it multiplies
            t = resize(u, rois, self.pool, fill=a["2a"].pc.shift, relu=True, layout=L, planes_out=True)
        else:
            t = resize(u, rois, self.pool, fill=a["2a"].pc.shift, relu=True, layout=L)
        s = resize(v, rois, self.pool, fill=a["1"].pc.shift, layout=L)
        out_shape = crop_shape[:-1] + (_cout(a["2c"]),)
        nxt = self.blocks[1] if len(self.blocks) > 1 else None
        return a["2c"](a["2b"](t, layout=L, planes_out=planes and _reads_planes(a["2c"], crop_shape[:-1] + (_cout(a["2b"]),), L)), residual=s, layout=L,
                       planes_out=planes and nxt is not None and block_takes_planes(nxt, out_shape, L))

    def _run_rest(self, x, rest, L):
        """The identity blocks behind the first: each hands its output to the next as planes where that block can take them."""
        for i, b in enumerate(rest):
            nxt = rest[i + 1] if i + 1 < len(rest) else None
            x = run_block(b, x, L, planes=self.dtype == "f32",
                          out_planes=self.dtype == "f32" and nxt is not None and block_takes_planes(nxt, tuple(x.shape[:-1]) + (_cout(b["2c"]),), L))
        return x

    def __call__(self, feat, rois):
        resize = ops.roi_crop_resize_bf16 if self.dtype == "bf16" else ops.roi_crop_resize
        L = self.layout
        if self.hoist:
            x = self._first_block_hoisted(feat, rois, resize)
            rest = self.blocks[1:]
        else:
            x = resize(feat, rois, self.pool, layout=L)     # (n,7,7,1024), or (7,7,n,1024) position-major
            rest = self.blocks
        x = self._run_rest(x, rest, L)
        if self.dtype == "bf16":
            return self.dense(ops.avgpool_bf16(x, 7, L))   # pooled features and the dense layers stay f32
        if L:
            return self.dense(ops.avgpool_pos_major(x))     # (n,2048)
        x = ops.pool2d(x, 7, 7, False)                      # (n,1,1,2048)
        return self.dense(x.reshape(x.shape[0], -1))


    def forward_batched(self, feat, rois, n_per_img):
        """The head over the RoIs of a BATCH of images in one pass: feat (B,R,C,Cf) (bf16 or f32 model), rois (B*n_per_img,4) (RoI r belongs
        to image r // n_per_img) -> (class probabilities (B*n,C), regressions).  Same layers, same per-row arithmetic as
        ``__call__`` per image; the GEMMs are B times taller (one launch per layer for the whole batch)."""
        assert self.hoist, "batched head: hoisted order"
        if self.dtype == "f32":                                 # the same code path as one image: only the resampling knows about the batch
            import functools
            # [roi][7][7][c] crops for the batched pass: a tile's nine taps then re-read the tile's OWN rows (L2) instead of the
            # neighbouring positions' (fabric: 2.4x the algorithmic bytes per launch in the position-major form); no tap can be skipped
            # (14 % more chunks), each launch alone is 0.5 % slower -- and four passes in flight 0.9 % faster (549-551 against 545-546
            # img/s, three alternating runs).  One-image passes tie (507 / 507) and the latency form loses 2 %: they keep position-major.
            L = BATCHED_F32_HEAD_LAYOUT if self.layout else 0
            x = self._first_block_hoisted(feat, rois, functools.partial(ops.roi_crop_resize, n_per_img=n_per_img), layout=L)
            x = self._run_rest(x, self.blocks[1:], L)
            if L:
                return self.dense(ops.avgpool_pos_major(x))
            x = ops.pool2d(x, 7, 7, False)
            return self.dense(x.reshape(x.shape[0], -1))
        L, a = (self.layout and BATCHED_BF16_HEAD_LAYOUT), self.blocks[0]
        u = a["2a"](feat, act=None)                         # conv + BN on every image's map (M = B * rows * cols)
        v = a["1"](feat)
        t = ops.roi_crop_resize_bf16_batch(u, rois, n_per_img, self.pool, fill=a["2a"].pc.shift, relu=True, layout=L)
        s = ops.roi_crop_resize_bf16_batch(v, rois, n_per_img, self.pool, fill=a["1"].pc.shift, layout=L)
        x = a["2c"](a["2b"](t, layout=L), residual=s, layout=L)
        for b in self.blocks[1:]:
            x = run_block(b, x, L)
        return self.dense(ops.avgpool_bf16(x, 7, L))


class VggHead:
    """RoiResizeConv -> Flatten (h,w,c order) -> fc1, fc2 (ReLU) -> dense x2 (vgg.py:226-255)."""
    pool = 7

    def __init__(self, weights, num_classes):
        self.fc1 = ConvUnit(weights, "fc1", act="relu")
        self.fc2 = ConvUnit(weights, "fc2", act="relu")
        self.dense = _MergedDense(weights, num_classes)

    def units(self):
        return [self.fc1, self.fc2, self.dense.unit]

    def __call__(self, feat, rois):
        x = ops.roi_crop_resize(feat, rois, self.pool)      # (n,7,7,512)
        n = x.shape[0]
        x = self.fc1(x.reshape(n, 1, 1, -1))
        x = self.fc2(x)
        return self.dense(x.reshape(n, -1))


def to_device_image(x):
    """(1,H,W,3) or (H,W,3) float array/tensor -> (1,H,W,3) f32 device tensor."""
    if isinstance(x, torch.Tensor):
        t = x.to(device="cuda", dtype=torch.float32)
    else:
        t = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).cuda()
    if t.dim() == 3:
        t = t.unsqueeze(0)
    return t.contiguous()
