"""Training step engine: what Keras' ``compile`` + ``train_on_batch`` do for the reference
(train_util.py:29-56, 93-120), on the HIP library.

A training model is the inference lowering (nets.py) plus, for every TRAINABLE convolution
(freeze_blocks=[1,2,3] -> ResNet stage 4, the RPN heads, stage 5 and the dense layers,
resnet.py:395, 432):
    forward   frcnn_conv2d_fwd on weights re-packed from the fp32 master copy after each update
    wgrad     frcnn_conv2d_wgrad straight into a slice of ONE flat gradient buffer
    dgrad     frcnn_conv2d_fwd_masked on the transposed filter (ReLU backward + shortcut add fused)
BatchNorm / Scale are frozen (bn_training=False everywhere in the reference) and stay folded.
One flat buffer of gradients means data-parallel training needs exactly ONE all-reduce per step
(RCCL through torch.distributed); the optimiser is one launch over the flat parameter buffer.
"""
import ctypes
import weakref

import numpy as np
import torch

from . import _lib, nets, ops
from .ops import _p, _stream


# ----------------------------------------------------------------------------- optimisers (args_util.py:48-59)
# ``iterations`` mirrors Keras' ``optimizer.iterations``: a variable created in the optimiser's __init__ that counts
# every update the OBJECT has made.  The reference builds one optimiser per run (args_util.py:56-59) and re-compiles
# with it for every phase (train_util.py:29-33): ``get_updates`` then re-creates the moment slots, but the counter
# keeps running, so Adam's bias correction in phase 2 starts from t = iterations + 1, not from 1 [Keras 2.0.8
# optimizers.py, Adam.get_updates].  SGD carries the same counter (it only feeds ``decay``, which is 0 here).
class SGD:
    def __init__(self, lr=1e-3, momentum=0.9):
        self.lr, self.momentum = lr, momentum
        self.iterations = 0


class Adam:
    def __init__(self, lr=1e-3, beta_1=0.9, beta_2=0.999, epsilon=1e-8):
        self.lr, self.beta_1, self.beta_2, self.epsilon = lr, beta_1, beta_2, epsilon
        self.iterations = 0


def optimizer_from_str(optimizer_str):
    """args_util.optimizer_from_str (args_util.py:48-59)."""
    return SGD(lr=1e-3, momentum=0.9) if optimizer_str == "sgd" else Adam(lr=1e-3)


# ----------------------------------------------------------------------------- flat parameter storage
class ParamSet:
    """fp32 master weights, gradients and optimiser slots of all trainable tensors, each ONE flat
    device buffer.  ``views[name]`` = list of (weight_view, grad_view) in Keras get_weights() order."""

    def __init__(self, weights, names):
        self.names = list(names)
        sizes = [[int(np.prod(a.shape)) for a in weights[n]] for n in self.names]
        self.total = sum(sum(s) for s in sizes)
        self.w = torch.empty(self.total, dtype=torch.float32, device="cuda")
        self.g = torch.zeros(self.total, dtype=torch.float32, device="cuda")
        self.views = {}
        off = 0
        for n in self.names:
            vs = []
            for a in weights[n]:
                k = int(np.prod(a.shape))
                wv = self.w[off:off + k].view(*a.shape)
                wv.copy_(torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)))
                vs.append((wv, self.g[off:off + k].view(*a.shape)))
                off += k
            self.views[n] = vs
        self.slots = None

    def reset_optimizer(self):
        """Keras re-creates the optimiser SLOTS (momentum / Adam moments) on every compile() (train_util.py:31-33);
        the step counter lives on the optimiser object and keeps running (see SGD / Adam above)."""
        self.slots = [torch.zeros_like(self.w), torch.zeros_like(self.w)]

    def adopt_slots(self, other):
        """Take over another ParamSet's optimiser slots (same layout): load_weights after compile keeps them."""
        if other.slots is not None and other.total == self.total and other.names == self.names:
            self.slots = [s.clone() for s in other.slots]

    def step(self, opt, l2, grad_scale=1.0):
        opt.iterations = getattr(opt, "iterations", 0) + 1
        if isinstance(opt, Adam):
            _lib.call("frcnn_adam", _p(self.w), _p(self.g), _p(self.slots[0]), _p(self.slots[1]), self.total, float(opt.lr),
                      float(opt.beta_1), float(opt.beta_2), float(opt.epsilon), int(opt.iterations), float(l2), float(grad_scale), _stream())
        else:
            _lib.call("frcnn_sgd_momentum", _p(self.w), _p(self.g), _p(self.slots[0]), self.total, float(opt.lr),
                      float(opt.momentum), float(l2), float(grad_scale), _stream())

    def sumsq(self, out=None):
        if out is None:
            out = torch.empty(1, dtype=torch.float32, device="cuda")
        ws = ops._ws(_lib.load().frcnn_sumsq_workspace_bytes())
        _lib.call("frcnn_sumsq", _p(self.w), self.total, _p(out), _p(ws), ws.numel(), _stream())
        return out

    def export(self, weights):
        for n in self.names:
            weights[n] = [wv.detach().cpu().numpy().copy() for wv, _ in self.views[n]]


# ----------------------------------------------------------------------------- one conv in a training graph
class TConv:
    def __init__(self, unit, params=None, needs_dgrad=False):
        self.u, self.params, self.needs_dgrad = unit, params, needs_dgrad
        self.trainable = params is not None and unit.conv in params.views
        # mixed precision (BASELINE configs[4]): a unit built with dtype "bf16" keeps bf16 activations and gradients and
        # bf16 packed filters; its master weights, weight gradients, BatchNorm fold and optimiser state stay f32
        self.bf16 = getattr(unit, "dtype", "f32") == "bf16"
        self.x = self.y = None
        if not self.trainable:
            if unit.pc is None:
                unit.lower()
            self.pc = unit.pc
            if needs_dgrad:
                k4 = self._kernel4d(np.asarray(unit.weights[unit.conv][0], dtype=np.float32))
                self.pd = (ops.PackedDgradBf16 if self.bf16 else ops.PackedDgrad)(k4, self.pc.scale)
            return
        # frozen BatchNorm / Scale -> constant epilogue scale and shift offset (nets.ConvUnit.lower)
        w = unit.weights
        cout = self._kernel4d(np.asarray(w[unit.conv][0])).shape[3]
        scale = np.ones(cout)
        const = np.zeros(cout)
        if unit.bn is not None:
            g, b, m, v = (np.asarray(a, dtype=np.float64) for a in w[unit.bn])
            scale = g / np.sqrt(v + unit.eps)
            const = b - m * scale
        if unit.scale_name is not None:
            g2, b2 = (np.asarray(a, dtype=np.float64) for a in w[unit.scale_name])
            scale, const = scale * g2, const * g2 + b2
        self.has_bn = unit.bn is not None or unit.scale_name is not None
        self.scale = torch.from_numpy(scale.astype(np.float32)).cuda() if self.has_bn else None
        self.const = torch.from_numpy(const.astype(np.float32)).cuda() if self.has_bn else None
        vs = params.views[unit.conv]
        self.wv, self.gw = vs[0]
        self.bv, self.gb = vs[1] if len(vs) > 1 else (None, None)
        k4 = self.wv if self.wv.dim() == 4 else self.wv.view(1, 1, *self.wv.shape)
        self.k4, self.gk4 = k4, (self.gw if self.gw.dim() == 4 else self.gw.view(1, 1, *self.gw.shape))
        self.kh, self.kw, self.cin, self.cout = (int(v) for v in k4.shape)
        if self.bf16:
            assert self.cin % 64 == 0 and (not needs_dgrad or self.cout % 64 == 0), "bf16 training layers need 64-multiple channel counts"
            kp, kpd, wdt, cls_pc, cls_pd = self.kh * self.kw * self.cin, self.kh * self.kw * self.cout, torch.bfloat16, ops.PackedConvBf16, ops.PackedDgradBf16
        else:
            kp = _lib.load().frcnn_conv_packed_k(self.kh, self.kw, self.cin)
            kpd = _lib.load().frcnn_conv_packed_k(self.kh, self.kw, self.cout)
            wdt, cls_pc, cls_pd = torch.float32, ops.PackedConv, ops.PackedDgrad
        self.pc = cls_pc.__new__(cls_pc)
        self.pc.kh, self.pc.kw, self.pc.cin, self.pc.cout = self.kh, self.kw, self.cin, self.cout
        self.pc.w = torch.empty((self.cout, kp), dtype=wdt, device="cuda")
        self.pc.scale = self.scale
        self.pc.shift = torch.empty(self.cout, dtype=torch.float32, device="cuda")
        if needs_dgrad:
            self.pd = cls_pd.__new__(cls_pd)
            self.pd.kh, self.pd.kw, self.pd.cin, self.pd.cout = self.kh, self.kw, self.cout, self.cin
            self.pd.w = torch.empty((self.cin, kpd), dtype=wdt, device="cuda")
            self.pd.scale = self.pd.shift = None
        self.refresh()

    @staticmethod
    def _kernel4d(k):
        return k.reshape(1, 1, *k.shape) if k.ndim == 2 else k

    def refresh(self):
        """Re-pack from the master weights (after an optimiser step)."""
        if self.trainable:
            refresh_packed(make_refresh_jobs([self]))

    def pack_job(self):
        ptr = lambda t: None if t is None else t.data_ptr()
        return _lib.PackJob(w_hwio=ptr(self.k4), packed=ptr(self.pc.w), packed_dgrad=ptr(self.pd.w) if self.needs_dgrad else None,
                            bias=ptr(self.bv), scale=ptr(self.scale), shift_const=ptr(self.const), shift=ptr(self.pc.shift),
                            kh=self.kh, kw=self.kw, cin=self.cin, cout=self.cout)

    def forward(self, x, residual=None):
        self.x = x
        if self.bf16:
            self.y = ops.conv2d_bf16(x, self.pc, self.u.stride, self.u.padding, self.u.act, residual)
        else:
            self.y = ops.conv2d(x, self.pc, self.u.stride, self.u.padding, self.u.act, residual)
        return self.y

    def wgrad(self, g):
        """g: gradient w.r.t. this layer's post-BN pre-activation output.  Nothing reads a weight gradient before the
        optimiser, so the launch is only QUEUED here: flush_weight_grads issues every layer's weight gradient of the
        step together (frcnn_conv2d_wgrad_batch) and the bias gradients in one more launch."""
        if self.trainable:
            gc = g if g.is_contiguous() else g.contiguous()
            _PENDING_WGRAD.append((self.x, gc, self.kh, self.kw, self.u.stride, self.u.padding, self.scale, self.gk4))
            if self.gb is not None:
                _PENDING_BIAS.append((gc, self.scale, self.gb))
            if len(_PENDING_WGRAD) >= WGRAD_FLUSH_JOBS:
                _launch_pending_wgrads()

    def dgrad(self, g, residual=None, mask=None):
        if self.bf16:
            return ops.conv2d_dgrad_bf16(g, self.pd, self.u.padding, residual=residual, mask=mask)
        return ops.conv2d_dgrad(g, self.pd, self.u.padding, residual=residual, mask=mask)


_PENDING_WGRAD = []         # (x, g, kh, kw, stride, padding, scale, dw) of this step's trainable convs, in backward order
_PENDING_BIAS = []          # (g, scale, dbias); the tensors are kept alive until the flush
# Weight gradients are issued in batches of this many layers (frcnn_conv2d_wgrad_batch) on a second HIP stream, beside
# the input-gradient chain that produces the next batch's operands; the rest goes out at the end of the backward pass.
WGRAD_FLUSH_JOBS = int(__import__("os").environ.get("FRCNN_WGRAD_FLUSH", "8"))
# The f32 weight gradients of a step run on the split-bf16 engine (ops.WGRAD_ENGINE; layers with cin, cout >= 128): both operands
# split exactly into three bf16 pieces inside the kernel, error against fp64 at the native kernel's level, bitwise reproducible.
# Stage 4's 3x3 layers 63.9 -> 38.7 us, the detector head's 243 -> 159 us each.  "native" restores v_mfma_f32_32x32x2_f32.
WGRAD_ENGINE = __import__("os").environ.get("FRCNN_TRAIN_WGRAD", "bf16x6")
# The forward convolutions of an f32 step (the frozen stages of the next image and the trainable layers) under ops.f32_engine: the
# split engine's launch policy applies as in inference (stages 2-3 on 64x64 tiles, stage 4's wide 1x1, rpn_conv1 / 3x3 split-K);
# a trainable layer's three filter planes are re-derived after every update.  RPN step 2.32 -> 2.26 ms, detector step 4.39 -> 4.22.
# Round 5: FRCNN_TRAIN_F32_ENGINE=f16x3 puts the forward AND the input-gradient launches on the f16x3 engine (three matrix instructions
# per block of products instead of six): a gradient tensor carries the magnitude record its producing launch published, like an
# activation; the records of a step come from two arenas (one per stream); header + two fp16 planes of every trainable filter are
# re-derived after the update (frcnn_refresh_h3_planes).  Measured 1.97 / 3.62 ms against 2.04 / 3.71 (a step's launches are the
# 2 394- / 3 136-row split-K forms: latency- and L2-bound, not MFMA-bound), and NOT the default: under ONE power-of-two scale per
# tensor an element below 2^-22 of the tensor's largest vanishes -- harmless in activations, but a gradient tensor's rows (one RoI's
# against another's) can lie further apart than that, and the exact split keeps every element's 24 bits.
F32_ENGINE = __import__("os").environ.get("FRCNN_TRAIN_F32_ENGINE", "bf16x6")
# A step on an input shape seen more than STEP_GRAPH_AFTER times replays from captured hipGraphs (_StepDriver._capture_step): the
# ~110 launches of a mixed-precision RPN step took ~1.15 ms of Python to enqueue against 1.27 ms of kernels.  At most
# STEP_GRAPH_SHAPES shapes stay captured (least recently used first out); FRCNN_TRAIN_GRAPH=0 keeps every step eager.
STEP_GRAPHS = __import__("os").environ.get("FRCNN_TRAIN_GRAPH", "1") != "0"
STEP_GRAPH_AFTER = int(__import__("os").environ.get("FRCNN_TRAIN_GRAPH_AFTER", "2"))
STEP_GRAPH_SHAPES = int(__import__("os").environ.get("FRCNN_TRAIN_GRAPH_SHAPES", "8"))
_WGRAD_STREAM = None


def _wgrad_stream():
    global _WGRAD_STREAM
    if _WGRAD_STREAM is None:
        _WGRAD_STREAM = torch.cuda.Stream()
    return _WGRAD_STREAM


def _launch_pending_wgrads():
    if not _PENDING_WGRAD:
        return
    prev, ops.WGRAD_ENGINE = ops.WGRAD_ENGINE, WGRAD_ENGINE
    try:
        rec = _RECORDER
        if rec is not None:
            # a step being captured (_StepDriver._capture_step): the batch becomes a graph of its own, replayed on the weight-gradient
            # stream.  The pieces are captured one after the other on ONE stream, so the allocator sees a single timeline: whatever
            # this batch reads or uses as workspace must stay allocated until the whole capture ends, or a later piece of the main
            # lane -- which runs BESIDE this one at replay -- would be handed the same memory
            rec.cut("wgrad")
            rec.keep.append(ops.conv2d_wgrad_batch(_PENDING_WGRAD))
            rec.keep.append(list(_PENDING_WGRAD))
            rec.cut("main")
            _PENDING_WGRAD.clear()
            return
        side, cur = _wgrad_stream(), torch.cuda.current_stream()
        side.wait_stream(cur)                               # the operands were produced on the main stream
        with torch.cuda.stream(side):
            ops.conv2d_wgrad_batch(_PENDING_WGRAD)
    finally:
        ops.WGRAD_ENGINE = prev
    for job in _PENDING_WGRAD:
        job[0].record_stream(side)
        job[1].record_stream(side)
    _PENDING_WGRAD.clear()


def flush_weight_grads():
    """End of the backward pass: whatever weight gradients are still queued, then (after rejoining the side stream) all
    bias gradients (dbias[co] = scale[co] * sum_m g[m][co]) in one launch.  Launched one by one (round 1) each of these
    small GEMMs paid its own ramp, tail and reduction launch: 44 + 2 launches per RPN step, now a handful."""
    _launch_pending_wgrads()
    if _WGRAD_STREAM is not None and _RECORDER is None:     # (a replayed step joins the lanes after its last piece)
        torch.cuda.current_stream().wait_stream(_WGRAD_STREAM)
    if not _PENDING_BIAS:
        return
    jobs = (_lib.ColsumJob * len(_PENDING_BIAS))()
    for j, (g, scale, out) in zip(jobs, _PENDING_BIAS):
        j.g, j.scale, j.out = g.data_ptr(), (None if scale is None else scale.data_ptr()), out.data_ptr()
        j.cout = g.shape[-1]
        j.m = g.numel() // g.shape[-1]
        j.g_is_bf16 = 1 if g.dtype == torch.bfloat16 else 0
    _lib.call("frcnn_colsum_batch", jobs, len(_PENDING_BIAS), _stream())
    _PENDING_BIAS.clear()


_RECORDER = None
_DBG = __import__("os").environ.get("FRCNN_DBG", "")
_TICK = __import__("os").environ.get("FRCNN_TICK", "0") != "0"


class _Pieces:
    """A run of LINEAR hipGraphs captured one after the other on the current (side) stream into one memory pool; ``cut(lane)`` ends
    the piece being captured and starts the next.  Why pieces and not one graph with a forked branch: the runtime spreads a graph's
    parallel branches over its hardware queues, the prefix stream's among them -- the NEXT image's frozen stages then queue behind
    this step's backward pass instead of running beside it (rocprofv3 timeline, round 6) -- while a linear graph stays on the queue
    of the stream it is replayed on."""

    def __init__(self):
        self.pool, self.items, self.cur, self.keep = None, [], None, []

    def begin(self, lane):
        g = torch.cuda.CUDAGraph()
        if self.pool is None:
            self.pool = torch.cuda.graph_pool_handle()
        g.capture_begin(pool=self.pool, capture_error_mode="thread_local")   # (another thread -- an RCCL watchdog -- may call into HIP meanwhile)
        self.cur = (lane, g)

    def end(self):
        lane, g = self.cur
        self.cur = None
        g.capture_end()
        self.items.append((lane, g))

    def cut(self, lane):
        self.end()
        self.begin(lane)

    def abort(self):
        if self.cur is not None:
            try:
                self.cur[1].capture_end()
            except Exception:
                pass
            self.cur = None


def make_refresh_jobs(tconvs):
    """ctypes job tables (f32 layers, bf16 layers) of frcnn_refresh_packed[_bf16] for the trainable convs
    (pointers are fixed for a trainer's life)."""
    f32 = [c.pack_job() for c in tconvs if c.trainable and not c.bf16]
    b16 = [c.pack_job() for c in tconvs if c.trainable and c.bf16]
    return (_lib.PackJob * len(f32))(*f32), (_lib.PackJob * len(b16))(*b16)


def refresh_packed(jobs):
    f32, b16 = jobs
    if len(f32):
        _lib.call("frcnn_refresh_packed", f32, len(f32), _stream())
    if len(b16):
        _lib.call("frcnn_refresh_packed_bf16", b16, len(b16), _stream())


class TBlock:
    """identity_block / conv_block (and TimeDistributed twins) with backward."""

    def __init__(self, units, params, needs_input_grad, input_is_relu=True):
        self.c2a = TConv(units["2a"], params, needs_dgrad=needs_input_grad)
        self.c2b = TConv(units["2b"], params, needs_dgrad=True)
        self.c2c = TConv(units["2c"], params, needs_dgrad=True)
        self.c1 = TConv(units["1"], params, needs_dgrad=needs_input_grad) if "1" in units else None
        self.needs_input_grad, self.input_is_relu = needs_input_grad, input_is_relu

    def convs(self):
        return [c for c in (self.c2a, self.c2b, self.c2c, self.c1) if c is not None]

    def forward(self, x):
        self.x = x
        s = self.c1.forward(x) if self.c1 is not None else x
        t = self.c2a.forward(x)
        t = self.c2b.forward(t)
        return self.c2c.forward(t, residual=s)

    def backward(self, g):
        """g: gradient w.r.t. the block's pre-ReLU output (already masked by output > 0).
        Returns the gradient w.r.t. the block input's pre-ReLU value (or None)."""
        self.c2c.wgrad(g)
        g2 = self.c2c.dgrad(g, mask=self.c2b.y)
        self.c2b.wgrad(g2)
        g1 = self.c2b.dgrad(g2, mask=self.c2a.y)
        self.c2a.wgrad(g1)
        if self.c1 is not None:
            self.c1.wgrad(g)
        if not self.needs_input_grad:
            return None
        mask = self.x if self.input_is_relu else None
        if self.c1 is None:
            return self.c2a.dgrad(g1, residual=g, mask=mask)
        tmp = self.c1.dgrad(g)
        return self.c2a.dgrad(g1, residual=tmp, mask=mask)


DP_SYNC = True          # False: leave the collective out of the step (bench.py / bench_train.py measure what it costs that way)


def _sync_grads_begin(params):
    """Data-parallel exchange: ONE all-reduce (sum) of the flat gradient buffer over RCCL, STARTED here (behind the last
    weight-gradient batch on the current stream) and not waited for.  Returns (handle or None, 1/world scale)."""
    from . import dp
    if not DP_SYNC:
        return None, 1.0
    return dp.allreduce_sum_begin(params.g)


def _l2_of(weights, names):
    """sum(w^2) of frozen regularised layers: a constant of the run (kernel and bias, resnet.py:26-27)."""
    return float(sum(float((np.asarray(a, dtype=np.float64) ** 2).sum()) for n in names for a in weights[n]))


def _regularised_conv_names(weights):
    return [n for n in weights if n.startswith(("conv1", "res", "rpn_", "dense_", "fc", "block")) and not n.startswith("bn")]


# ----------------------------------------------------------------------------- trainable base networks
class ResNetBaseTrain:
    """conv1..stage 4 with the stages outside freeze_blocks trainable (resnet.py:395-448)."""

    def __init__(self, base_model, params):
        from .weights import resnet_block_names
        net = base_model.net
        freeze = set(base_model.freeze_blocks)
        self.bf16 = getattr(net, "dtype", "f32") == "bf16"
        self.net, self.stem, self.frozen_blocks, self.blocks = net, net.stem, [], []
        first = True
        for (stage, block, _), units in zip(resnet_block_names(net.depth), net.blocks):
            if stage in freeze:
                assert first, "a frozen stage after a trainable one is not supported"
                self.frozen_blocks.append(units)
            else:
                self.blocks.append(TBlock(units, params, needs_input_grad=not first))
                first = False

    @staticmethod
    def trainable_names(base_model):
        from .weights import resnet_block_names
        net, freeze = base_model.net, set(base_model.freeze_blocks)
        names = []
        for (stage, block, _), units in zip(resnet_block_names(net.depth), net.blocks):
            if stage not in freeze:
                names += [u.conv for u in units.values()]
        return names

    def convs(self):
        return [c for b in self.blocks for c in b.convs()]

    def forward_frozen(self, x):
        """Stem and the frozen stages: no trainable weight is read, so a step driver may run this for the NEXT image
        while the previous step is still in its backward pass (_StepDriver._run_step)."""
        t = self.net.stem_pool(x)                               # (bf16: conv1 + BN + ReLU + pool + bf16 store in one launch)
        for units in self.frozen_blocks:
            t = nets.run_block(units, t)
        return t

    def forward_rest(self, t):
        for b in self.blocks:
            t = b.forward(t)
        return t

    def forward(self, x):
        return self.forward_rest(self.forward_frozen(x))

    def backward(self, g):
        """g: gradient w.r.t. the pre-ReLU value of the feature map (already masked)."""
        for b in reversed(self.blocks):
            g = b.backward(g)


class VggBaseTrain:
    """vgg16_base (vgg.py:91-141): 3x3 conv + ReLU chain with 2x2 pools; blocks outside freeze_blocks train."""
    POOL_AFTER = ("block1_conv2", "block2_conv2", "block3_conv3", "block4_conv3")

    def __init__(self, base_model, params):
        net = base_model.net
        train = set(self.trainable_names(base_model))
        self.layers = []                       # (name, unit-or-TConv, trainable)
        seen_trainable = False
        for name, u in net.convs:
            if name in train:
                self.layers.append((name, TConv(u, params, needs_dgrad=seen_trainable), True))
                seen_trainable = True
            else:
                assert not seen_trainable, "a frozen block after a trainable one is not supported"
                self.layers.append((name, u, False))

    @staticmethod
    def trainable_names(base_model):
        freeze = set(base_model.freeze_blocks)
        return [name for name, _ in base_model.net.convs if int(name[5]) not in freeze]

    def convs(self):
        return [l for _, l, t in self.layers if t]

    def _run(self, x, layers):
        for name, layer, trainable in layers:
            x = layer.forward(x) if trainable else layer(x)
            if name in self.POOL_AFTER:
                y = ops.pool2d(x, 2, 2, True)
                if trainable:
                    self.pools[name] = (x, y)
                x = y
        return x

    def forward_frozen(self, x):
        """The frozen leading blocks (see ResNetBaseTrain.forward_frozen)."""
        return self._run(x, [l for l in self.layers if not l[2]])

    def forward_rest(self, x):
        self.pools = {}
        return self._run(x, [l for l in self.layers if l[2]])

    def forward(self, x):
        return self.forward_rest(self.forward_frozen(x))

    def backward(self, g):
        for name, layer, trainable in reversed(self.layers):
            if not trainable:
                return
            if name in self.POOL_AFTER:                 # g arrives w.r.t. the pool output: route it, then ReLU-mask
                x, y = self.pools[name]
                gx = torch.empty_like(x)
                _lib.call("frcnn_maxpool_bwd", _p(x), _p(y), _p(g.contiguous()), x.shape[0], x.shape[1], x.shape[2], x.shape[3], 2, _p(gx), _stream())
                _lib.call("frcnn_relu_bwd_inplace", _p(gx), _p(x), gx.numel(), _stream())
                g = gx
            layer.wgrad(g)
            if not layer.needs_dgrad:
                return
            src = layer.x                                   # this conv's input = previous layer's ReLU output (or a pool output)
            prev_pooled = any(src is py for (_, py) in self.pools.values())
            g = layer.dgrad(g, mask=None if prev_pooled else src)


def _make_base_train(base_model, params):
    return (ResNetBaseTrain if isinstance(base_model.net, nets.ResNetBase) else VggBaseTrain)(base_model, params)


def _base_trainable_names(base_model):
    return (ResNetBaseTrain if isinstance(base_model.net, nets.ResNetBase) else VggBaseTrain).trainable_names(base_model)


def _reg_sumsq_frozen(weights, model_layer_names, train_names, l2):
    """sum(w^2) over the model's regularised layers that do NOT train (a constant of the run)."""
    if not l2:
        return 0.0
    return _l2_of(weights, [n for n in model_layer_names if n not in train_names])


def _base_layer_names(base_model):
    return [u.conv for u in base_model.net.units()]


# ----------------------------------------------------------------------------- one step: host side
class _StepDriver:
    """What is common to RpnTrainer and DetTrainer.train_on_batch: host arrays -> device inputs, the device part of the
    step (subclass ``_device_step``: forward, losses, backward, weight gradients, then ``_update``), the three numbers
    Keras returns.

    (Replaying the device part from a hipGraph was built and measured again in round 2 -- third step on an input shape
    captured, weight-gradient side stream forked and joined inside the capture, bit-identical to the eager steps -- and
    interleaved A/B rounds in one process show nothing: fp32 3.65-3.86 vs 3.70-3.88 ms, mixed 2.19-2.31 vs 2.20-2.29 ms per
    RPN step.  The ~110 launches of a step are short DEPENDENT kernels; their boundaries cost the same from a graph.
    Removed again: DESIGN 11.)"""

    def _init_driver(self):
        self._pin_sets = [_PinnedSet(), _PinnedSet()]   # two grow-only pinned staging areas, used alternately
        self._pin_next = 0
        self._conv_ws = ops.ConvWorkspace()
        self._conv_ws_prefix = ops.ConvWorkspace()   # split-K tickets of the launches on the prefix stream
        self._loss_ring = [None] * LOSS_RING          # steps whose three scalars are still on their way to the host
        self._loss_pos = 0
        self._pending_update = None                   # (all-reduce handle, 1/world, optimiser) of a step whose update is not enqueued yet
        self._cur = None
        self._graphs = {"graphs": __import__("collections").OrderedDict(), "seen": {}, "epoch": None, "token": None}
        _LIVE_DRIVERS.add(self)
        self._lower_frozen()

    def _frozen_units(self):
        """ConvUnits the prefix stream runs (stem + frozen stages of the base): they read no trainable weight."""
        base = getattr(self, "base", None)
        if base is None:
            return []
        if hasattr(base, "frozen_blocks"):
            return [base.stem] + [u for units in base.frozen_blocks for u in units.values()]
        return [layer for _, layer, trainable in base.layers if not trainable]

    def _lower_frozen(self):
        """Pack the frozen layers' filters NOW, on the stream the trainer is built on, and mark the point with an event
        the prefix stream waits for: a lazy lowering at first use would run on the prefix stream, unordered against
        main-stream readers of the same packed filters (predict) and allocated in that stream's pool."""
        for u in self._frozen_units():
            if u.pc is None:
                u.lower()
        # ... and the derived forms the frozen stages launch: a frozen conv_block runs branch2a + shortcut as ONE
        # launch on a concatenated filter (nets.DualUnit), and a bf16 base runs conv1 as the fused stem -- both were still
        # built lazily at first use, i.e. on the prefix stream
        base = getattr(self, "base", None)
        for units in getattr(base, "frozen_blocks", []) if base is not None else []:
            if "1" in units:
                pair = nets._pair(units["2a"], units["1"])
                if pair is not None:
                    pair.lower()
        net = getattr(base, "net", None)
        if net is not None and hasattr(net, "lower_fused_stem"):
            net.lower_fused_stem()
        self._frozen_ready = torch.cuda.Event()
        self._frozen_ready.record()

    def _publish_losses(self):
        """Called by ``_device_step`` as soon as the two loss kernels are enqueued (and by a skipped step at once): the L2
        sum over the trainable weights -- they do not change before the optimiser runs at the END of the step -- then the
        three scalars leave for pinned host memory, both on a stream of their own (round 6: the 47-89 MB sum used to sit on the
        step's critical path between the loss kernels and the backward pass).  train_on_batch's return value is therefore
        available after the FORWARD pass; the backward pass and the update run on while the caller reads the losses and
        stages the next image (Keras semantics kept: every way of reading weights back is stream-ordered behind the step; the
        optimiser waits for the sum's event before it rewrites the weights, ``_apply_update``)."""
        self._send_losses()

    def _send_losses(self):
        out3, slot = self._cur
        main, side = torch.cuda.current_stream(), _loss_stream()
        side.wait_stream(main)
        with torch.cuda.stream(side):
            if self.l2:
                self.params.sumsq(out=out3[2:3])
            slot[0].copy_(out3, non_blocking=True)
            slot[1].record()
        out3.record_stream(side)
        self._weights_read = slot[1]                     # the optimiser must not rewrite the weights under the sum

    def _device_step(self, dev, out, pre=None):
        """The device part of one step on the current stream: forward + losses, the three scalars on their way to the host,
        backward, update.  (A captured step replays ``_fwd_part`` and ``_bwd_part`` from two hipGraphs: _StepGraph.)"""
        self._fwd_part(dev, out, pre)
        self._publish_losses()
        self._bwd_part()
        self._update()

    def _update(self, sq_out=None):
        """After the backward pass: the ONE exchange of the flat gradient buffer, optimiser, re-pack.

        world == 1: optimiser and re-pack are enqueued at once.  world > 1: the all-reduce is only STARTED (on the
        collective's own stream, behind this step's last gradient kernel) and the optimiser is left pending:
        ``_finish_update`` enqueues it behind the collective the next time anything needs the new weights -- the next
        step, after that step has staged its image and put upload + frozen stages on the prefix stream (so host staging
        and those launches run BESIDE the collective instead of behind it), or any read-out of the weights
        (sync_weights / compile).  Same kernels in the same order per tensor: bit-identical to the serial form."""
        flush_weight_grads()
        self._exchange_and_apply()

    def _exchange_and_apply(self):
        handle, scale = _sync_grads_begin(self.params)
        if handle is None:
            self._apply_update(scale)
        else:
            self._pending_update = (handle, scale, self.optimizer)

    def _finish_update(self):
        """Wait (stream-wise on RCCL; gloo also blocks the host) for the pending all-reduce, then optimiser + re-pack."""
        if self._pending_update is not None:
            (handle, scale, opt), self._pending_update = self._pending_update, None
            handle.wait()
            with ops.conv_workspace(self._conv_ws):
                self._apply_update(scale, opt)

    def _apply_update(self, scale, opt=None):
        p = self.params
        ev = getattr(self, "_weights_read", None)
        if ev is not None:
            torch.cuda.current_stream().wait_event(ev)   # (the L2 sum of this step's report, on the loss stream)
            self._weights_read = None
        p.step(opt or self.optimizer, self.l2, scale)
        if self._refresh_jobs is None:
            self._refresh_jobs = make_refresh_jobs(self._tconvs())
        refresh_packed(self._refresh_jobs)
        # the packed f32 filters were rewritten IN PLACE: bf16 planes derived from them for the split-bf16 engine
        # (ops.PackedConv.x6_planes, cached on tensor identity) are stale now
        self._refresh_x6()
        self._refresh_h3()

    def _refresh_h3(self):
        """The f16x3 engine's twin of _refresh_x6: header (max|w|) + two fp16 planes of every filter that has them, three launches."""
        live = [pk for c in self._tconvs() for pk in (c.pc, getattr(c, "pd", None)) if getattr(pk, "_h3", None) is not None and pk._h3_src is pk.w]
        key = tuple(id(pk._h3) for pk in live)
        if getattr(self, "_h3_key", None) != key:
            jobs = (_lib.X6Job * max(1, len(live)))()
            for j, pk in zip(jobs, live):
                j.w_packed, j.planes_bf16, j.rows, j.kpad = pk.w.data_ptr(), pk._h3.data_ptr(), pk.w.shape[0], pk.w.shape[1]
            self._h3_key, self._h3_jobs = key, jobs
        if live:
            _lib.call("frcnn_refresh_h3_planes", self._h3_jobs, len(live), _stream())

    def _amax_arenas(self):
        """(prefix stream's, main stream's) magnitude-record arenas of a step under the f16x3 engine.  Records never cross the two
        streams (the frozen prefix's output is measured again by its first reader): the next step's prefix clears ITS arena while this
        step's main part may still be running."""
        a = getattr(self, "_amax", None)
        if a is None:
            a = self._amax = (ops.AmaxArena(256), ops.AmaxArena(256))
        return a

    def _refresh_x6(self):
        """The packed f32 filters were rewritten IN PLACE: the three bf16 planes the split-bf16 engine multiplies with
        (ops.PackedConv.x6_planes, cached on tensor identity) are stale.  Every plane set that exists -- a layer gets one the first
        time the engine's policy picks it, forward or input gradient -- is re-derived in ONE launch (frcnn_refresh_x6_planes)."""
        live = [pk for c in self._tconvs() for pk in (c.pc, getattr(c, "pd", None)) if getattr(pk, "_x6", None) is not None and pk._x6_src is pk.w]
        key = tuple(id(pk._x6) for pk in live)
        if getattr(self, "_x6_key", None) != key:
            jobs = (_lib.X6Job * max(1, len(live)))()
            for j, pk in zip(jobs, live):
                j.w_packed, j.planes_bf16, j.rows, j.kpad = pk.w.data_ptr(), pk._x6.data_ptr(), pk.w.shape[0], pk.w.shape[1]
            self._x6_key, self._x6_jobs = key, jobs
        if live:
            _lib.call("frcnn_refresh_x6_planes", self._x6_jobs, len(live), _stream())

    def _stage(self, host_inputs):
        """Host arrays (any dtype: Keras hands float64 images and bool targets) -> pinned float32 staging memory in ONE
        pass (the cast happens while writing into pinned memory; the image's cast is cut into row bands over a few
        threads -- numpy releases the GIL inside the cast loop -- because one core converts a float64 600x1000 image in
        ~1.1 ms, more than the GPU part of a mixed-precision RPN step leaves to hide it), ready for an asynchronous
        upload.  Two staging areas used alternately, each guarded by the event of its last upload: a deferred step
        (``defer=True``) lets the host stage the NEXT image while this one's step is still running.  The areas are
        flat and grow-only, sized by the largest request so far: real VOC / KITTI training walks through hundreds of
        image shapes, and a per-shape cache would pin (and never return) a fresh 7 MB block for most of them."""
        k = self._pin_next
        self._pin_next = k ^ 1
        pset = self._pin_sets[k]
        views = pset.views([shape for _, shape in host_inputs])
        for pin, (a, shape) in zip(views, host_inputs):
            _cast_into(pin.numpy(), np.asarray(a).reshape(shape))
        return pset, views

    def _run_step(self, host_inputs, skip, defer=False):
        """host_inputs: list of (array, device shape); the arrays may have any dtype (cast to float32 on the way).
        Returns Keras' [total, loss 1, loss 2] -- or, with ``defer``, a PendingLosses whose ``result()`` is that list: the
        step is then only ENQUEUED when this returns, and the caller may prepare the next image meanwhile."""
        assert self.optimizer is not None, "call compile() first"
        sg = self._step_graph(host_inputs) if (STEP_GRAPHS and not skip) else None
        if sg is None:
            out3 = torch.zeros(3, dtype=torch.float32, device="cuda")                      # loss 1, loss 2, sum of squares
            out = [out3[0:1], out3[1:2], out3[2:3]]
        slot = self._loss_ring[self._loss_pos]
        if slot is None:
            slot = self._loss_ring[self._loss_pos] = [torch.empty(3, dtype=torch.float32).pin_memory(), torch.cuda.Event(), None]
        elif slot[2] is not None and slot[2]() is not None:
            slot[2]().result()                          # the ring wrapped around an unread step: read it before its slot is reused
        self._loss_pos = (self._loss_pos + 1) % LOSS_RING
        self._cur = (out3, slot) if sg is None else None
        try:
            if skip:
                self._finish_update()
                with ops.conv_workspace(self._conv_ws):
                    self.params.g.zero_()
                    self._publish_losses()
                    self._update()
            else:
                # Upload and the FROZEN leading part of the base (stem .. last frozen stage: it reads no trainable weight) go
                # to a second stream.  When the caller defers its steps the host is a step ahead, so this part of image i+1
                # runs beside the backward pass of image i, whose short dependent launches leave most of the chip idle -- and,
                # data parallel, beside image i's all-reduce, whose optimiser is enqueued only now (_finish_update); the
                # main stream joins before the first trainable layer.  Same kernels, same order per tensor: bit-identical.
                # inputs that are ALREADY float32 device tensors (the managers' fast paths: rpn_util.rpn_inputs_dev,
                # det_util.get_training_input_dev) skip the cast / staging / upload; they were produced on the manager's stream and
                # carry the event both of this step's streams wait for
                on_dev = [isinstance(a, torch.Tensor) and a.is_cuda for a, _ in host_inputs]
                if sg is not None:
                    self._replay_step(sg, host_inputs, on_dev, slot)
                else:
                    self._eager_step(host_inputs, on_dev, out)
        finally:
            # a step that died half way (OOM, FrcnnError) must not leave its queued weight-gradient jobs to the next
            # step -- possibly another model's -- to launch into this step's buffers
            _PENDING_WGRAD.clear()
            _PENDING_BIAS.clear()
            self._cur = None
        pending = PendingLosses(slot[0], slot[1], self.l2, self.frozen_sumsq)
        slot[2] = weakref.ref(pending)
        return pending if defer else pending.result()


    def _eager_step(self, host_inputs, on_dev, out):
        pset, views = self._stage([hi for hi, d in zip(host_inputs, on_dev) if not d])
        main, side = torch.cuda.current_stream(), _prefix_stream()
        side.wait_event(self._frozen_ready)         # the frozen layers' packed filters (lowered on the build stream)
        arena_pre, arena_main = self._amax_arenas() if F32_ENGINE == "f16x3" else (None, None)
        with torch.cuda.stream(side), ops.conv_workspace(self._conv_ws_prefix), ops.f32_engine(F32_ENGINE), ops.amax_arena(arena_pre):
            ops.amax_begin()
            up = iter(views)
            dev = []
            for (a, shape), d in zip(host_inputs, on_dev):
                if d:
                    assert a.dtype == torch.float32 and a.is_contiguous(), "device inputs of train_on_batch: contiguous float32"
                    ev = getattr(a, "_ready", None)
                    if ev is not None:
                        side.wait_event(ev)
                    a.record_stream(side)
                    dev.append(a.reshape(shape))
                else:
                    dev.append(next(up).to("cuda", non_blocking=True))
            pset.mark_uploaded()
            pre = self._frozen_prefix(dev)
            if pre is not None and getattr(pre, "_amax", None) is not None:
                pre._amax = None                    # (a record of the prefix arena: not read on the main stream)
        self._finish_update()
        main.wait_stream(side)
        for t in dev + ([pre] if pre is not None else []):
            t.record_stream(main)                   # allocated on the side stream, read (and released) under the main one
        with ops.conv_workspace(self._conv_ws), ops.f32_engine(F32_ENGINE), ops.amax_arena(arena_main):
            ops.amax_begin()
            self._device_step(dev, out, pre)

    # ------------------------------------------------------------------ the step as three hipGraphs
    def _graph_key(self, host_inputs):
        return (tuple(shape for _, shape in host_inputs), F32_ENGINE, WGRAD_ENGINE, WGRAD_FLUSH_JOBS, ops.AUTO_TILE,
                id(self._conv_ws), id(self._conv_ws_prefix))

    def _frozen_token(self):
        """The frozen layers' packed filters THEMSELVES (held, not their id()s: an id can be re-used by a later object)."""
        return tuple(u.pc for u in self._frozen_units())

    def _step_graph(self, host_inputs):
        """The captured form of a step on these input shapes, or None (the step then runs eagerly): a shape is captured on its
        (STEP_GRAPH_AFTER + 1)-th step -- the eager ones before it have lowered every lazily built filter form and sized the
        split-K workspaces, so the capture allocates nothing that must outlive it and contains no set-up launch."""
        from . import models
        st = self._graphs
        epoch = models.weights_epoch()
        if st["epoch"] != epoch:
            # some model re-lowered layers since the last step.  A trainer's own packed filters never move (TConv), the frozen
            # layers' could (set_weights on a frozen layer): the captured prefix holds their addresses
            token = self._frozen_token()
            if st["token"] is not None and (len(st["token"]) != len(token) or any(a is not b for a, b in zip(st["token"], token))):
                self.drop_step_graphs()
            st["epoch"], st["token"] = epoch, token
        key = self._graph_key(host_inputs)
        sg = st["graphs"].get(key)
        if sg is not None:
            if next(reversed(st["graphs"])) != key:
                st["graphs"].move_to_end(key)
            return sg
        seen = st["seen"].get(key, 0) + 1
        if seen <= STEP_GRAPH_AFTER:
            if len(st["seen"]) > 4096:
                st["seen"].clear()
            st["seen"][key] = seen
            return None
        evicted = False
        while len(st["graphs"]) >= STEP_GRAPH_SHAPES:
            _, old = st["graphs"].popitem(last=False)
            old.close()
            evicted = True
        if evicted:
            old = None
            torch.cuda.empty_cache()                    # an evicted step's private memory pools go back to the device (scripts/dev/r6_soak_graphs.py:
                                                        # without this the reservation grew by ~80 MB per eviction until the allocator ran dry)
        st["seen"].pop(key, None)
        sg = st["graphs"][key] = self._capture_step([shape for _, shape in host_inputs])
        return sg

    def drop_step_graphs(self):
        """Destroy every captured step (their memory pools go back to the allocator); the next steps run eagerly and re-capture."""
        had = bool(self._graphs["graphs"])
        for sg in self._graphs["graphs"].values():
            sg.close()
        self._graphs["graphs"].clear()
        self._graphs["seen"].clear()
        if had:
            sg = None
            torch.cuda.empty_cache()

    def _capture_step(self, shapes):
        """Capture the device part of a step on inputs of ``shapes`` (see _StepGraph)."""
        from .pipeline import no_gc
        self._finish_update()
        torch.cuda.synchronize()
        sg = _StepGraph()
        has_prefix = getattr(self, "base", None) is not None
        sg.inbox = [torch.empty(shape, dtype=torch.float32, device="cuda") for shape in shapes]
        # the image is read by the prefix graph only, straight from where the feed wrote it; everything else (and the first
        # input of a base-less detector) is read by the step's main part and moves into its own copy at the start of G1
        sg.static = [t if (i == 0 and has_prefix) else torch.empty_like(t) for i, t in enumerate(sg.inbox)]
        sg.out3 = torch.zeros(3, dtype=torch.float32, device="cuda")
        out = [sg.out3[0:1], sg.out3[1:2], sg.out3[2:3]]
        arena_pre, arena_main = self._amax_arenas() if F32_ENGINE == "f16x3" else (None, None)
        global _RECORDER
        cur_saved = self._cur
        recs = []
        try:
            with no_gc(), torch.cuda.stream(_capture_stream()):
                pre_stage = None
                if has_prefix:
                    rec0 = _Pieces()                        # (a pool of its own: G0 of the next image runs beside this step's backward pass)
                    recs.append(rec0)
                    with ops.conv_workspace(self._conv_ws_prefix), ops.f32_engine(F32_ENGINE), ops.amax_arena(arena_pre):
                        rec0.begin("prefix")
                        ops.amax_begin()
                        pre_stage = self._frozen_prefix(sg.static)
                        rec0.end()
                    sg.g0, sg.pre_stage = rec0.items[0][1], pre_stage
                rec = _Pieces()
                recs.append(rec)
                with ops.conv_workspace(self._conv_ws), ops.f32_engine(F32_ENGINE), ops.amax_arena(arena_main):
                    rec.begin("main")
                    for dst, src in zip(sg.static, sg.inbox):
                        if dst is not src:
                            dst.copy_(src)
                    pre = None
                    if pre_stage is not None:
                        pre = torch.empty_like(pre_stage)   # the prefix graph of the NEXT image overwrites pre_stage during this step's backward pass
                        pre.copy_(pre_stage)
                    sg.out3.zero_()
                    ops.amax_begin()
                    self._cur = (sg.out3, None)
                    self._fwd_part(sg.static, out, pre)
                    rec.end()
                    sg.g1 = rec.items[0][1]
                    rec.begin("main")
                    _RECORDER = rec
                    self._bwd_part()
                    flush_weight_grads()
                    _RECORDER = None
                    rec.end()
                sg.bwd = rec.items[1:]
                rec.keep.clear()
        except BaseException:
            _RECORDER = None
            for r in recs:
                r.abort()
            _PENDING_WGRAD.clear()
            _PENDING_BIAS.clear()
            sg.close()
            raise
        finally:
            self._cur = cur_saved
        torch.cuda.synchronize()
        sg.fwd_done = torch.cuda.Event()
        return sg

    def _replay_step(self, sg, host_inputs, on_dev, slot):
        """One step from its captured form: the inputs move into the graphs' input buffers, then G0 (frozen prefix, prefix stream)
        -> G1 (trainable forward, losses, L2 sum) -> the three scalars leave -> the backward pieces (input-gradient chain on this
        stream, each weight-gradient batch on the weight-gradient stream behind the piece that produced its operands) -> exchange,
        optimiser, re-pack (eager launches: the optimiser's scalars and step counter stay ordinary arguments)."""
        pset, views = self._stage([hi for hi, d in zip(host_inputs, on_dev) if not d])
        main, side = torch.cuda.current_stream(), _prefix_stream()
        side.wait_event(self._frozen_ready)
        if sg.used and "nowait" not in _DBG:
            side.wait_event(sg.fwd_done)            # the previous replay's G1 has taken its inputs and the prefix output
        with torch.cuda.stream(side):
            up = iter(views)
            for (a, shape), d, dst in zip(host_inputs, on_dev, sg.inbox):
                if "nocopy" in _DBG:
                    continue
                if d:
                    assert a.dtype == torch.float32 and a.is_contiguous(), "device inputs of train_on_batch: contiguous float32"
                    ev = getattr(a, "_ready", None)
                    if ev is not None:
                        side.wait_event(ev)
                    a.record_stream(side)
                    dst.copy_(a.reshape(shape))
                else:
                    dst.copy_(next(up), non_blocking=True)
            pset.mark_uploaded()
            if sg.g0 is not None:
                sg.g0.replay()
        self._finish_update()
        main.wait_stream(side)
        if sg.loss_sent is not None:
            main.wait_event(sg.loss_sent)           # the previous replay's scalars have left out3
        sg.g1.replay()
        if _TICK:
            sg.out3[2:3].add_(0.0)
        sg.fwd_done.record(main)
        sg.used = True
        self._cur = (sg.out3, slot)
        self._send_losses()
        sg.loss_sent = slot[1]
        wside = None
        for lane, g in sg.bwd:
            if lane == "wgrad":
                wside = _wgrad_stream()
                wside.wait_stream(main)
                with torch.cuda.stream(wside):
                    g.replay()
            else:
                g.replay()
        if wside is not None:
            main.wait_stream(wside)
        with ops.conv_workspace(self._conv_ws):
            self._exchange_and_apply()


class _StepGraph:
    """A training step on one set of input shapes as a handful of LINEAR hipGraphs (see _Pieces).

    G0 (own memory pool): stem + frozen stages of the image in ``inbox[0]`` -> ``pre_stage``; replayed on the prefix stream, beside
    the previous step's backward pass.  G1: the other inputs ``inbox[k]`` -> ``static[k]`` and ``pre_stage`` -> a copy of its own (so
    the next image's G0 and input copies may start as soon as G1 has run: ``fwd_done``), trainable forward, the two loss kernels
    (value + gradient) -> ``out3`` (the L2 sum joins it on the loss stream).  ``bwd`` (G1's pool: they read G1's activations): the input-gradient chain cut at
    every weight-gradient flush, alternating with the weight-gradient batches, which replay on the weight-gradient stream; the bias
    gradients end the last main piece.  Same kernels, same arguments, same order per tensor as the eager step: the weights after N
    replayed steps equal the eager ones bit for bit (tests/test_train_graph_gpu.py)."""

    def __init__(self):
        self.g0 = self.g1 = None
        self.bwd = []
        self.inbox = self.static = self.out3 = self.pre_stage = None
        self.fwd_done = self.loss_sent = None
        self.used = False

    def close(self):
        graphs = [g for _, g in reversed(self.bwd)] + [self.g1, self.g0]
        if any(g is not None for g in graphs):
            torch.cuda.synchronize()
        for g in graphs:
            if g is not None:
                g.reset()
        self.g0 = self.g1 = None
        self.bwd = []
        self.inbox = self.static = self.out3 = self.pre_stage = None


_CAPTURE_STREAM = None


def _capture_stream():
    global _CAPTURE_STREAM
    if _CAPTURE_STREAM is None:
        _CAPTURE_STREAM = torch.cuda.Stream()
    return _CAPTURE_STREAM


class _PinnedSet:
    """One flat, grow-only pinned float32 staging area; ``views(shapes)`` hands out one view per input."""

    def __init__(self):
        self.buf, self.uploaded = None, None

    def views(self, shapes):
        sizes = [int(np.prod(s)) for s in shapes]
        offs = np.cumsum([0] + [-(-n // 64) * 64 for n in sizes])       # 256-byte aligned pieces
        if self.uploaded is not None:
            self.uploaded.synchronize()                 # the last upload out of this area has left
        if self.buf is None or self.buf.numel() < int(offs[-1]):
            self.buf = torch.empty(int(offs[-1] * 1.25) + 64, dtype=torch.float32).pin_memory()
        return [self.buf[int(o):int(o) + n].view(*s) for o, n, s in zip(offs, sizes, shapes)]

    def mark_uploaded(self):
        if self.uploaded is None:
            self.uploaded = torch.cuda.Event()
        self.uploaded.record()


_CAST_POOL = None
CAST_THREADS = int(__import__("os").environ.get("FRCNN_CAST_THREADS", "4"))


def _cast_into(dst, src):
    """dst[...] = src cast to float32 (numpy 'unsafe' casting: bool / float64 / int inputs).  Large arrays go over a
    small thread pool in contiguous bands of the leading non-unit axis."""
    global _CAST_POOL
    if src.size < (1 << 18) or CAST_THREADS <= 1:
        np.copyto(dst, src, casting="unsafe")
        return
    if _CAST_POOL is None:
        from concurrent.futures import ThreadPoolExecutor
        _CAST_POOL = ThreadPoolExecutor(CAST_THREADS)
    d2, s2 = dst.reshape(-1, dst.shape[-1]), src.reshape(-1, src.shape[-1])
    rows = d2.shape[0]
    step = -(-rows // CAST_THREADS)
    futs = [_CAST_POOL.submit(np.copyto, d2[i:i + step], s2[i:i + step], "unsafe") for i in range(0, rows, step)]
    for f in futs:
        f.result()


LOSS_RING = 8
_LIVE_DRIVERS = weakref.WeakSet()


def finish_pending_updates():
    """Enqueue the optimiser of every trainer whose data-parallel all-reduce is still pending (timing loops call this
    before their final synchronisation so the last step's update is inside the timed region)."""
    for d in list(_LIVE_DRIVERS):
        d._finish_update()

_PREFIX_STREAM = None
_LOSS_STREAM = None


def _loss_stream():
    global _LOSS_STREAM
    if _LOSS_STREAM is None:
        _LOSS_STREAM = torch.cuda.Stream()
    return _LOSS_STREAM


def _prefix_stream():
    global _PREFIX_STREAM
    if _PREFIX_STREAM is None:
        _PREFIX_STREAM = torch.cuda.Stream()
    return _PREFIX_STREAM


class PendingLosses:
    """The three scalars of an enqueued training step (loss 1, loss 2, sum of squared trainable weights) on their way
    into pinned host memory; ``result()`` waits for that copy and returns what Keras' train_on_batch returns."""

    def __init__(self, pin, event, l2, frozen_sumsq):
        self._pin, self._event, self._l2, self._frozen = pin, event, l2, frozen_sumsq
        self._value = None

    def result(self):
        if self._value is None:
            self._event.synchronize()
            l1, l2v, sq = (float(v) for v in self._pin.tolist())
            reg_term = self._l2 * (sq + self._frozen) if self._l2 else 0.0
            self._value = [l1 + l2v + reg_term, l1, l2v]
            self._pin = self._event = None
        return self._value


# ----------------------------------------------------------------------------- RPN steps 1 and 3
class RpnTrainer(_StepDriver):
    """rpn_model.compile(...) + rpn_model.train_on_batch(x, [y_class, y_bbreg]) (train_util.py:31-54):
    forward base + heads, the two RPN losses (+ L2), backward through the heads and whatever part of the
    base trains (step 1: ResNet stage 4 / VGG blocks 3-5; step 3: nothing -- freeze_blocks covers the
    whole base, train_rpn_step3.py:59-76), optimiser step.  Returns [total, cls_loss, reg_loss]."""

    def __init__(self, rpn_model, l2=0.0, l2_base=None):
        self.model, self.l2 = rpn_model, l2
        w = rpn_model.weights
        self.A = rpn_model.anchors_per_loc
        head_names = ["rpn_conv1", "rpn_out_cls", "rpn_out_bbreg"]
        train_names = _base_trainable_names(rpn_model.base) + head_names
        self.params = ParamSet(w, train_names)
        self.base = _make_base_train(rpn_model.base, self.params)
        self.base_trains = len(self.base.convs()) > 0
        # mixed precision: rpn_conv1 runs in bf16 like the base; the two output layers (9 / 36 channels: no bf16
        # input-gradient form, negligible work) train in f32 on a widened copy of its output
        self.bf16 = getattr(rpn_model.base.net, "dtype", "f32") == "bf16"
        hd = rpn_model.head
        u_conv, u_cls, u_reg = ((nets.ConvUnit(w, "rpn_conv1", padding="same", act="relu", dtype="bf16"), nets.ConvUnit(w, "rpn_out_cls", act="sigmoid"),
                                 nets.ConvUnit(w, "rpn_out_bbreg")) if self.bf16 else (hd.conv, hd.cls, hd.reg))
        self.rpn_conv = TConv(u_conv, self.params, needs_dgrad=self.base_trains)
        self.rpn_cls = TConv(u_cls, self.params, needs_dgrad=True)
        self.rpn_reg = TConv(u_reg, self.params, needs_dgrad=True)
        # Keras sums the regulariser of every layer that was GIVEN one: the heads always (when l2 != 0), the
        # base only if it was built with regularisers (step 1 yes, step 3 no: train_rpn_step3.py:70)
        base_reg = rpn_model.base.weight_regularizer is not None if l2_base is None else bool(l2_base)
        reg_layers = head_names + (_base_layer_names(rpn_model.base) if base_reg else [])
        self.frozen_sumsq = _reg_sumsq_frozen(w, reg_layers, train_names, l2)
        self.l2_mask_base = base_reg
        assert base_reg or not self.base_trains or not l2, "a trainable base without regularisers next to regularised heads is not supported"
        self.optimizer = None
        self._refresh_jobs = None
        self._init_driver()

    def compile(self, optimizer, loss=None):
        self._finish_update()                   # a pending data-parallel update belongs to the old slots
        self.optimizer = optimizer
        self.params.reset_optimizer()

    def _tconvs(self):
        return self.base.convs() + [self.rpn_conv, self.rpn_cls, self.rpn_reg]

    def _frozen_prefix(self, dev):
        return self.base.forward_frozen(dev[0])

    def forward(self, x, pre=None):
        self.feat = self.base.forward_rest(pre if pre is not None else self.base.forward_frozen(x))
        h = self.rpn_conv.forward(self.feat)
        if self.bf16:
            h = ops.cast_f32(h)
        return self.rpn_cls.forward(h), self.rpn_reg.forward(h), h

    def train_on_batch(self, x, y, skip=False, defer=False):
        """x (1,H,W,3); y = [y_class (1,R,C,2A) bool, y_bbreg (1,R,C,8A) f32].  ``skip`` = this rank has no
        usable image this step: it still joins the all-reduce with zero gradients (train_util.py:112-114).
        ``defer``: return a PendingLosses instead of waiting for the step (see _StepDriver._run_step)."""
        if skip:
            return self._run_step([], True, defer)
        cells = int(np.prod(np.shape(y[0])[:-1]))
        return self._run_step([(x, (1,) + tuple(np.shape(x)[-3:])), (y[0], (cells, 2 * self.A)), (y[1], (cells, 8 * self.A))], False, defer)

    def _fwd_part(self, dev, out, pre=None):
        xd, yc, yr = dev
        loss1, loss2, sq = out
        cls, reg, h = self.forward(xd, pre)
        cells = cls.shape[1] * cls.shape[2]
        g_cls = torch.empty_like(cls)
        g_reg = torch.empty_like(reg)
        ws = torch.empty(2 * int(_lib.load().frcnn_loss_workspace_bytes()), dtype=torch.uint8, device="cuda")     # one half per loss
        half = ctypes.c_void_p(ws.data_ptr() + ws.numel() // 2)
        _lib.call("frcnn_loss_rpn_cls_ws", _p(yc), _p(cls), cells, self.A, _p(loss1), _p(g_cls), _p(ws), _stream())
        _lib.call("frcnn_loss_rpn_reg_ws", _p(yr), _p(reg), cells, self.A, _p(loss2), _p(g_reg), half, _stream())
        self._bwd_state = (g_cls, g_reg, h)

    def _bwd_part(self):
        (g_cls, g_reg, h), self._bwd_state = self._bwd_state, None
        self.rpn_cls.wgrad(g_cls)
        self.rpn_reg.wgrad(g_reg)
        tmp = self.rpn_cls.dgrad(g_cls)
        gh = self.rpn_reg.dgrad(g_reg, residual=tmp, mask=h)
        if self.bf16:
            gh = ops.cast_bf16(gh)
        self.rpn_conv.wgrad(gh)
        if self.base_trains:
            self.base.backward(self.rpn_conv.dgrad(gh, mask=self.feat))

    def sync_weights(self):
        """Write the trained master weights back into the model's Keras-keyed weight dict.  Only the layers that train
        are re-lowered: the frozen layers' packed filters stay (the prefix stream may be reading them)."""
        self._finish_update()
        self.params.export(self.model.weights)
        self.model.invalidate(only=set(self.params.names))


# ----------------------------------------------------------------------------- detector steps 2 and 4
class _ResNetHeadTrain:
    def __init__(self, head, params):
        self.blocks = [TBlock(units, params, needs_input_grad=True, input_is_relu=(i > 0)) for i, units in enumerate(head.blocks)]

    @staticmethod
    def names(head):
        return [u.conv for units in head.blocks for u in units.values()]

    def convs(self):
        return [c for b in self.blocks for c in b.convs()]

    def forward(self, crop):
        c = crop
        for b in self.blocks:
            c = b.forward(c)
        self.h5 = c
        if c.dtype == torch.bfloat16:                           # pooled features and the dense layers stay f32
            return ops.avgpool_bf16(c, 7).reshape(c.shape[0], 1, 1, c.shape[-1])
        return ops.pool2d(c, 7, 7, False)                       # (n,1,1,2048)

    def backward(self, g_pooled):
        """g_pooled (n,1,1,C) -> gradient w.r.t. the RoI crops (n,7,7,Cf)."""
        n = self.h5.shape[0]
        gx = torch.empty_like(self.h5)
        entry = "frcnn_avgpool_bwd_masked_bf16" if self.h5.dtype == torch.bfloat16 else "frcnn_avgpool_bwd_masked"
        _lib.call(entry, _p(g_pooled.contiguous()), _p(self.h5), n, 7, self.h5.shape[-1], _p(gx), _stream())
        for b in reversed(self.blocks):
            gx = b.backward(gx)
        return gx


class _VggHeadTrain:
    def __init__(self, head, params):
        self.fc1 = TConv(head.fc1, params, needs_dgrad=True)
        self.fc2 = TConv(head.fc2, params, needs_dgrad=True)

    @staticmethod
    def names(head):
        return ["fc1", "fc2"]

    def convs(self):
        return [self.fc1, self.fc2]

    def forward(self, crop):
        self.crop_shape = crop.shape
        n = crop.shape[0]
        return self.fc2.forward(self.fc1.forward(crop.reshape(n, 1, 1, -1)))      # (n,1,1,4096)

    def backward(self, g):
        """g: gradient w.r.t. fc2's ReLU output."""
        g = g.contiguous()
        _lib.call("frcnn_relu_bwd_inplace", _p(g), _p(self.fc2.y), g.numel(), _stream())
        self.fc2.wgrad(g)
        g1 = self.fc2.dgrad(g, mask=self.fc1.y)
        self.fc1.wgrad(g1)
        return self.fc1.dgrad(g1).reshape(self.crop_shape)


class DetTrainer(_StepDriver):
    """detector.compile + detector.train_on_batch([image or conv features, rois], [y_cls, y_reg])
    (train_util.py:95-118, 159-182): [base forward,] RoiResizeConv, head, the two detector losses (+ L2),
    backward through the head, the RoI crop (a deterministic gather per feature cell) and the trainable part of the base.
    Step 2 feeds images through the detector's own base; step 4 feeds cached conv features (no base)."""

    def __init__(self, det_model, l2=0.0):
        self.model, self.l2 = det_model, l2
        w = det_model.weights
        self.C = det_model.num_classes
        is_resnet = isinstance(det_model.head, nets.ResNetHead)
        head_cls = _ResNetHeadTrain if is_resnet else _VggHeadTrain
        # dense_class / dense_reg train as ONE merged GEMM; keep a merged master entry
        kc, bc = w["dense_class_%d" % self.C]
        kr, br = w["dense_reg_%d" % self.C]
        self.merged = {"dense": [np.concatenate([kc, kr], axis=1), np.concatenate([bc, br])]}
        base_names = _base_trainable_names(det_model.base) if det_model.base is not None else []
        head_names = head_cls.names(det_model.head)
        train_names = base_names + head_names
        all_w = dict(w)
        all_w.update(self.merged)
        self.params = ParamSet(all_w, train_names + ["dense"])
        self.base = _make_base_train(det_model.base, self.params) if det_model.base is not None else None
        self.base_trains = self.base is not None and len(self.base.convs()) > 0
        self.head = head_cls(det_model.head, self.params)
        self.bf16 = getattr(det_model.head, "dtype", "f32") == "bf16"
        self.dense = TConv(nets.ConvUnit(all_w, "dense"), self.params, needs_dgrad=True)      # f32: 101 outputs, tiny
        reg_layers = head_names + (_base_layer_names(det_model.base) if det_model.base is not None and det_model.base.weight_regularizer is not None else [])
        self.frozen_sumsq = _reg_sumsq_frozen(w, reg_layers, train_names, l2)
        self.optimizer = None
        self._refresh_jobs = None
        self._init_driver()

    def compile(self, optimizer, loss=None):
        self._finish_update()                   # a pending data-parallel update belongs to the old slots
        self.optimizer = optimizer
        self.params.reset_optimizer()

    def _tconvs(self):
        return (self.base.convs() if self.base is not None else []) + self.head.convs() + [self.dense]

    def _frozen_prefix(self, dev):
        return self.base.forward_frozen(dev[0]) if self.base is not None else None

    def forward(self, x, rois, pre=None):
        if self.base is not None:
            self.feat = self.base.forward_rest(pre if pre is not None else self.base.forward_frozen(x))
        else:
            self.feat = x
        if self.bf16 and self.feat.dtype != torch.bfloat16:     # step 4: cached f32 conv features feed a bf16 head
            self.feat = ops.cast_bf16(self.feat)
        crop = (ops.roi_crop_resize_bf16 if self.bf16 else ops.roi_crop_resize)(self.feat, rois, 7)
        self.pooled = self.head.forward(crop)
        y = self.dense.forward(self.pooled)                    # (n,1,1,C+4(C-1))
        y2 = y.reshape(y.shape[0], -1)
        cls, reg = ops.dense_heads_split(y2, self.C)
        return cls, reg, y2

    def train_on_batch(self, x, y, skip=False, defer=False):
        """x = [image (1,H,W,3) or conv features (1,R,C,Cf), rois (1,n,4)]; y = [y_class (1,n,C), y_bbreg (1,n,8(C-1))]."""
        if skip:
            return self._run_step([], True, defer)
        n, C, K4 = int(np.size(x[1])) // 4, self.C, 4 * (self.C - 1)
        return self._run_step([(x[0], (1,) + tuple(np.shape(x[0])[-3:])), (x[1], (n, 4)), (y[0], (n, C)), (y[1], (n, 2 * K4))], False, defer)

    def _fwd_part(self, dev, out, pre=None):
        xd, rois, yc, yr = dev
        loss1, loss2, sq = out
        n, C, K4 = rois.shape[0], self.C, 4 * (self.C - 1)
        cls, reg, y2 = self.forward(xd, rois, pre)
        g = torch.empty((n, C + K4), dtype=torch.float32, device="cuda")     # [d logits | d reg]
        _lib.call("frcnn_loss_det_cls", _p(yc), _p(cls), n, C, _p(loss1), _p(g), C + K4, _stream())
        _lib.call("frcnn_loss_det_reg", _p(yr), _p(reg), n, C - 1, _p(loss2), ctypes.c_void_p(g.data_ptr() + 4 * C), C + K4, _stream())
        self._bwd_state = (g, rois)

    def _bwd_part(self):
        (g, rois), self._bwd_state = self._bwd_state, None
        n, C, K4 = rois.shape[0], self.C, 4 * (self.C - 1)
        g4 = g.reshape(n, 1, 1, C + K4)
        self.dense.wgrad(g4)
        gcrop = self.head.backward(self.dense.dgrad(g4))
        if self.base_trains and self.bf16:
            gfeat = ops.cast_bf16(ops.roi_crop_resize_bwd_bf16(gcrop, rois, self.feat.shape[1], self.feat.shape[2]))   # f32 gather, then bf16
            _lib.call("frcnn_relu_bwd_inplace_bf16", _p(gfeat), _p(self.feat), gfeat.numel(), _stream())
            self.base.backward(gfeat.reshape(self.feat.shape))
        elif self.base_trains:
            gfeat = ops.roi_crop_resize_bwd(gcrop, rois, self.feat.shape[1], self.feat.shape[2])
            _lib.call("frcnn_relu_bwd_inplace", _p(gfeat), _p(self.feat), gfeat.numel(), _stream())
            self.base.backward(gfeat.reshape(self.feat.shape))

    def sync_weights(self):
        self._finish_update()
        w = self.model.weights
        self.params.export(w)                                    # per-layer entries (+ the merged "dense")
        dense = w.pop("dense")
        C = self.C
        w["dense_class_%d" % C] = [dense[0][:, :C].copy(), dense[1][:C].copy()]
        w["dense_reg_%d" % C] = [dense[0][:, C:].copy(), dense[1][C:].copy()]
        self.model.invalidate(only=set(self.params.names))
        self.model.head.dense = nets._MergedDense(w, C)
