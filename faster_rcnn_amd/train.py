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

import numpy as np
import torch

from . import _lib, nets, ops
from .ops import _p, _stream


# ----------------------------------------------------------------------------- optimisers (args_util.py:48-59)
class SGD:
    def __init__(self, lr=1e-3, momentum=0.9):
        self.lr, self.momentum = lr, momentum


class Adam:
    def __init__(self, lr=1e-3, beta_1=0.9, beta_2=0.999, epsilon=1e-8):
        self.lr, self.beta_1, self.beta_2, self.epsilon = lr, beta_1, beta_2, epsilon


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
        self.t = 0

    def reset_optimizer(self):
        """Keras re-creates the optimiser slots on every compile() (train_util.py:31-33)."""
        self.slots = [torch.zeros_like(self.w), torch.zeros_like(self.w)]
        self.t = 0

    def step(self, opt, l2, grad_scale=1.0):
        self.t += 1
        if isinstance(opt, Adam):
            _lib.call("frcnn_adam", _p(self.w), _p(self.g), _p(self.slots[0]), _p(self.slots[1]), self.total, float(opt.lr),
                      float(opt.beta_1), float(opt.beta_2), float(opt.epsilon), self.t, float(l2), float(grad_scale), _stream())
        else:
            _lib.call("frcnn_sgd_momentum", _p(self.w), _p(self.g), _p(self.slots[0]), self.total, float(opt.lr),
                      float(opt.momentum), float(l2), float(grad_scale), _stream())

    def sumsq(self):
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
        self.x = self.y = None
        if not self.trainable:
            if unit.pc is None:
                unit.lower()
            self.pc = unit.pc
            if needs_dgrad:
                self.pd = ops.PackedDgrad(self._kernel4d(np.asarray(unit.weights[unit.conv][0], dtype=np.float32)), self.pc.scale)
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
        kp = _lib.load().frcnn_conv_packed_k(self.kh, self.kw, self.cin)
        self.pc = ops.PackedConv.__new__(ops.PackedConv)
        self.pc.kh, self.pc.kw, self.pc.cin, self.pc.cout = self.kh, self.kw, self.cin, self.cout
        self.pc.w = torch.empty((self.cout, kp), dtype=torch.float32, device="cuda")
        self.pc.scale = self.scale
        self.pc.shift = torch.empty(self.cout, dtype=torch.float32, device="cuda")
        if needs_dgrad:
            self.pd = ops.PackedDgrad.__new__(ops.PackedDgrad)
            self.pd.kh, self.pd.kw, self.pd.cin, self.pd.cout = self.kh, self.kw, self.cout, self.cin
            self.pd.w = torch.empty((self.cin, _lib.load().frcnn_conv_packed_k(self.kh, self.kw, self.cout)), dtype=torch.float32, device="cuda")
            self.pd.scale = self.pd.shift = None
        self.refresh()

    @staticmethod
    def _kernel4d(k):
        return k.reshape(1, 1, *k.shape) if k.ndim == 2 else k

    def refresh(self):
        """Re-pack from the master weights (after an optimiser step)."""
        if not self.trainable:
            return
        _lib.call("frcnn_pack_conv_weights", _p(self.k4), self.kh, self.kw, self.cin, self.cout, _p(self.pc.w), _stream())
        _lib.call("frcnn_fold_bias", _p(self.bv), _p(self.scale), _p(self.const), _p(self.pc.shift), self.cout, _stream())
        if self.needs_dgrad:
            _lib.call("frcnn_pack_conv_weights_dgrad", _p(self.k4), _p(self.scale), self.kh, self.kw, self.cin, self.cout, _p(self.pd.w), _stream())

    def forward(self, x, residual=None):
        self.x = x
        self.y = ops.conv2d(x, self.pc, self.u.stride, self.u.padding, self.u.act, residual)
        return self.y

    def wgrad(self, g):
        """g: gradient w.r.t. this layer's post-BN pre-activation output."""
        if self.trainable:
            ops.conv2d_wgrad(self.x, g, self.kh, self.kw, self.u.stride, self.u.padding, scale=self.scale,
                             dw=self.gk4, dbias=self.gb, want_bias=self.gb is not None)

    def dgrad(self, g, residual=None, mask=None):
        return ops.conv2d_dgrad(g, self.pd, self.u.padding, residual=residual, mask=mask)


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


def _sync_grads(params):
    """Data-parallel exchange: ONE all-reduce (sum) of the flat gradient buffer over RCCL."""
    from . import dp
    return dp.allreduce_sum_(params.g)


def _l2_of(weights, names):
    """sum(w^2) of frozen regularised layers: a constant of the run (kernel and bias, resnet.py:26-27)."""
    return float(sum(float((np.asarray(a, dtype=np.float64) ** 2).sum()) for n in names for a in weights[n]))


def _regularised_conv_names(weights):
    return [n for n in weights if n.startswith(("conv1", "res", "rpn_", "dense_", "fc", "block")) and not n.startswith("bn")]


# ----------------------------------------------------------------------------- RPN step 1
class RpnTrainer:
    """rpn_model.compile(...) + rpn_model.train_on_batch(x, [y_class, y_bbreg]) (train_util.py:31-54)
    for a ResNet-50/101 RPN: forward conv1..res4f + heads, the two RPN losses (+ L2), backward through
    stage 4 and the heads, optimiser step.  Returns [total, cls_loss, reg_loss] like Keras."""

    def __init__(self, rpn_model, l2=0.0):
        base = rpn_model.base.net
        assert isinstance(base, nets.ResNetBase), "training is implemented for the ResNet graphs"
        self.model, self.l2 = rpn_model, l2
        w = rpn_model.weights
        self.A = rpn_model.anchors_per_loc
        freeze = set(rpn_model.base.freeze_blocks)
        from .weights import resnet_block_names
        names = resnet_block_names(base.depth)
        train_names = []
        for (stage, block, _), units in zip(names, base.blocks):
            if stage not in freeze:
                train_names += [u.conv for u in units.values()]
        train_names += ["rpn_conv1", "rpn_out_cls", "rpn_out_bbreg"]
        self.params = ParamSet(w, train_names)
        self.frozen_blocks, self.blocks = [], []
        first_trainable = True
        for (stage, block, _), units in zip(names, base.blocks):
            if stage in freeze:
                self.frozen_blocks.append(units)
            else:
                self.blocks.append(TBlock(units, self.params, needs_input_grad=not first_trainable))
                first_trainable = False
        self.stem = base.stem
        self.rpn_conv = TConv(rpn_model.head.conv, self.params, needs_dgrad=True)
        self.rpn_cls = TConv(rpn_model.head.cls, self.params, needs_dgrad=True)
        self.rpn_reg = TConv(rpn_model.head.reg, self.params, needs_dgrad=True)
        reg_names = [n for n in _regularised_conv_names(w) if n in w and not n.startswith(("dense_", "fc")) and not n.startswith("res5")]
        self.frozen_sumsq = _l2_of(w, [n for n in reg_names if n not in train_names]) if l2 else 0.0
        self.optimizer = None

    def compile(self, optimizer, loss=None):
        self.optimizer = optimizer
        self.params.reset_optimizer()

    def _tconvs(self):
        out = []
        for b in self.blocks:
            out += b.convs()
        return out + [self.rpn_conv, self.rpn_cls, self.rpn_reg]

    def forward(self, x):
        t = self.stem(x)
        t = ops.pool2d(t, 3, 2, True)
        for units in self.frozen_blocks:
            t = nets.run_block(units, t)
        for b in self.blocks:
            t = b.forward(t)
        self.feat = t
        h = self.rpn_conv.forward(t)
        return self.rpn_cls.forward(h), self.rpn_reg.forward(h), h

    def train_on_batch(self, x, y, skip=False):
        """x (1,H,W,3); y = [y_class (1,R,C,2A) bool, y_bbreg (1,R,C,8A) f32].  ``skip`` = this rank has no
        usable image this step: it still joins the all-reduce with zero gradients (train_util.py:112-114)."""
        assert self.optimizer is not None, "call compile() first"
        p = self.params
        loss1 = torch.zeros(1, dtype=torch.float32, device="cuda")
        loss2 = torch.zeros(1, dtype=torch.float32, device="cuda")
        if skip:
            p.g.zero_()
        else:
            xd = nets.to_device_image(x)
            cls, reg, h = self.forward(xd)
            cells = cls.shape[1] * cls.shape[2]
            yc = torch.from_numpy(np.ascontiguousarray(y[0], dtype=np.float32)).cuda().reshape(cells, 2 * self.A)
            yr = torch.from_numpy(np.ascontiguousarray(y[1], dtype=np.float32)).cuda().reshape(cells, 8 * self.A)
            g_cls = torch.empty_like(cls)
            g_reg = torch.empty_like(reg)
            _lib.call("frcnn_loss_rpn_cls", _p(yc), _p(cls), cells, self.A, _p(loss1), _p(g_cls), _stream())
            _lib.call("frcnn_loss_rpn_reg", _p(yr), _p(reg), cells, self.A, _p(loss2), _p(g_reg), _stream())
            self.rpn_cls.wgrad(g_cls)
            self.rpn_reg.wgrad(g_reg)
            tmp = self.rpn_cls.dgrad(g_cls)
            gh = self.rpn_reg.dgrad(g_reg, residual=tmp, mask=h)
            self.rpn_conv.wgrad(gh)
            g = self.rpn_conv.dgrad(gh, mask=self.feat)
            for b in reversed(self.blocks):
                g = b.backward(g)
        sq = p.sumsq() if self.l2 else None
        scale = _sync_grads(p)
        p.step(self.optimizer, self.l2, scale)
        for c in self._tconvs():
            c.refresh()
        l1, l2v = float(loss1.item()), float(loss2.item())
        reg_term = self.l2 * (float(sq.item()) + self.frozen_sumsq) if self.l2 else 0.0
        return [l1 + l2v + reg_term, l1, l2v]

    def sync_weights(self):
        """Write the trained master weights back into the model's Keras-keyed weight dict."""
        self.params.export(self.model.weights)
        self.model.invalidate()


# ----------------------------------------------------------------------------- detector step 2
class DetTrainer:
    """detector.compile + detector.train_on_batch([image, rois], [y_cls, y_reg]) (train_util.py:95-118):
    base forward (stage 4 trainable), RoiResizeConv, stage 5, average pool, dense x2, the two detector
    losses (+ L2), backward through the head, the RoI crop (atomic scatter) and stage 4."""

    def __init__(self, det_model, l2=0.0):
        assert det_model.base is not None and isinstance(det_model.base.net, nets.ResNetBase)
        self.model, self.l2 = det_model, l2
        w = det_model.weights
        base = det_model.base.net
        self.C = det_model.num_classes
        freeze = set(det_model.base.freeze_blocks)
        from .weights import resnet_block_names
        names = resnet_block_names(base.depth)
        # dense_class / dense_reg train as ONE merged GEMM; keep a merged master entry
        kc, bc = w["dense_class_%d" % self.C]
        kr, br = w["dense_reg_%d" % self.C]
        self.merged = {"dense": [np.concatenate([kc, kr], axis=1), np.concatenate([bc, br])]}
        train_names = []
        for (stage, block, _), units in zip(names, base.blocks):
            if stage not in freeze:
                train_names += [u.conv for u in units.values()]
        for units in det_model.head.blocks:
            train_names += [u.conv for u in units.values()]
        all_w = dict(w)
        all_w.update(self.merged)
        self.params = ParamSet(all_w, train_names + ["dense"])
        self.stem, self.frozen_blocks, self.blocks = base.stem, [], []
        first = True
        for (stage, block, _), units in zip(names, base.blocks):
            if stage in freeze:
                self.frozen_blocks.append(units)
            else:
                self.blocks.append(TBlock(units, self.params, needs_input_grad=not first))
                first = False
        self.head_blocks = [TBlock(units, self.params, needs_input_grad=True, input_is_relu=(i > 0))
                            for i, units in enumerate(det_model.head.blocks)]
        self.dense = TConv(nets.ConvUnit(all_w, "dense"), self.params, needs_dgrad=True)
        reg_names = [n for n in _regularised_conv_names(w) if not n.startswith("rpn_")]
        self.frozen_sumsq = _l2_of(w, [n for n in reg_names if n not in train_names and not n.startswith("dense_")]) if l2 else 0.0
        self.optimizer = None

    def compile(self, optimizer, loss=None):
        self.optimizer = optimizer
        self.params.reset_optimizer()

    def _tconvs(self):
        out = []
        for b in self.blocks + self.head_blocks:
            out += b.convs()
        return out + [self.dense]

    def forward(self, x, rois):
        t = self.stem(x)
        t = ops.pool2d(t, 3, 2, True)
        for units in self.frozen_blocks:
            t = nets.run_block(units, t)
        for b in self.blocks:
            t = b.forward(t)
        self.feat = t
        c = ops.roi_crop_resize(t, rois, 7)
        for b in self.head_blocks:
            c = b.forward(c)
        self.h5 = c
        pooled = ops.pool2d(c, 7, 7, False)
        y = self.dense.forward(pooled)                         # (n,1,1,C+4(C-1))
        n = y.shape[0]
        y2 = y.reshape(n, -1)
        return ops.softmax_rows(y2, self.C), y2[:, self.C:].contiguous(), y2

    def train_on_batch(self, x, y, skip=False):
        """x = [image (1,H,W,3), rois (1,n,4)]; y = [y_class (1,n,C), y_bbreg (1,n,8(C-1))]."""
        assert self.optimizer is not None, "call compile() first"
        p = self.params
        loss1 = torch.zeros(1, dtype=torch.float32, device="cuda")
        loss2 = torch.zeros(1, dtype=torch.float32, device="cuda")
        if skip:
            p.g.zero_()
        else:
            xd = nets.to_device_image(x[0])
            rois = torch.from_numpy(np.ascontiguousarray(x[1], dtype=np.float32)).cuda().reshape(-1, 4)
            n, C, K4 = rois.shape[0], self.C, 4 * (self.C - 1)
            cls, reg, y2 = self.forward(xd, rois)
            yc = torch.from_numpy(np.ascontiguousarray(y[0], dtype=np.float32)).cuda().reshape(n, C)
            yr = torch.from_numpy(np.ascontiguousarray(y[1], dtype=np.float32)).cuda().reshape(n, 2 * K4)
            g = torch.empty((n, C + K4), dtype=torch.float32, device="cuda")     # [d logits | d reg]
            _lib.call("frcnn_loss_det_cls", _p(yc), _p(cls), n, C, _p(loss1), _p(g), C + K4, _stream())
            _lib.call("frcnn_loss_det_reg", _p(yr), _p(reg), n, C - 1, _p(loss2), ctypes.c_void_p(g.data_ptr() + 4 * C), C + K4, _stream())
            g4 = g.reshape(n, 1, 1, C + K4)
            self.dense.wgrad(g4)
            gp = self.dense.dgrad(g4)                                           # (n,1,1,2048)
            gx = torch.empty_like(self.h5)
            _lib.call("frcnn_avgpool_bwd_masked", _p(gp), _p(self.h5), n, 7, self.h5.shape[-1], _p(gx), _stream())
            for b in reversed(self.head_blocks):
                gx = b.backward(gx)
            gfeat = ops.roi_crop_resize_bwd(gx, rois, self.feat.shape[1], self.feat.shape[2])
            _lib.call("frcnn_relu_bwd_inplace", _p(gfeat), _p(self.feat), gfeat.numel(), _stream())
            gb = gfeat.reshape(self.feat.shape)
            for b in reversed(self.blocks):
                gb = b.backward(gb)
        sq = p.sumsq() if self.l2 else None
        scale = _sync_grads(p)
        p.step(self.optimizer, self.l2, scale)
        for c in self._tconvs():
            c.refresh()
        l1, l2v = float(loss1.item()), float(loss2.item())
        reg_term = self.l2 * (float(sq.item()) + self.frozen_sumsq) if self.l2 else 0.0
        return [l1 + l2v + reg_term, l1, l2v]

    def sync_weights(self):
        w = self.model.weights
        self.params.export(self.merged if False else w)          # per-layer entries
        dense = w.pop("dense")
        C = self.C
        w["dense_class_%d" % C] = [dense[0][:, :C].copy(), dense[1][:C].copy()]
        w["dense_reg_%d" % C] = [dense[0][:, C:].copy(), dense[1][C:].copy()]
        self.model.invalidate()
        self.model.head.dense = nets._MergedDense(w, C)
