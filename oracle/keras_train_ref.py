"""One Keras training step of the reference, restated with torch autograd on the CPU (float64).
TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED for the same reason as keras_ref.py.

compile + train_on_batch as train_util.py:29-56 / 93-120 drive them:
  total = loss_1 + loss_2 + sum of the l2 regularisers of EVERY regularised layer of the model
          (kernel and bias; frozen layers included -- their penalty is a constant, resnet.py:26-27)
  gradients w.r.t. the trainable weights only (freeze_blocks, resnet.py:395, 432; BatchNorm frozen)
  Keras SGD(momentum):  v = m*v - lr*g; w += v      Keras Adam: lr_t = lr*sqrt(1-b2^t)/(1-b1^t) ...
Loss functions: loss_functions.py:15-76 with the Keras-2.0.8/TF backend formulas [3P]:
  K.binary_crossentropy: clip p to [1e-7, 1-1e-7], logits = log(p/(1-p)), sigmoid_cross_entropy_with_logits;
  categorical_crossentropy: p /= sum(p); clip; -sum(y*log p); a tensor-valued loss is averaged by Keras.
"""
import numpy as np
import torch

from .keras_ref import KerasGraphs

EPS = 1e-7


def cls_loss_rpn(y_true, y_pred, A):
    sel, z = y_true[..., :A], y_true[..., A:]
    p = y_pred.clamp(EPS, 1 - EPS)
    x = torch.log(p / (1 - p))
    bce = x.clamp(min=0) - x * z + torch.log1p(torch.exp(-x.abs()))
    return (sel * bce).sum() / 256.0                                   # loss_functions.py:24


def smooth_l1(d):
    a = d.abs()
    small = (a <= 1.0).to(d.dtype)
    return small * (0.5 * a * a) + (1 - small) * (a - 0.5)


def bbreg_loss_rpn(y_true, y_pred, A):
    mask, t = y_true[..., :4 * A], y_true[..., 4 * A:]
    tensor_loss = 10.0 * mask * smooth_l1(t - y_pred).sum() / 2400.0   # :44 -- mask OUTSIDE the sum
    return tensor_loss.mean()                                          # Keras averages a tensor-valued loss


def bbreg_loss_det(y_true, y_pred, K):
    mask, t = y_true[..., :4 * K], y_true[..., 4 * K:]
    return (mask * smooth_l1(t - y_pred)).sum() / (1e-4 + mask).sum()  # :65


def cls_loss_det(y_true, y_pred):
    p = y_pred / y_pred.sum(dim=-1, keepdim=True)
    p = p.clamp(EPS, 1 - EPS)
    return (-(y_true * torch.log(p)).sum(dim=-1)).mean()               # :76


def _block_plan(depth):
    if depth == 101:
        return {2: ["a", "b", "c"], 3: ["a", "b1", "b2", "b3"], 4: ["a"] + ["b%d" % i for i in range(1, 23)]}
    return {2: list("abc"), 3: list("abcd"), 4: list("abcdef")}


def conv_layer_names(depth, stages):
    names = []
    for s in stages:
        blocks = _block_plan(depth)[s] if s < 5 else list("abc")
        for b in blocks:
            for suf in ("2a", "2b", "2c") + (("1",) if b == "a" else ()):
                names.append("res%d%s_branch%s" % (s, b, suf))
    return names


class Optim:
    """Keras 2.0.8 SGD(momentum) / Adam (optimizers.py, get_updates).  ``t`` is the optimiser object's ``iterations``
    variable: created in __init__, incremented by every update, NOT reset when the model is compiled again;
    ``recompile()`` is what a new ``model.compile`` + first ``train_on_batch`` does to the same optimiser object
    (train_util.py:29-33): the slot variables (momentum / Adam moments) are created afresh, the counter runs on."""

    def __init__(self, kind, lr, momentum=0.9):
        self.kind, self.lr, self.momentum, self.t, self.slots = kind, lr, momentum, 0, {}

    def recompile(self, lr=None):
        self.slots = {}
        if lr is not None:
            self.lr = lr

    def step(self, params, grads):
        self.t += 1
        new = {}
        for k, p in params.items():
            g = grads[k]
            if self.kind == "sgd":
                v = self.momentum * self.slots.get(k, torch.zeros_like(p)) - self.lr * g
                self.slots[k] = v
                new[k] = p + v
            else:
                m, v = self.slots.get(k, (torch.zeros_like(p), torch.zeros_like(p)))
                lr_t = self.lr * np.sqrt(1 - 0.999 ** self.t) / (1 - 0.9 ** self.t)
                m = 0.9 * m + 0.1 * g
                v = 0.999 * v + 0.001 * g * g
                self.slots[k] = (m, v)
                new[k] = p - lr_t * m / (v.sqrt() + 1e-8)
        return new


def _prepare(weights, trainable, dtype):
    w, params = {}, {}
    for name, arrs in weights.items():
        ts = []
        for i, a in enumerate(arrs):
            t = torch.tensor(np.asarray(a), dtype=dtype)
            if name in trainable:
                t.requires_grad_(True)
                params[(name, i)] = t
            ts.append(t)
        w[name] = ts
    return w, params


def _l2_penalty(w, names, l2):
    tot = 0.0
    for n in names:
        for t in w[n]:
            tot = tot + (t * t).sum()
    return l2 * tot


def _finish(weights, w, params, total, optim):
    grads = torch.autograd.grad(total, list(params.values()))
    grads = dict(zip(params.keys(), grads))
    new = optim.step({k: v.detach() for k, v in params.items()}, grads)
    out = {n: [np.asarray(a) for a in arrs] for n, arrs in weights.items()}
    for (name, i), t in new.items():
        out[name][i] = t.numpy()
    return out, grads


VGG_BLOCK_CONVS = {1: 2, 2: 2, 3: 3, 4: 3, 5: 3}


def vgg_conv_names(blocks):
    return ["block%d_conv%d" % (b, i) for b in blocks for i in range(1, VGG_BLOCK_CONVS[b] + 1)]


def rpn_train_step(weights, x, y_class, y_bbreg, A, optim, depth=50, freeze_blocks=(1, 2, 3), l2=0.0, dtype=torch.float64,
                   arch="resnet", l2_base=True, mixed=False):
    """One compile()d train_on_batch of an RPN model (train_rpn_step1.py:59-90; step 3 = every base block
    frozen and only the heads regularised, train_rpn_step3.py:59-76).  Returns (new_weights, [total, l1, l2], grads)."""
    heads = ["rpn_conv1", "rpn_out_cls", "rpn_out_bbreg"]
    if arch == "resnet":
        base_train = conv_layer_names(depth, [s for s in (2, 3, 4) if s not in freeze_blocks])
        base_all = ["conv1"] + conv_layer_names(depth, (2, 3, 4))
    else:
        base_train = vgg_conv_names([b for b in (1, 2, 3, 4, 5) if b not in freeze_blocks])
        base_all = vgg_conv_names((1, 2, 3, 4, 5))
    w, params = _prepare(weights, set(base_train) | set(heads), dtype)
    g = KerasGraphs(w, dtype, mixed=mixed)                  # mixed: the bf16 storage model of BASELINE configs[4] (keras_ref)
    feat = g.resnet_base(x, depth) if arch == "resnet" else g.vgg_base(x)
    cls, reg = g.rpn(feat)
    yc = torch.tensor(np.asarray(y_class, dtype=np.float64), dtype=dtype)
    yr = torch.tensor(np.asarray(y_bbreg, dtype=np.float64), dtype=dtype)
    l1 = cls_loss_rpn(yc, cls, A)
    l2v = bbreg_loss_rpn(yr, reg, A)
    reg_names = (base_all if l2_base else []) + heads
    total = l1 + l2v + (_l2_penalty(w, reg_names, l2) if l2 else 0.0)
    new, grads = _finish(weights, w, params, total, optim)
    return new, [float(total.detach()), float(l1.detach()), float(l2v.detach())], grads


def det_train_step(weights, x, rois, y_class, y_bbreg, C, optim, depth=50, freeze_blocks=(1, 2, 3), l2=0.0, dtype=torch.float64,
                   arch="resnet", with_base=True, mixed=False):
    """One train_on_batch of a detector model.  with_base=True: step 2 (image in, own base; train_det_step2.py:78-87);
    with_base=False: step 4 (x = conv features, only the head trains and is regularised; train_det_step4.py:67-95)."""
    dn = ["dense_class_%d" % C, "dense_reg_%d" % C]
    if arch == "resnet":
        base_train = conv_layer_names(depth, [s for s in (2, 3, 4) if s not in freeze_blocks]) if with_base else []
        base_all = ["conv1"] + conv_layer_names(depth, (2, 3, 4))
        head = conv_layer_names(depth, [5])
    else:
        base_train = vgg_conv_names([b for b in (1, 2, 3, 4, 5) if b not in freeze_blocks]) if with_base else []
        base_all = vgg_conv_names((1, 2, 3, 4, 5))
        head = ["fc1", "fc2"]
    w, params = _prepare(weights, set(base_train) | set(head) | set(dn), dtype)
    g = KerasGraphs(w, dtype, mixed=mixed)
    if with_base:
        feat = g.resnet_base(x, depth) if arch == "resnet" else g.vgg_base(x)
    else:
        feat = g.q(torch.tensor(np.asarray(x), dtype=dtype))          # step 4 (mixed): cached f32 features are cast once
    rr = np.asarray(rois).reshape(-1, 4)
    if arch == "resnet":
        cls, reg = g.resnet_classifier_logits(feat, rr, C, depth)
    else:
        from .keras_ref import roi_resize_torch
        t = roi_resize_torch(feat[0], rr, 7).reshape(len(rr), -1)
        t = g._dense(t, "fc1").clamp(min=0)
        t = g._dense(t, "fc2").clamp(min=0)
        cls = torch.softmax(g._dense(t, dn[0]), dim=1)
        reg = g._dense(t, dn[1])
    yc = torch.tensor(np.asarray(y_class, dtype=np.float64), dtype=dtype)[0]
    yr = torch.tensor(np.asarray(y_bbreg, dtype=np.float64), dtype=dtype)[0]
    l1 = cls_loss_det(yc, cls)
    l2v = bbreg_loss_det(yr, reg, C - 1)
    reg_names = (base_all if with_base else []) + head + dn
    total = l1 + l2v + (_l2_penalty(w, reg_names, l2) if l2 else 0.0)
    new, grads = _finish(weights, w, params, total, optim)
    return new, [float(total.detach()), float(l1.detach()), float(l2v.detach())], grads
