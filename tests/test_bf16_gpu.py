"""bf16 conv path (BASELINE configs[3]: ResNet-101, bf16 conv + fp32 NMS).

Tolerances: a bf16 value carries 8 significant bits (relative rounding 2^-9 = 2e-3).  A single conv
is compared against an f64 conv of the SAME bf16-rounded operands (isolates the kernel: <= 1e-2 of
the output scale after one bf16 store); the 101-layer network against the f64 oracle on unrounded
weights, measured against the TENSOR's scale (activations reach ~30, so one bf16 ulp of a large value
is ~0.1 absolute and lands on small neighbours through the next layer's sums): relative RMS error
<= 2e-2 (rounding noise of ~100 stored activations as a random walk: 2e-3 * sqrt(100)) and max
absolute error <= 5e-2 * max|x|.  NMS / decode stay fp32/int exact given the scores."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.mark.parametrize("case", [(1, 15, 22, 64, 64, 1, 1, "valid", 0), (1, 15, 22, 128, 192, 3, 1, "same", 0), (2, 7, 7, 512, 512, 3, 1, "same", 1),
                                  (1, 30, 41, 256, 128, 1, 2, "valid", 2), (1, 19, 23, 512, 36, 1, 1, "valid", 3),
                                  (1, 15, 22, 128, 192, 3, 1, "same", 302),        # split-K: 3 slices of 18 chunks, mid-filter starts
                                  (1, 38, 94, 1024, 256, 1, 1, "valid", 0),        # R101 stage-4 reduce at C4 size: auto split-K
                                  (1, 38, 94, 256, 256, 3, 1, "same", 0),          # R101 stage-4 3x3
                                  (5, 1, 1, 2048, 64, 1, 1, "valid", 802),         # 8 slices of 32 chunks, 5 valid rows
                                  (2, 7, 7, 512, 512, 3, 1, "same", 41), (1, 15, 22, 128, 192, 3, 1, "same", 42),    # 8-wave tiles
                                  (1, 30, 41, 256, 128, 1, 2, "valid", 43), (3, 7, 7, 512, 200, 3, 1, "same", 44),   # 128x64 8-wave, 16-wave
                                  # the 256-wide tiles staged straight into LDS (45: 256x256, 46: 128x256): a single ragged tile
                                  # of 3x3 halo rows, row / column tails, stride 2, many k-chunks
                                  (3, 7, 7, 512, 256, 3, 1, "same", 45), (2, 15, 22, 128, 512, 3, 1, "same", 46),
                                  (1, 30, 41, 256, 320, 1, 2, "valid", 45), (1, 30, 41, 256, 320, 1, 2, "valid", 46),
                                  (6, 14, 14, 1024, 512, 1, 1, "valid", 46), (6, 14, 14, 256, 1024, 3, 1, "same", 45),
                                  (2, 15, 22, 128, 192, 3, 1, "same", 47), (6, 14, 14, 256, 1024, 3, 1, "same", 47),     # 128x128 (the auto pick)
                                  (1, 30, 41, 256, 320, 1, 2, "valid", 48), (3, 7, 7, 512, 200, 3, 1, "same", 48),      # 64x64
                                  # round 3 dev codes: 49 = the 128x128 tile on a RING of four LDS buffers (three chunks in flight),
                                  # 60 = row strips (A resident in LDS, the workgroup walks all column tiles; 1x1 stride 1 only:
                                  # other shapes fall back to 47): one / two / many k chunks, ragged rows and columns
                                  (2, 15, 22, 128, 192, 3, 1, "same", 49), (6, 14, 14, 256, 1024, 3, 1, "same", 49), (1, 19, 23, 64, 136, 1, 1, "valid", 49),
                                  (1, 19, 23, 64, 136, 1, 1, "valid", 60), (3, 37, 41, 256, 1024, 1, 1, "valid", 60), (2, 33, 35, 512, 328, 1, 1, "valid", 60),
                                  (1, 30, 41, 256, 320, 1, 2, "valid", 60)])
def test_conv2d_bf16(case):
    from faster_rcnn_amd import ops
    from oracle import keras_ref
    n, h, w, cin, cout, k, stride, padding, tile = case
    rs = np.random.RandomState(abs(hash(case)) % (2 ** 31))
    bf = lambda a: torch.from_numpy(a).to(torch.bfloat16)
    x = bf(rs.randn(n, h, w, cin).astype(np.float32))
    wt = bf((rs.randn(k, k, cin, cout) * np.sqrt(2.0 / (k * k * cin))).astype(np.float32))
    scale = (1 + 0.1 * rs.randn(cout)).astype(np.float32)
    shift = (0.1 * rs.randn(cout)).astype(np.float32)
    want = keras_ref.conv2d(x.double().numpy(), wt.double().numpy(), None, stride, padding, dtype=torch.float64)
    want = want * torch.from_numpy(scale).double() + torch.from_numpy(shift).double()
    res = bf(rs.randn(*want.shape).astype(np.float32))
    want = (want + res.double()).clamp(min=0)
    pc = ops.PackedConvBf16(wt.float(), scale, shift)
    got = ops.conv2d_bf16(x.cuda(), pc, stride, padding, "relu", res.cuda(), tile=tile)
    assert got.dtype == torch.bfloat16 and tuple(got.shape) == tuple(want.shape)
    err = ((got.cpu().double() - want).abs() / want.abs().clamp(min=1.0)).max().item()
    assert err < 1e-2, err
    got32 = ops.conv2d_bf16(x.cuda(), pc, stride, padding, "relu", res.cuda(), out_f32=True, tile=tile)     # (60 with an f32 output: the tiled form)
    err32 = ((got32.cpu().double() - want).abs() / want.abs().clamp(min=1.0)).max().item()
    assert err32 < 1e-4, err32                      # f32 accumulate on identical operands
    if tile in (45, 46, 47, 48, 49, 60):            # same k order, same MFMA: bit for bit the 128x128 tile's result,
        ref42 = ops.conv2d_bf16(x.cuda(), pc, stride, padding, "relu", res.cuda(), tile=42)
        for _ in range(8):                          # every time (the staging is asynchronous: a race would come and go)
            assert torch.equal(ops.conv2d_bf16(x.cuda(), pc, stride, padding, "relu", res.cuda(), tile=tile), ref42)


def test_resnet101_bf16_network():
    from faster_rcnn_amd import resnet
    from faster_rcnn_amd.weights import synthetic_resnet
    from oracle.keras_ref import KerasGraphs
    A, C = 18, 10
    w = synthetic_resnet(101, anchors_per_loc=A, num_classes=C, seed=3)
    base = resnet.resnet101_base(weights=w, dtype="bf16")
    rpn = resnet.resnet101_rpn(base, include_conv=True, anchors_per_loc=A)
    rs = np.random.RandomState(0)
    x = (rs.randint(0, 256, (131, 176, 3)).astype(np.float64) - np.array([103.939, 116.779, 123.68]))[None]
    cls, reg, feat = rpn.predict_on_batch(x)
    ref = KerasGraphs(w, torch.float64)
    f64 = ref.resnet_base(x, 101)
    c64, r64 = ref.rpn(f64)
    def err(a, b):
        a = torch.as_tensor(np.asarray(a)).double()
        d = a - b
        return float(d.pow(2).mean().sqrt() / b.pow(2).mean().sqrt()), float(d.abs().max() / b.abs().max())

    def ok(a, b):
        rms, mx = err(a, b)
        return rms < 2e-2 and mx < 5e-2
    assert cls.dtype == np.float32 and ok(feat, f64) and ok(cls, c64) and ok(reg, r64), (err(feat, f64), err(cls, c64), err(reg, r64))
    rows, cols = feat.shape[1:3]
    rois = np.array([[0, 0, cols - 1, rows - 1], [1, 1, 4, 5], [2, 0, 9, 3], [3, 2, 4, 3]], dtype=np.float32)
    det = resnet.resnet101_classifier(len(rois), C, weights=w, dtype="bf16")
    fmap = torch.from_numpy(f64.float().numpy()).to(torch.bfloat16).cuda()
    out_cls, out_reg = det.forward_dev(fmap, torch.from_numpy(rois).cuda())
    k64, g64 = ref.resnet_classifier(f64.float(), rois, C, 101)
    assert ok(out_cls.cpu().numpy(), k64) and ok(out_reg.cpu().numpy(), g64), (err(out_cls.cpu().numpy(), k64), err(out_reg.cpu().numpy(), g64))


# ----------------------------------------------------------------------------- mixed-precision training pieces
def _bf(a):
    return torch.from_numpy(np.asarray(a, dtype=np.float32)).to(torch.bfloat16)


@pytest.mark.parametrize("case", [(1, 13, 17, 64, 128, 3, "same"), (2, 7, 7, 128, 256, 3, "same"), (1, 19, 23, 256, 64, 1, "valid"),
                                  (1, 38, 63, 1024, 256, 1, "valid"), (3, 9, 11, 64, 64, 3, "same"), (1, 61, 47, 192, 320, 1, "valid")])
def test_bf16_backward_kernels(case):
    """Input gradient (bf16 MFMA conv on the transposed, flipped, scale-folded filter with fused ReLU mask and
    shortcut gradient) and weight / bias gradient (bf16 in, f32 out) against torch autograd in f64 on the SAME
    bf16-rounded tensors: only the f32 accumulation and the bf16 output rounding separate the two."""
    import torch.nn.functional as F
    from faster_rcnn_amd import ops
    from oracle import keras_ref
    n, h, w, cin, cout, k, padding = case
    rs = np.random.RandomState(abs(hash(case)) % (2 ** 31))
    x = _bf(rs.randn(n, h, w, cin))
    wt = (rs.randn(k, k, cin, cout) * np.sqrt(2.0 / (k * k * cin))).astype(np.float32)
    scale = (1 + 0.1 * rs.randn(cout)).astype(np.float32)
    g = _bf(rs.randn(n, h, w, cout))
    res = _bf(rs.randn(n, h, w, cin))
    mask = _bf(rs.randn(n, h, w, cin))
    xt = x.double().requires_grad_(True)
    wtt = torch.from_numpy(wt).double().requires_grad_(True)
    y = keras_ref.conv2d(xt, wtt, None, 1, padding, dtype=torch.float64) * torch.from_numpy(scale).double()
    y.backward(g.double())
    # weight / bias gradient
    dw, db = ops.conv2d_wgrad_bf16(x.cuda(), g.cuda(), k, k, 1, padding, scale=torch.from_numpy(scale).cuda())
    assert dw.dtype == torch.float32
    assert ((dw.cpu().double() - wtt.grad).abs().max() / wtt.grad.abs().max()).item() < 1e-4
    want_db = g.double().sum((0, 1, 2)) * torch.from_numpy(scale).double()
    assert ((db.cpu().double() - want_db).abs().max() / want_db.abs().max()).item() < 1e-4
    # input gradient: the packed filter is bf16(w * scale), so compare against THAT filter's exact gradient
    pd = ops.PackedDgradBf16(wt, scale)
    wq = _bf(wt * scale).double()                           # bf16(w[ci][co] * scale[co]), what the pack stores
    xt2 = x.double().requires_grad_(True)
    keras_ref.conv2d(xt2, wq, None, 1, padding, dtype=torch.float64).backward(g.double())
    want = (xt2.grad + res.double()) * (mask.double() > 0)
    got = ops.conv2d_dgrad_bf16(g.cuda(), pd, padding, residual=res.cuda(), mask=mask.cuda())
    assert got.dtype == torch.bfloat16
    err = ((got.cpu().double() - want).abs() / want.abs().clamp(min=1.0)).max().item()
    assert err < 1e-2, err                                  # one bf16 rounding of the output


def test_refresh_packed_bf16_matches_single_pack():
    import ctypes
    from faster_rcnn_amd import _lib, ops
    rs = np.random.RandomState(9)
    keep, jobs = [], []
    for kh, kw, cin, cout in [(3, 3, 64, 128), (1, 1, 256, 64), (3, 3, 128, 192), (1, 1, 1024, 256)] * 9:      # 36 jobs: two launches
        w = torch.from_numpy(rs.randn(kh, kw, cin, cout).astype(np.float32)).cuda()
        bias, scale, const = (torch.from_numpy(rs.randn(cout).astype(np.float32)).cuda() for _ in range(3))
        packed = torch.zeros((cout, kh * kw * cin), dtype=torch.bfloat16, device="cuda")
        pdg = torch.zeros((cin, kh * kw * cout), dtype=torch.bfloat16, device="cuda")
        shift = torch.zeros(cout, device="cuda")
        keep.append((w, bias, scale, const, packed, pdg, shift))
        jobs.append(_lib.PackJob(w_hwio=w.data_ptr(), packed=packed.data_ptr(), packed_dgrad=pdg.data_ptr(), bias=bias.data_ptr(),
                                 scale=scale.data_ptr(), shift_const=const.data_ptr(), shift=shift.data_ptr(), kh=kh, kw=kw, cin=cin, cout=cout))
    arr = (_lib.PackJob * len(jobs))(*jobs)
    _lib.call("frcnn_refresh_packed_bf16", arr, len(jobs), ops._stream())
    for w, bias, scale, const, packed, pdg, shift in keep:
        kh, kw, cin, cout = w.shape
        assert torch.equal(packed, ops.PackedConvBf16(w, scale, None).w)
        # dgrad form: rows = cin, k = ((co // 64) * RS + tap') * 64 + co % 64, value bf16(w[flip(tap')][ci][co] * scale[co])
        ws = (w * scale).flip(0, 1)                                         # [r'][s'][ci][co]
        ref = ws.permute(2, 0, 1, 3).reshape(cin, kh * kw, cout // 64, 64).permute(0, 2, 1, 3).reshape(cin, -1).to(torch.bfloat16)
        assert torch.equal(pdg, ref)
        assert torch.allclose(shift, bias * scale + const, rtol=1e-6, atol=1e-6)


def _cos(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a @ b) / (a.norm() * b.norm() + 1e-30))


@pytest.mark.parametrize("which", ["det", "rpn"])
def test_mixed_precision_training_step_tracks_f32(which):
    """BASELINE configs[4] "mixed bf16": bf16 activations / gradients / packed filters with f32 master weights,
    weight gradients and optimiser.  One SGD step from identical weights and inputs against the f32 trainer: the
    losses agree to bf16 accuracy and every trained tensor moves in the same direction."""
    from faster_rcnn_amd import resnet, train
    from faster_rcnn_amd.weights import synthetic_resnet
    A, C, H, W = 9, 21, 192, 256
    rs = np.random.RandomState(4)
    x = (rs.randint(0, 256, (H, W, 3)).astype(np.float64) - np.array([103.939, 116.779, 123.68]))[None]
    rows, cols = resnet.get_conv_rows_cols(H, W)
    results = {}
    for dt in ("f32", "bf16"):
        w = synthetic_resnet(50, anchors_per_loc=A, num_classes=C, seed=7)
        base = resnet.resnet50_base(weights=w, weight_regularizer=resnet.WEIGHT_REGULARIZER, bias_regularizer=resnet.BIAS_REGULARIZER, dtype=dt)
        r2 = np.random.RandomState(5)
        if which == "rpn":
            m = resnet.resnet50_rpn(base, anchors_per_loc=A)
            can_use = r2.rand(1, rows, cols, A) < 0.3; is_pos = r2.rand(1, rows, cols, A) < 0.1
            y = [np.concatenate([can_use, is_pos], axis=3),
                 np.concatenate([np.repeat(can_use & is_pos, 4, axis=3).astype(np.float32), (r2.randn(1, rows, cols, 4 * A) * is_pos.repeat(4, axis=3)).astype(np.float32)], axis=3)]
            xin = x
        else:
            m = resnet.resnet50_classifier(16, C, base)
            n = 16
            x1 = r2.randint(0, cols - 6, n); y1 = r2.randint(0, rows - 6, n)
            rois = np.stack([x1, y1, x1 + 1 + r2.randint(0, 5, n), y1 + 1 + r2.randint(0, 5, n)], axis=1).astype(np.float32)[None]
            ci = r2.randint(0, C, n)
            yc = np.zeros((1, n, C), np.float32); yc[0, np.arange(n), ci] = 1
            lab = np.zeros((n, 4 * (C - 1)), np.float32); tg = np.zeros((n, 4 * (C - 1)), np.float32)
            for i, c in enumerate(ci):
                if c < C - 1:
                    lab[i, 4 * c:4 * c + 4] = 1; tg[i, 4 * c:4 * c + 4] = r2.randn(4)
            y = [yc, np.concatenate([lab, tg], axis=1)[None]]
            xin = [x, rois]
        m.compile(train.SGD(1e-3, 0.9))
        tr = m._trainer
        before = tr.params.w.clone()
        losses = m.train_on_batch(xin, y)
        results[dt] = (losses, (tr.params.w - before).cpu(), tr.params)
    (l32, d32, p32), (l16, d16, _) = results["f32"], results["bf16"]
    for a, b in zip(l32, l16):
        assert abs(a - b) <= 3e-2 * max(1.0, abs(a)), (l32, l16)
    assert _cos(d32, d16) > 0.97                                # the whole update vector
    off = 0
    worst = 1.0
    for name in p32.names:                                      # and every large tensor on its own
        for wv, _ in p32.views[name]:
            k = wv.numel()
            if k >= 4096 and float(d32[off:off + k].norm()) > 0:
                worst = min(worst, _cos(d32[off:off + k], d16[off:off + k]))
            off += k
    assert worst > 0.9, worst


@pytest.mark.parametrize("case", [(2, 15, 18, 160, 136, 3, 2, "same"), (1, 37, 50, 256, 128, 1, 2, "valid"), (3, 7, 7, 192, 136, 3, 1, "same")])
def test_wgrad_bf16_big_tile_strided_and_ragged(case):
    """The 128x128 bf16 weight-gradient tile (cin, cout >= 128, multiples of 8): stride 2 with the asymmetric SAME halo, a strided
    1x1, ragged channel tiles and rows wrapping over images -- against autograd in f64 on the same bf16-rounded tensors; twice the
    same bits."""
    from faster_rcnn_amd import ops
    from oracle import keras_ref
    n, h, w, cin, cout, k, stride, padding = case
    rs = np.random.RandomState(abs(hash(case)) % (2 ** 31))
    x = _bf(rs.randn(n, h, w, cin))
    wtt = torch.zeros(k, k, cin, cout, dtype=torch.float64, requires_grad=True)
    y = keras_ref.conv2d(x.double(), wtt, None, stride, padding, dtype=torch.float64)
    g = _bf(rs.randn(*y.shape))
    y.backward(g.double())
    dw, _ = ops.conv2d_wgrad_bf16(x.cuda(), g.cuda(), k, k, stride, padding, want_bias=False)
    dw2, _ = ops.conv2d_wgrad_bf16(x.cuda(), g.cuda(), k, k, stride, padding, want_bias=False)
    assert torch.equal(dw, dw2)
    assert ((dw.cpu().double() - wtt.grad).abs().max() / wtt.grad.abs().max()).item() < 1e-4


def test_wgrad_bf16_widening_fallback():
    """Channel counts that are not multiples of 8 cannot use the 16-byte bf16 staging: the kernel that widens to f32
    while staging takes over (same contract)."""
    from faster_rcnn_amd import ops
    from oracle import keras_ref
    rs = np.random.RandomState(12)
    x, g = _bf(rs.randn(1, 19, 23, 64)), _bf(rs.randn(1, 19, 23, 36))
    xt = x.double()
    wtt = torch.zeros(1, 1, 64, 36, dtype=torch.float64, requires_grad=True)
    keras_ref.conv2d(xt, wtt, None, 1, "valid", dtype=torch.float64).backward(g.double())
    dw, db = ops.conv2d_wgrad_bf16(x.cuda(), g.cuda(), 1, 1, 1, "valid")
    assert ((dw.cpu().double() - wtt.grad).abs().max() / wtt.grad.abs().max()).item() < 1e-4
    assert ((db.cpu().double() - g.double().sum((0, 1, 2))).abs().max()).item() < 1e-3


@pytest.mark.parametrize("which", ["det", "rpn"])
def test_mixed_precision_training_step_against_the_bf16_storage_oracle(which):
    """The mixed-precision step against the ORACLE, not against the product's own f32 trainer: keras_train_ref evaluated
    under the bf16 storage model (oracle/keras_ref.py ``mixed=True``: f64 arithmetic, one bf16 rounding wherever the product
    stores an activation, an activation gradient or a packed filter in bf16).  Storage precision is then common to both
    sides; what separates them is f32-vs-f64 accumulation and the bf16 ulps that flips.  Bars: the three losses within
    2e-3 relative; per trained tensor the SGD update's cosine >= 0.999 and relative Frobenius error <= 0.05 (measured worst:
    0.9996 / 0.029, res4a_branch2a); the whole update vector cosine >= 0.9995 (the f32-trainer comparison above only
    manages 0.97: that gap IS the storage precision)."""
    from faster_rcnn_amd import resnet, train
    from faster_rcnn_amd.weights import synthetic_resnet
    from oracle import keras_train_ref as kt
    A, C, H, W = 9, 21, 192, 256
    rs = np.random.RandomState(4)
    x = (rs.randint(0, 256, (H, W, 3)).astype(np.float64) - np.array([103.939, 116.779, 123.68]))[None]
    rows, cols = resnet.get_conv_rows_cols(H, W)
    w0 = synthetic_resnet(50, anchors_per_loc=A, num_classes=C, seed=7)
    base = resnet.resnet50_base(weights={k: [a.copy() for a in v] for k, v in w0.items()}, dtype="bf16",
                                weight_regularizer=resnet.WEIGHT_REGULARIZER, bias_regularizer=resnet.BIAS_REGULARIZER)
    r2 = np.random.RandomState(5)
    if which == "rpn":
        m = resnet.resnet50_rpn(base, anchors_per_loc=A)
        can_use = r2.rand(1, rows, cols, A) < 0.3; is_pos = r2.rand(1, rows, cols, A) < 0.1
        y = [np.concatenate([can_use, is_pos], axis=3),
             np.concatenate([np.repeat(can_use & is_pos, 4, axis=3).astype(np.float32), (r2.randn(1, rows, cols, 4 * A) * is_pos.repeat(4, axis=3)).astype(np.float32)], axis=3)]
        m.compile(train.SGD(1e-3, 0.9))
        losses = m.train_on_batch(x, y)
        ref_w, ref_losses, _ = kt.rpn_train_step(w0, x, y[0], y[1], A, kt.Optim("sgd", 1e-3), l2=1e-4, mixed=True)
        names = kt.conv_layer_names(50, [4]) + ["rpn_conv1", "rpn_out_cls", "rpn_out_bbreg"]
    else:
        n = 16
        m = resnet.resnet50_classifier(n, C, base)
        x1 = r2.randint(0, cols - 6, n); y1 = r2.randint(0, rows - 6, n)
        rois = np.stack([x1, y1, x1 + 1 + r2.randint(0, 5, n), y1 + 1 + r2.randint(0, 5, n)], axis=1).astype(np.float32)[None]
        ci = r2.randint(0, C, n)
        yc = np.zeros((1, n, C), np.float32); yc[0, np.arange(n), ci] = 1
        lab = np.zeros((n, 4 * (C - 1)), np.float32); tg = np.zeros((n, 4 * (C - 1)), np.float32)
        for i, c in enumerate(ci):
            if c < C - 1:
                lab[i, 4 * c:4 * c + 4] = 1; tg[i, 4 * c:4 * c + 4] = r2.randn(4)
        y = [yc, np.concatenate([lab, tg], axis=1)[None]]
        m.compile(train.SGD(1e-3, 0.9))
        losses = m.train_on_batch([x, rois], y)
        ref_w, ref_losses, _ = kt.det_train_step(w0, x, rois, y[0], y[1], C, kt.Optim("sgd", 1e-3), l2=1e-4, mixed=True)
        names = kt.conv_layer_names(50, [4, 5]) + ["dense_class_%d" % C, "dense_reg_%d" % C]
    for a, b in zip(losses, ref_losses):
        assert abs(a - b) <= 2e-3 * max(1.0, abs(b)), (losses, ref_losses)
    tot_dot = tot_g = tot_w = 0.0
    report = []
    for name in names:
        o = np.asarray(w0[name][0], np.float64)
        dg = m.get_layer(name).get_weights()[0].astype(np.float64) - o
        dw = np.asarray(ref_w[name][0], np.float64) - o
        cos = float((dg * dw).sum() / (np.linalg.norm(dg) * np.linalg.norm(dw)))
        fro = float(np.linalg.norm(dg - dw) / np.linalg.norm(dw))
        report.append((name, round(cos, 5), round(fro, 4)))
        tot_dot += (dg * dw).sum(); tot_g += (dg * dg).sum(); tot_w += (dw * dw).sum()
    print("mixed vs bf16-storage oracle (%s): losses %s / %s; worst %s" % (which, losses, ref_losses, min(report, key=lambda r: r[1])))
    assert min(r[1] for r in report) >= 0.999, min(report, key=lambda r: r[1])
    assert max(r[2] for r in report) <= 0.05, max(report, key=lambda r: r[2])
    assert tot_dot / np.sqrt(tot_g * tot_w) >= 0.9995


def test_batched_pipeline_equals_per_image_pipeline():
    """BatchedInferencePipeline (B images per pass: trunk and RPN heads at batch B, ONE detector-head pass over all B x n RoIs)
    against InferencePipeline on each image alone, both without split-K: a row's k order does not depend on how tall the GEMM
    is, so every output is bit-identical -- RPN maps, proposals, detector outputs, detections."""
    import numpy as np
    from faster_rcnn_amd import ops, resnet, util
    from faster_rcnn_amd.pipeline import BatchedInferencePipeline, InferencePipeline
    from faster_rcnn_amd.weights import synthetic_resnet
    anchors = util.get_anchors([32, 64, 128])
    A, C, B, n = len(anchors), 10, 3, 40
    w = synthetic_resnet(50, anchors_per_loc=A, num_classes=C, seed=9)
    base = resnet.resnet50_base(weights=w, dtype="bf16")
    rpn = resnet.resnet50_rpn(base, include_conv=True, anchors_per_loc=A)
    det = resnet.resnet50_classifier(n, C, weights=w, dtype="bf16")
    rs = np.random.RandomState(5)
    x = torch.from_numpy((rs.randint(0, 256, (B, 176, 240, 3)) - 110.0).astype(np.float32)).cuda()
    with ops.conv_workspace(ops.NO_SPLIT_K):
        bp = BatchedInferencePipeline(rpn, det, anchors, B, max_proposals=n)
        out = bp.forward_dev(x)
        single = InferencePipeline(rpn, det, anchors, max_proposals=n)
        per = [single.forward_dev(x[i:i + 1].contiguous()) for i in range(B)]
    torch.cuda.synchronize()
    for i, o in enumerate(per):
        assert torch.equal(out["rpn_cls"][i], o["rpn_cls"][0]) and torch.equal(out["rpn_reg"][i], o["rpn_reg"][0]) and torch.equal(out["feat"][i], o["feat"][0])
        assert int(out["n_rois"][i]) == int(o["n_rois"]) > 0 and torch.equal(out["rois"][i], o["rois"])
        assert torch.equal(out["cls"][i], o["cls"]) and torch.equal(out["reg"][i], o["reg"])
        assert int(out["n_dets"][i]) == int(o["n_dets"])
        for k in ("det_bbox", "det_cls", "det_prob", "det_roi"):                     # (det_packed also holds three pad words)
            assert torch.equal(out[k][i], o[k]), k
    # a "batch" of one is the per-image pipeline too
    with ops.conv_workspace(ops.NO_SPLIT_K):
        one = BatchedInferencePipeline(rpn, det, anchors, 1, max_proposals=n).forward_dev(x[1:2].contiguous())
    torch.cuda.synchronize()
    assert torch.equal(one["cls"][0], per[1]["cls"]) and torch.equal(one["det_bbox"][0], per[1]["det_bbox"]) and int(one["n_dets"][0]) == int(per[1]["n_dets"])
    # and from a captured graph (what bench.py --config c4 replays), fed another batch
    bp.capture(176, 240)
    x2 = torch.from_numpy((rs.randint(0, 256, (B, 176, 240, 3)) - 110.0).astype(np.float32)).cuda()
    bp._static_in.copy_(x2)
    bp._graph.replay()
    torch.cuda.synchronize()
    with ops.conv_workspace(ops.NO_SPLIT_K):
        want = bp.forward_dev(x2)
    torch.cuda.synchronize()
    for i in range(B):
        assert torch.equal(bp._static_out["det_bbox"][i], want["det_bbox"][i]) and torch.equal(bp._static_out["det_prob"][i], want["det_prob"][i])
        assert torch.equal(bp._static_out["cls"][i], want["cls"][i])


def test_batched_roi_resize_equals_per_image():
    import numpy as np
    from faster_rcnn_amd import ops
    rs = np.random.RandomState(2)
    B, R, Cc, Cf, n = 3, 9, 13, 64, 7
    feat = torch.from_numpy(rs.randn(B, R, Cc, Cf).astype(np.float32)).cuda().to(torch.bfloat16)
    x1, y1 = rs.randint(0, Cc - 2, B * n), rs.randint(0, R - 2, B * n)
    rois = np.stack([x1, y1, x1 + 1 + rs.randint(0, 6, B * n), y1 + 1 + rs.randint(0, 5, B * n)], 1).astype(np.float32)
    rois[4] = [3, 3, 3, 8]                                             # an empty RoI -> the fill vector
    rois[9] = [2, 1, Cc + 2, 4]                                        # out of the map -> fill
    rois_d = torch.from_numpy(rois).cuda()
    fill = torch.from_numpy(rs.randn(Cf).astype(np.float32)).cuda()
    for layout in (0, 1):
        for relu in (False, True):
            got = ops.roi_crop_resize_bf16_batch(feat, rois_d, n, 7, fill=fill, relu=relu, layout=layout)
            for i in range(B):
                want = ops.roi_crop_resize_bf16(feat[i], rois_d[i * n:(i + 1) * n], 7, fill=fill, relu=relu, layout=layout)
                part = got[:, :, i * n:(i + 1) * n] if layout else got[i * n:(i + 1) * n]
                assert torch.equal(part, want)


@pytest.mark.parametrize("shape", [(1, 37, 53), (2, 160, 224), (1, 75, 131), (1, 600, 1500)])
def test_fused_bf16_stem(shape):
    """frcnn_stem_bf16_fwd (conv1 7x7/2 'same' + BN + Scale + ReLU + 3x3/2 max-pool + bf16 store in one launch, resnet.py:408-412)
    against f64 arithmetic on the SAME bf16-rounded pixels and filter taps: what is left is the f32 accumulation order, i.e. at
    most one bf16 rounding step on the stored value.  Sizes: odd and even, pooled extents that do not fill the last 4 x 16
    workgroup patch, a batch of two, and configs[3]'s own 600x1500."""
    import numpy as np
    import torch.nn.functional as F
    from faster_rcnn_amd import nets, ops
    n, h, w = shape
    rs = np.random.RandomState(h)
    x = (rs.randint(0, 256, (n, h, w, 3)) - np.array([103.939, 116.779, 123.68])).astype(np.float32)
    wts = {"conv1": [(rs.randn(7, 7, 3, 64) * 0.05).astype(np.float32), (rs.randn(64) * 0.1).astype(np.float32)],
           "bn_conv1": [(rs.rand(64) + 0.5).astype(np.float32), (rs.randn(64) * 0.2).astype(np.float32), (rs.randn(64) * 0.5).astype(np.float32), (rs.rand(64) + 0.5).astype(np.float32)],
           "scale_conv1": [(rs.rand(64) + 0.5).astype(np.float32), (rs.randn(64) * 0.1).astype(np.float32)]}
    unit = nets.ConvUnit(wts, "conv1", "bn_conv1", "scale_conv1", nets.BN_EPS_STEM, stride=2, padding="same", act="relu")
    kernel, scale, shift = unit.folded()
    got = ops.stem_bf16(torch.from_numpy(x).cuda(), ops.PackedStemBf16(kernel, scale, shift)).float().cpu()
    ho, wo = (h + 1) // 2, (w + 1) // 2
    assert tuple(got.shape) == (n, (ho - 3) // 2 + 1, (wo - 3) // 2 + 1, 64)
    q = lambda t: t.to(torch.bfloat16).to(torch.float64)
    xq, kq = q(torch.from_numpy(x)).permute(0, 3, 1, 2), q(torch.from_numpy(kernel)).permute(3, 2, 0, 1)
    pt, pl = max((ho - 1) * 2 + 7 - h, 0), max((wo - 1) * 2 + 7 - w, 0)
    y = F.conv2d(F.pad(xq, (pl // 2, pl - pl // 2, pt // 2, pt - pt // 2)), kq, stride=2)
    y = (y * torch.from_numpy(scale).double().view(1, -1, 1, 1) + torch.from_numpy(shift).double().view(1, -1, 1, 1)).clamp(min=0)
    want = F.max_pool2d(y, 3, 2).permute(0, 2, 3, 1)
    err = (got.double() - want).abs()
    bar = want.abs() * 2.0 ** -8 + 1e-6                       # one bf16 rounding step (8 significant bits) of the exact value
    assert bool((err <= bar).all()), float((err / bar).max())
    assert float((got.double() == want.to(torch.bfloat16).double()).double().mean()) > 0.97     # nearly all values: the very same bf16


def test_fused_stem_network_close_to_f32_stem_network():
    """The bf16 trunk behind the fused bf16 stem against the same trunk behind the round-2 f32 stem (FUSED_BF16_STEM off): the two
    differ by the stem's operand rounding only -- relative RMS of the conv4 map well under the bf16 parity bars."""
    import numpy as np
    from faster_rcnn_amd import nets, resnet
    from faster_rcnn_amd.weights import synthetic_resnet
    w = synthetic_resnet(50, anchors_per_loc=9, seed=3)
    x = torch.from_numpy((np.random.RandomState(0).randint(0, 256, (1, 160, 224, 3)) - 110.0).astype(np.float32)).cuda()
    outs = []
    for fused in (True, False):
        nets.FUSED_BF16_STEM = fused
        try:
            outs.append(resnet.resnet50_base(weights=w, dtype="bf16").net(x).float())
        finally:
            nets.FUSED_BF16_STEM = True
    a, b = outs
    rms = float(((a - b) ** 2).mean().sqrt() / (b ** 2).mean().sqrt())
    assert rms < 1e-2, rms


def test_fused_bf16_stem_random_sizes():
    """The fused stem on random image sizes (odd / even, down to the smallest the pool accepts, pooled extents on and off the
    4 x 16 workgroup patch) against the three launches it replaces fed the SAME bf16-rounded pixels and taps: f32 conv of
    bf16-exact operands accumulates the same products in another order, so the stored bf16 values agree to one rounding step."""
    import numpy as np
    from faster_rcnn_amd import ops
    rs = np.random.RandomState(77)
    wt = (rs.randn(7, 7, 3, 64) * 0.05).astype(np.float32)
    sc, sh = (rs.rand(64) + 0.5).astype(np.float32), rs.randn(64).astype(np.float32)
    q = lambda a: torch.from_numpy(a).to(torch.bfloat16).float().numpy()
    ps, pc = ops.PackedStemBf16(wt, sc, sh), ops.PackedConv(q(wt), sc, sh)
    for it in range(12):
        n, h, w = int(rs.randint(1, 3)), int(rs.randint(11, 140)), int(rs.randint(11, 200))
        x = (rs.randint(0, 256, (n, h, w, 3)) - 110.0).astype(np.float32)
        got = ops.stem_bf16(torch.from_numpy(x).cuda(), ps).float()
        ref = ops.cast_bf16(ops.pool2d(ops.conv2d(torch.from_numpy(q(x)).cuda(), pc, 2, "same", "relu"), 3, 2, True)).float()
        assert got.shape == ref.shape, (n, h, w)
        err = (got - ref).abs()
        assert bool((err <= ref.abs() * 2.0 ** -7 + 1e-6).all()), ((n, h, w), float(err.max()))
        assert float((got == ref).float().mean()) > 0.95, (n, h, w)
