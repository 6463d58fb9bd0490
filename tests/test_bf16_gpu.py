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
                                  (1, 30, 41, 256, 128, 1, 2, "valid", 43), (3, 7, 7, 512, 200, 3, 1, "same", 44)])  # 128x64 8-wave, 16-wave
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
    got32 = ops.conv2d_bf16(x.cuda(), pc, stride, padding, "relu", res.cuda(), out_f32=True, tile=tile)
    err32 = ((got32.cpu().double() - want).abs() / want.abs().clamp(min=1.0)).max().item()
    assert err32 < 1e-4, err32                      # f32 accumulate on identical operands


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
