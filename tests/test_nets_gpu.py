"""GPU parity of the lowered networks vs the torch-CPU restatement of the Keras graphs.
Tolerance (north_star): fp32 class scores / box regressions within 1e-4 of the reference
path; measured against the float64 evaluation of the same graph, scaled by max(1,|want|)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

TOL = 1e-4


def rel_err(got, want):
    got = torch.as_tensor(np.asarray(got)).double()
    want = torch.as_tensor(np.asarray(want)).double()
    return ((got - want).abs() / want.abs().clamp(min=1.0)).max().item()


def image(h, w, seed=0):
    rs = np.random.RandomState(seed)
    return (rs.randint(0, 256, (h, w, 3)).astype(np.float64) - np.array([103.939, 116.779, 123.68]))[None]


@pytest.mark.parametrize("depth", [50, 101])
def test_resnet_rpn_and_head(depth):
    from faster_rcnn_amd import resnet
    from faster_rcnn_amd.weights import synthetic_resnet
    from oracle.keras_ref import KerasGraphs
    C = 21 if depth == 50 else 10
    A = 9
    w = synthetic_resnet(depth, anchors_per_loc=A, num_classes=C, seed=3)
    base = (resnet.resnet50_base if depth == 50 else resnet.resnet101_base)(weights=w)
    rpn = resnet.resnet50_rpn(base, include_conv=True, anchors_per_loc=A)
    x = image(131, 176)
    cls, reg, feat = rpn.predict_on_batch(x)
    rows, cols = resnet.get_conv_rows_cols(131, 176)
    assert cls.shape == (1, rows, cols, A) and reg.shape == (1, rows, cols, 4 * A) and feat.shape == (1, rows, cols, 1024)
    ref = KerasGraphs(w, torch.float64)
    f64 = ref.resnet_base(x, depth)
    c64, r64 = ref.rpn(f64)
    assert rel_err(feat, f64) < TOL, rel_err(feat, f64)
    assert rel_err(cls, c64) < TOL and rel_err(reg, r64) < TOL
    # detector head on a handful of RoIs, fed with the ORACLE's feature map so errors don't compound
    rois = np.array([[0, 0, cols - 1, rows - 1], [1, 1, 4, 5], [2, 0, 9, 3], [3, 2, 4, 3], [0, 3, 10, 8]], dtype=np.float32)
    det = (resnet.resnet50_classifier if depth == 50 else resnet.resnet101_classifier)(len(rois), C, weights=w)
    f32map = f64.float().numpy()
    out_cls, out_reg = det.predict([f32map, rois[None]])
    k64, g64 = ref.resnet_classifier(torch.from_numpy(f32map), rois, C, depth)
    assert out_cls.shape == (1, len(rois), C) and out_reg.shape == (1, len(rois), 4 * (C - 1))
    assert rel_err(out_cls[0], k64) < TOL and rel_err(out_reg[0], g64) < TOL
    assert abs(out_cls[0].sum(axis=1) - 1).max() < 1e-5


def test_vgg_rpn_and_head():
    from faster_rcnn_amd import vgg
    from faster_rcnn_amd.weights import synthetic_vgg16
    from oracle.keras_ref import KerasGraphs
    w = synthetic_vgg16(seed=4)
    base = vgg.vgg16_base(weights=w)
    rpn = vgg.vgg16_rpn(base, include_conv=True, anchors_per_loc=9)
    x = image(97, 130, seed=1)
    cls, reg, feat = rpn.predict_on_batch(x)
    assert feat.shape == (1, 97 // 16, 130 // 16, 512)
    ref = KerasGraphs(w, torch.float64)
    f64 = ref.vgg_base(x)
    c64, r64 = ref.rpn(f64)
    assert rel_err(feat, f64) < TOL and rel_err(cls, c64) < TOL and rel_err(reg, r64) < TOL
    rois = np.array([[0, 0, 7, 5], [1, 1, 4, 5], [2, 0, 6, 3]], dtype=np.float32)
    det = vgg.vgg16_classifier(len(rois), 21, weights=w)
    f32map = f64.float().numpy()
    out_cls, out_reg = det.predict([f32map, rois[None]])
    k64, g64 = ref.vgg_classifier(torch.from_numpy(f32map), rois, 21)
    assert rel_err(out_cls[0], k64) < TOL and rel_err(out_reg[0], g64) < TOL


def test_fp32_oracle_is_inside_tolerance_too():
    """the float32 evaluation of the restatement ("Keras CPU path") and the float64 one agree
    to the same tolerance, so 1e-4 vs f64 is a meaningful bound for 'matches the CPU path'."""
    from faster_rcnn_amd.weights import synthetic_resnet
    from oracle.keras_ref import KerasGraphs
    w = synthetic_resnet(50, seed=3)
    x = image(67, 83)
    a = KerasGraphs(w, torch.float32).resnet_base(x)
    b = KerasGraphs(w, torch.float64).resnet_base(x)
    assert rel_err(a, b) < TOL


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_head_hoist_equals_reference_order(dtype):
    """res5a_branch2a / branch1 applied to the conv4 map and resampled (nets.ResNetHead hoist) against the
    reference order (resample, then convolve every crop): same class scores / regressions, including RoIs
    the crop rejects (empty or out of range -> zeros in, BatchNorm shift out)."""
    from faster_rcnn_amd import nets
    from faster_rcnn_amd.weights import synthetic_resnet
    from oracle.keras_ref import KerasGraphs
    C = 21
    w = synthetic_resnet(50, anchors_per_loc=9, num_classes=C, seed=5)
    rs = np.random.RandomState(2)
    rows, cols = 38, 63
    feat = np.maximum(rs.randn(1, rows, cols, 1024), 0).astype(np.float32)
    x1 = rs.randint(0, cols - 8, 40); y1 = rs.randint(0, rows - 8, 40)
    rois = np.stack([x1, y1, x1 + 1 + rs.randint(0, 7, 40), y1 + 1 + rs.randint(0, 7, 40)], axis=1).astype(np.float32)
    rois[5] = [0, 0, 0, 0]                                  # empty: what the padded tail of the RoI buffer holds
    rois[6] = [10, 10, 9, 12]                               # negative width
    rois[7] = [60, 30, 70, 36]                              # reaches past the map
    rois[8] = [0, 0, cols, rows]                            # the whole map
    fd = torch.from_numpy(feat).cuda()
    rd = torch.from_numpy(rois).cuda()
    if dtype == "bf16":
        fd = fd.to(torch.bfloat16)
    a = nets.ResNetHead(w, 50, C, dtype=dtype, hoist=True)(fd, rd)
    b = nets.ResNetHead(w, 50, C, dtype=dtype, hoist=False, pos_major=False)(fd, rd)
    tol = 2e-5 if dtype == "f32" else 2e-2                  # bf16: the two orders round different tensors to bf16
    assert rel_err(a[0].cpu(), b[0].cpu()) < tol and rel_err(a[1].cpu(), b[1].cpu()) < tol
    # the position-major layout alone (tap skipping) must not change a single bit (plain launches: split-K cuts
    # the compacted chunk sequence at other places, which regroups the partial sums)
    from faster_rcnn_amd import ops
    with ops.conv_workspace(ops.NO_SPLIT_K):
        b2 = nets.ResNetHead(w, 50, C, dtype=dtype, hoist=False, pos_major=False)(fd, rd)
        c = nets.ResNetHead(w, 50, C, dtype=dtype, hoist=False, pos_major=True)(fd, rd)
    assert torch.equal(c[0], b2[0]) and torch.equal(c[1], b2[1])
    if dtype == "f32":                                      # and both sit inside the oracle bar
        valid = [i for i in range(40) if i not in (5, 6, 7)]
        k64, g64 = KerasGraphs(w, torch.float64).resnet_classifier(torch.from_numpy(feat), rois[valid], C, 50)
        assert rel_err(a[0].cpu()[valid], k64) < TOL and rel_err(a[1].cpu()[valid], g64) < TOL
