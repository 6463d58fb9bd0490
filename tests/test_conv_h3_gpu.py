"""The f16x3 fp32 conv engine (csrc/conv_h3.hip, frcnn_conv2d_fwd_h3): fp32 convolution on the fp16 matrix cores by a two-way operand
split with a scaled low part -- three matrix instructions per block of products where the split-bf16 engine needs six.  Held to the
SAME bars as tests/test_conv_x6_gpu.py holds that engine to: error against an fp64 reference at or below the native f32 MFMA
kernel's (same thresholds, unchanged), exact results on integer operands, layouts agree, skipped padding taps change nothing, runs
are bitwise reproducible -- plus what is new here: the magnitude records the engine scales its operands with."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

from tests.test_conv_x6_gpu import err, ref_conv      # noqa: E402  (the fp64 reference and the error measure of the split-bf16 tests)


CASES = [
    # n, h, w, cin, cout, k, stride, padding, tile
    (1, 40, 52, 64, 128, 1, 1, "valid", 81),
    (1, 40, 52, 64, 128, 1, 1, "valid", 84),
    (2, 23, 31, 64, 96, 3, 1, "same", 81),        # ragged rows / columns, batch, halo
    (2, 23, 31, 64, 96, 3, 1, "same", 83),
    (1, 33, 29, 128, 192, 3, 2, "same", 82),      # stride 2 SAME (pad on one side only for odd sizes); 256x128 on eight waves
    (1, 30, 44, 256, 64, 1, 2, "valid", 84),      # strided 1x1 (conv_block shortcut)
    (3, 14, 14, 96, 256, 3, 1, "same", 82),
    (2, 23, 31, 64, 96, 3, 1, "same", 86),        # 256x128 on sixteen waves, two LDS buffers
    (1, 40, 52, 64, 320, 1, 1, "valid", 86),
    (1, 33, 29, 128, 192, 3, 2, "same", 86),
    (2, 23, 31, 64, 64, 3, 1, "same", 87),        # 128x64 on four waves: the 64-column layers
    (1, 40, 52, 96, 160, 1, 1, "valid", 87),      # ragged column tiles (160 = 2.5 x 64)
]


@pytest.mark.parametrize("case", CASES)
def test_h3_matches_fp64_as_well_as_the_native_kernel(case):
    from faster_rcnn_amd import ops
    n, h, w, cin, cout, k, stride, padding, tile = case
    rs = np.random.RandomState(2000 + CASES.index(case))
    x = rs.randn(n, h, w, cin).astype(np.float32)
    wt = (rs.randn(k, k, cin, cout) * np.sqrt(2.0 / (k * k * cin))).astype(np.float32)
    scale = (1 + 0.1 * rs.randn(cout)).astype(np.float32)
    shift = (0.1 * rs.randn(cout)).astype(np.float32)
    pc = ops.PackedConv(wt, scale, shift)
    xd = torch.from_numpy(x).cuda()
    got = ops.conv2d(xd, pc, stride, padding, "relu", tile=tile)
    nat = ops.conv2d(xd, pc, stride, padding, "relu", tile=0)
    res = rs.randn(*got.shape).astype(np.float32)
    got_r = ops.conv2d(xd, pc, stride, padding, None, residual=torch.from_numpy(res).cuda(), tile=tile)
    ref, mag = ref_conv(x, wt, stride, padding, scale, shift, None, "relu")
    ref_r, _ = ref_conv(x, wt, stride, padding, scale, shift, res, None)
    e_h3, e_nat = err(got.cpu().numpy(), ref, mag), err(nat.cpu().numpy(), ref, mag)
    print(case, "f16x3 %.3g native %.3g" % (e_h3, e_nat))
    assert got.shape == nat.shape
    assert e_h3 <= 5e-7 and e_h3 <= max(3.0 * e_nat, 4e-7)          # the split-bf16 engine's bar, unchanged
    assert err(got_r.cpu().numpy(), ref_r, mag) <= 6e-7
    # the launch left max|y| in the output's magnitude record: an upper bound that is attained
    assert float(got._amax.max()) == float(got.abs().max())
    assert float(got_r._amax.max()) == float(got_r.abs().max())
    # bitwise reproducible, and the launch does not depend on what the output buffer held
    again = ops.conv2d(xd, pc, stride, padding, "relu", tile=tile, out=torch.full_like(got, 7.0))
    assert torch.equal(again, got)


@pytest.mark.parametrize("kind", ["all_positive", "wide_exponents", "integers", "subnormal_neighbours", "same_sign_residuals", "outlier"])
def test_h3_error_on_operands_that_do_not_cancel(kind):
    """The operand classes of test_x6_error_on_operands_that_do_not_cancel, same bars, and two that aim at THIS split: operands whose
    residuals a' - ah all have the same sign and nearly the largest size (the dropped al * bl term then adds up instead of
    averaging out: 2^-22 of every product), and a post-ReLU tensor with one value 4000 times the rest (the scale follows the
    outlier; everything else sits 12 binades lower)."""
    from faster_rcnn_amd import ops
    rs = np.random.RandomState({"all_positive": 1, "wide_exponents": 2, "integers": 3, "subnormal_neighbours": 4, "same_sign_residuals": 5, "outlier": 6}[kind])
    n, h, w, cin, cout, k = 1, 30, 44, 128, 128, 3
    if kind == "all_positive":
        x = rs.uniform(0.5, 1.0, (n, h, w, cin)).astype(np.float32); wt = rs.uniform(0.5, 1.0, (k, k, cin, cout)).astype(np.float32)
    elif kind == "wide_exponents":
        x = (rs.randn(n, h, w, cin) * 2.0 ** rs.randint(-12, 13, (n, h, w, cin))).astype(np.float32)
        wt = (rs.randn(k, k, cin, cout) * 2.0 ** rs.randint(-12, 13, (k, k, cin, cout))).astype(np.float32)
    elif kind == "integers":
        x = rs.randint(-60, 61, (n, h, w, cin)).astype(np.float32); wt = rs.randint(-60, 61, (k, k, cin, cout)).astype(np.float32)
    elif kind == "subnormal_neighbours":
        x = (rs.randn(n, h, w, cin) * 1e-30).astype(np.float32); wt = (rs.randn(k, k, cin, cout) * 1e-6).astype(np.float32)
    elif kind == "same_sign_residuals":
        x = (1.0 + 2.0 ** -11 * rs.uniform(0.9, 0.99, (n, h, w, cin))).astype(np.float32)
        wt = (1.0 + 2.0 ** -11 * rs.uniform(0.9, 0.99, (k, k, cin, cout))).astype(np.float32)
    else:
        x = np.maximum(rs.randn(n, h, w, cin), 0).astype(np.float32); x[0, 7, 9, 5] = 4000.0
        wt = (rs.randn(k, k, cin, cout) * 0.03).astype(np.float32)
    pc = ops.PackedConv(wt)
    xd = torch.from_numpy(x).cuda()
    ref, mag = ref_conv(x, wt, 1, "same")
    for tile in (81, 84, 86):
        with ops.conv_workspace(ops.NO_SPLIT_K):                 # like for like: both sum k = 0 .. K - 1 in one pass
            got = ops.conv2d(xd, pc, 1, "same", None, tile=tile).cpu().numpy()
            nat = ops.conv2d(xd, pc, 1, "same", None, tile=0).cpu().numpy()
        if kind == "integers":                                   # |operand| < 2048: both fp16 pieces hold it exactly, every partial sum is an integer f32 holds
            assert np.array_equal(got.astype(np.float64), ref) and np.array_equal(nat.astype(np.float64), ref)
            continue
        if kind == "subnormal_neighbours":
            scale = 1e36
            e_h3 = float((np.abs(got.astype(np.float64) - ref) * scale).max() / (mag * scale).max())
            e_nat = float((np.abs(nat.astype(np.float64) - ref) * scale).max() / (mag * scale).max())
        else:
            e_h3, e_nat = err(got, ref, mag), err(nat, ref, mag)
        print(kind, tile, "f16x3 %.3g native %.3g" % (e_h3, e_nat))
        assert e_h3 <= max(1.5 * e_nat, 4e-7), (kind, tile, e_h3, e_nat)


def test_h3_magnitude_bound_may_be_loose_but_not_wrong():
    """The activation scale comes from an UPPER BOUND: a record 2^8 too large changes nothing measurable; records merge by maximum;
    frcnn_amax_f32 measures exactly; frcnn_amax_clear zeroes; a launch without a record is refused."""
    from faster_rcnn_amd import _lib, ops
    rs = np.random.RandomState(11)
    x = rs.randn(1, 40, 52, 64).astype(np.float32)
    wt = (rs.randn(3, 3, 64, 128) * 0.05).astype(np.float32)
    pc = ops.PackedConv(wt)
    xd = torch.from_numpy(x).cuda()
    ref, mag = ref_conv(x, wt, 1, "same")
    tight = ops.conv2d(xd, pc, 1, "same", tile=84)
    assert getattr(xd, "_amax", None) is None                     # a measured record is not kept on an input buffer
    xd._amax = ops.amax_of(xd)
    assert float(xd._amax.max()) == float(np.abs(x).max())       # measured (no producer): exact
    e_tight = err(tight.cpu().numpy(), ref, mag)
    loose_in = torch.from_numpy(x).cuda()
    rec = torch.zeros_like(xd._amax)
    _lib.call("frcnn_amax_merge", rec.data_ptr(), xd._amax.data_ptr(), 256.0 * float(np.abs(x).max()), None, None)
    loose_in._amax = rec
    loose = ops.conv2d(loose_in, pc, 1, "same", tile=84)
    e_loose = err(loose.cpu().numpy(), ref, mag)
    print("tight %.3g loose (x256) %.3g" % (e_tight, e_loose))
    assert e_tight <= 4e-7 and e_loose <= 4e-7
    _lib.call("frcnn_amax_clear", rec.data_ptr(), 1, None)
    assert not rec.any().item()
    d = ops._conv_desc((1, 40, 52, 64), 3, 3, 128, 1, "same", 0, 0, 84)
    with pytest.raises(_lib.FrcnnError):
        _lib.call("frcnn_conv2d_fwd_h3", ctypes.byref(d), xd.data_ptr(), None, pc.h3_planes().data_ptr(), None, None, None, None,
                  tight.data_ptr(), None, None, 0, None)
    # a chain: pool and views inherit the bound, a native layer in an f16x3 scope leaves a record for the layer behind it
    before = ops.AMAX_MEASURED
    with ops.f32_engine("f16x3"):
        y = ops.conv2d(torch.from_numpy(rs.randn(1, 96, 128, 64).astype(np.float32)).cuda(), pc, 1, "same", "relu")      # 12 288 rows: the engine
        p = ops.pool2d(y, 3, 2, True)
        z = ops.conv2d(p, ops.PackedConv((rs.randn(1, 1, 128, 128) * 0.05).astype(np.float32)), 1, "valid")
        small = ops.conv2d(torch.from_numpy(rs.randn(1, 20, 20, 64).astype(np.float32)).cuda(), pc, 1, "same")        # 14 tiles: native, tracked
    assert ops.AMAX_MEASURED == before + 1                       # only the first input was measured by a pass of its own
    assert p._amax is y._amax and float(z._amax.max()) == float(z.abs().max())
    assert float(small._amax.max()) == float(small.abs().max())


def test_h3_position_major_layout_and_tap_skipping_are_bit_identical_to_nhwc():
    from faster_rcnn_amd import ops
    rs = np.random.RandomState(3)
    n, cin, cout = 300, 64, 128
    x = rs.randn(n, 7, 7, cin).astype(np.float32)
    wt = (rs.randn(3, 3, cin, cout) * 0.05).astype(np.float32)
    pc = ops.PackedConv(wt, np.ones(cout, np.float32), np.zeros(cout, np.float32))
    for tile in (81, 84, 86, 82):
        a = ops.conv2d(torch.from_numpy(x).cuda(), pc, 1, "same", "relu", tile=tile)                                   # (n,7,7,c)
        b = ops.conv2d(torch.from_numpy(np.ascontiguousarray(x.transpose(1, 2, 0, 3))).cuda(), pc, 1, "same", "relu", tile=tile, layout=1)   # (7,7,n,c)
        assert torch.equal(a, b.permute(2, 0, 1, 3))
    ref, mag = ref_conv(x, wt, 1, "same", None, None, None, "relu")
    assert err(a.cpu().numpy(), ref, mag) <= 6e-7


def test_h3_two_layers_in_one_launch_equal_two_launches():
    from faster_rcnn_amd import ops
    rs = np.random.RandomState(4)
    x = torch.from_numpy(rs.randn(1, 38, 63, 256).astype(np.float32)).cuda()
    w1 = (rs.randn(1, 1, 256, 128) * 0.06).astype(np.float32)
    w2 = (rs.randn(1, 1, 256, 512) * 0.06).astype(np.float32)
    s = lambda c: ((1 + 0.1 * rs.randn(c)).astype(np.float32), (0.1 * rs.randn(c)).astype(np.float32))
    (s1, h1), (s2, h2) = s(128), s(512)
    cat = np.concatenate([w1, w2], axis=3)
    both = ops.PackedConv(cat, np.concatenate([s1, s2]), np.concatenate([h1, h2]))
    for tile in (81, 84, 86):
        y1, y2 = ops.conv2d_dual(x, both, 128, 1, "valid", "relu", None, tile=tile)
        # the filter's scale is taken over the WHOLE concatenated filter: single-layer launches agree bit for bit when their filters
        # share that maximum, which two halves of one random draw need not -- so compare against slices of a one-layer launch
        whole = ops.conv2d(x, ops.PackedConv(cat, np.concatenate([s1, s2]), np.concatenate([h1, h2])), 1, "valid", None, tile=tile)
        assert torch.equal(y1, whole[..., :128].clamp(min=0)) and torch.equal(y2, whole[..., 128:])
        assert float(y1._amax.max()) == float(y1.abs().max()) and float(y2._amax.max()) == float(y2.abs().max())
    # a boundary that is not a tile boundary takes the per-column epilogue: 9 + 36 channels (the RPN output pair) padded to 64 k
    w3 = (rs.randn(1, 1, 256, 9) * 0.06).astype(np.float32); w4 = (rs.randn(1, 1, 256, 36) * 0.06).astype(np.float32)
    cat = np.concatenate([w3, w4], axis=3)
    pair = ops.PackedConv(cat, np.ones(45, np.float32), np.zeros(45, np.float32))
    y1, y2 = ops.conv2d_dual(x, pair, 9, 1, "valid", "sigmoid", None, tile=84)
    whole = ops.conv2d(x, ops.PackedConv(cat, np.ones(45, np.float32), np.zeros(45, np.float32)), 1, "valid", None, tile=84)
    assert torch.equal(y2, whole[..., 9:]) and torch.allclose(y1, torch.sigmoid(whole[..., :9]), atol=3e-6)
    assert float(y2._amax.max()) == float(y2.abs().max())


def test_h3_engine_policy_scope_and_tile_choice():
    from faster_rcnn_amd import _lib, ops
    lib = _lib.load()
    rs = np.random.RandomState(5)
    big = torch.from_numpy(rs.randn(1, 96, 128, 64).astype(np.float32)).cuda()          # 12 288 rows
    small = torch.from_numpy(rs.randn(1, 20, 20, 64).astype(np.float32)).cuda()
    pc = ops.PackedConv((rs.randn(3, 3, 64, 128) * 0.05).astype(np.float32))
    ops.CONV_PROFILE = []
    try:
        ops.conv2d(big, pc, 1, "same")
        with ops.f32_engine("f16x3"):
            ops.conv2d(big, pc, 1, "same")
            ops.conv2d(small, pc, 1, "same")                     # 14 tiles of 64x64, under X6_MIN_TILES: stays native
        ops.conv2d(big, pc, 1, "same")
        names = [r["kernel"] for r in ops.CONV_PROFILE]
        for r in ops.CONV_PROFILE:
            r["relaunch"]()                                      # the profile's re-launch closures run (bench.py times launches through them)
    finally:
        ops.CONV_PROFILE = None
    assert [("h3" in n) for n in names] == [False, True, False, False], names

    def cfg(shape, k, cout, n1=0, stride=1, padding="same"):
        d = ops._conv_desc(shape, k, k, cout, stride, padding, 0, 0, 0)
        return lib.frcnn_conv2d_h3_config(ctypes.byref(d), n1)
    assert cfg((1, 149, 249, 64), 3, 64) == 84                       # stage 2's 3x3
    assert cfg((1, 600, 1000, 64), 3, 64) == 87                      # VGG16 conv1_2
    assert cfg((1, 75, 125, 128), 3, 128) == 84                      # under 256 tiles of 128x128
    assert cfg((300, 7, 7, 512), 1, 2048) == 86 and cfg((300, 7, 7, 512), 3, 512) == 86
    assert cfg((1, 149, 249, 64), 1, 320, n1=64, padding="valid") == 84
    assert lib.frcnn_conv2d_h3_config(None, 0) < 0
    # round 6: beside other passes' launches (tile code 50) the big tile pays from 128 tiles of 128x128 on, 128x128 on eight waves below
    def shared(shape, k, cout):
        d = ops._conv_desc(shape, k, k, cout, 1, "same", 0, 0, 50)
        return lib.frcnn_conv2d_h3_config(ctypes.byref(d), 0)
    assert cfg((4, 38, 63, 256), 3, 256) == 84 and shared((4, 38, 63, 256), 3, 256) == 86      # stage 4 of a four-image pass: 150 tiles
    assert shared((1, 38, 63, 256), 3, 256) == 84                                               # one image: 38 tiles, 2 394 rows -- left alone
    assert shared((2, 38, 63, 256), 3, 256) == 81                                               # two images: 76 tiles, 4 788 rows
    assert shared((4, 149, 249, 64), 3, 64) == 87 and shared((4, 600, 1000, 128), 3, 128) == 86
    xs = torch.from_numpy(rs.randn(4, 38, 63, 256).astype(np.float32)).cuda()
    pcs = ops.PackedConv((rs.randn(3, 3, 256, 256) * 0.03).astype(np.float32), np.ones(256, np.float32), (0.1 * rs.randn(256)).astype(np.float32))
    with ops.f32_engine("f16x3"), ops.conv_workspace(ops.NO_SPLIT_K):
        alone = ops.conv2d(xs, pcs, 1, "same", "relu")
        with ops.tile_policy(True):
            beside = ops.conv2d(xs, pcs, 1, "same", "relu")
        two = ops.conv2d(xs[:2].contiguous(), pcs, 1, "same", "relu", tile=81)
    assert torch.equal(alone, beside) and torch.equal(two, alone[:2])                          # the tile does not change a bit
    # cin % 32 != 0 has no split form: an explicit tile code is not honoured silently
    odd = ops.PackedConv((rs.randn(1, 1, 48, 64) * 0.1).astype(np.float32))
    y = ops.conv2d(torch.from_numpy(rs.randn(1, 8, 8, 48).astype(np.float32)).cuda(), odd, 1, "valid", tile=81)
    assert y.shape == (1, 8, 8, 64)                                  # (took the native generic kernel)


def test_h3_planes_reassemble_the_filter():
    """frcnn_pack_conv_weights_h3: header = max|w|; hi + lo * 2^-11 under the header's scale gives every weight back to ONE f32
    unit in its last place (2^-23 of ITS OWN magnitude: a 13-bit residual kept to 11 bits), half of them exactly; weights 2^27 under
    the largest one keep an absolute 2^-49 of it."""
    from faster_rcnn_amd import _lib, ops
    rs = np.random.RandomState(21)
    for sh, mag in (((3, 3, 64, 96), 0.05), ((1, 1, 256, 1024), 3e-9), ((1, 1, 32, 8), 700.0)):
        pc = ops.PackedConv((rs.randn(*sh) * mag).astype(np.float32))
        raw = pc.h3_planes()
        assert raw.numel() == _lib.load().frcnn_conv_h3_planes_bytes(pc.w.shape[0], pc.w.shape[1])
        amax = float(raw[:4].view(torch.float32).item())
        assert amax == float(pc.w.abs().max())
        e = 14 - int(np.floor(np.log2(amax)))
        planes = raw[16:].view(torch.float16).view(2, *pc.w.shape).double()
        back = (planes[0] + planes[1] / 2048.0) * 2.0 ** -e
        w64 = pc.w.double()
        tol = torch.maximum(2.0 ** -23 * w64.abs(), torch.full_like(w64, 2.0 ** -49 * amax))
        assert bool(((back - w64).abs() <= tol).all()), float(((back - w64).abs() / tol).max())
        assert float(((back - w64).abs() > 0).double().mean()) < 0.75          # a good share of the weights come back exactly
        assert float(planes[0].abs().max()) < 32768.0 and float(planes[0].abs().max()) >= 16384.0


def test_h3_split_k_small_grid_long_k():
    """rpn_conv1 / stage-4 shapes (2 394 rows, k = 9 216 / 2 304): the engine's 64x64 split-K form against fp64, bitwise
    reproducible, tickets left zero (a second launch on the same workspace)."""
    from faster_rcnn_amd import ops
    rs = np.random.RandomState(8)
    for (h, w, cin, cout, k) in ((38, 63, 1024, 512, 3), (38, 63, 256, 256, 3), (19, 31, 2048, 192, 1)):
        x = rs.randn(1, h, w, cin).astype(np.float32)
        wt = (rs.randn(k, k, cin, cout) * np.sqrt(2.0 / (k * k * cin))).astype(np.float32)
        scale = (1 + 0.1 * rs.randn(cout)).astype(np.float32); shift = (0.1 * rs.randn(cout)).astype(np.float32)
        pc = ops.PackedConv(wt, scale, shift)
        xd = torch.from_numpy(x).cuda()
        ws = ops.ConvWorkspace()
        ops.CONV_PROFILE = []
        try:
            with ops.f32_engine("f16x3"), ops.conv_workspace(ws):
                got = ops.conv2d(xd, pc, 1, "same", "relu")
                again = ops.conv2d(xd, pc, 1, "same", "relu", out=torch.full_like(got, 3.0))
            names = [r["kernel"] for r in ops.CONV_PROFILE]
        finally:
            ops.CONV_PROFILE = None
        assert names[0].endswith("split-K") and "h3" in names[0], names
        ref, mag = ref_conv(x, wt, 1, "same", scale, shift, None, "relu")
        assert err(got.cpu().numpy(), ref, mag) <= 5e-7
        assert torch.equal(got, again)
        assert float(got._amax.max()) == float(got.abs().max())
        assert not ws.buf[:16384].any().item()                   # tickets back to zero
    # the eight-wave 128x128 split-K form (explicit: tile code 88)
    x = rs.randn(1, 56, 56, 512).astype(np.float32)
    wt = (rs.randn(3, 3, 512, 512) * np.sqrt(2.0 / 4608)).astype(np.float32)
    pc = ops.PackedConv(wt)
    ws = ops.ConvWorkspace()
    with ops.conv_workspace(ws):
        got = ops.conv2d(torch.from_numpy(x).cuda(), pc, 1, "same", None, tile=88)
    ref, mag = ref_conv(x, wt, 1, "same")
    assert err(got.cpu().numpy(), ref, mag) <= 5e-7 and not ws.buf[:16384].any().item()


def test_h3_plane_tensors_between_layers():
    """frcnn_conv2d_fwd_h3_planes: a 1x1 -> 3x3 -> 1x1 (+ residual) chain over 300 position-major RoI crops -- the detector head's
    block (resnet.py:282-313) -- with the two inner tensors handed on as fp16 planes, against the same chain through f32 tensors and
    against fp64: same bars.  The planes ARE the producer's f32 values to one unit in the last place, their exponent comes from the
    bound (never from the data), and a PlaneTensor offered to a launch that cannot read it is refused."""
    from faster_rcnn_amd import _lib, ops
    rs = np.random.RandomState(31)
    n, cin, c1, c3 = 300, 64, 384, 512
    x = np.maximum(rs.randn(7, 7, n, cin), 0).astype(np.float32)                 # position-major, post-ReLU like the crops
    w1 = (rs.randn(1, 1, cin, c1) * np.sqrt(2.0 / cin)).astype(np.float32)
    w2 = (rs.randn(3, 3, c1, c1) * np.sqrt(2.0 / (9 * c1))).astype(np.float32)
    w3 = (rs.randn(1, 1, c1, c3) * np.sqrt(2.0 / c1)).astype(np.float32)
    sb = lambda c: ((1 + 0.1 * rs.randn(c)).astype(np.float32), (0.1 * rs.randn(c)).astype(np.float32))
    (s1, h1), (s2, h2), (s3, h3) = sb(c1), sb(c1), sb(c3)
    p1, p2, p3 = ops.PackedConv(w1, s1, h1), ops.PackedConv(w2, s2, h2), ops.PackedConv(w3, s3, h3)
    res = rs.randn(7, 7, n, c3).astype(np.float32)
    xd, rd = torch.from_numpy(x).cuda(), torch.from_numpy(res).cuda()
    with ops.f32_engine("f16x3"), ops.conv_workspace(ops.NO_SPLIT_K):
        a1 = ops.conv2d(xd, p1, 1, "valid", "relu", layout=1)
        a2 = ops.conv2d(a1, p2, 1, "same", "relu", layout=1)
        a3 = ops.conv2d(a2, p3, 1, "valid", "relu", residual=rd, layout=1)
        ops.CONV_PROFILE = []
        try:
            b1 = ops.conv2d(xd, p1, 1, "valid", "relu", layout=1, planes_out=True)
            b2 = ops.conv2d(b1, p2, 1, "same", "relu", layout=1, planes_out=True)
            b3 = ops.conv2d(b2, p3, 1, "valid", "relu", residual=rd, layout=1)
            names = [r["kernel"] for r in ops.CONV_PROFILE]
            for r in ops.CONV_PROFILE:
                r["relaunch"]()
        finally:
            ops.CONV_PROFILE = None
        # a launch that cannot take planes (64x64 tiles): the request is ignored on the way out, refused on the way in
        small = ops.conv2d(torch.from_numpy(x[:, :, :40]).cuda().contiguous(), p1, 1, "valid", "relu", layout=1, planes_out=True)
        assert isinstance(small, torch.Tensor)
        with pytest.raises(_lib.FrcnnError):
            ops.conv2d(b1, ops.PackedConv((rs.randn(1, 1, c1, 32) * 0.05).astype(np.float32)), 1, "valid", layout=1)
    assert isinstance(b1, ops.PlaneTensor) and isinstance(b2, ops.PlaneTensor) and isinstance(b3, torch.Tensor)
    assert names == ["k_conv_igemm_h3_db<2,1,4,4>", "k_conv_igemm_h3_db<2,1,4,4,planes>", "k_conv_igemm_h3_db<2,1,4,4,planes>"], names
    # the planes are the producer's values (one unit in the last place; exact zeros stay zeros)
    v1, f1 = b1.float().double(), a1.double()
    assert bool(((v1 - f1).abs() <= 2.0 ** -23 * f1.abs() + 2.0 ** -40 * f1.abs().max()).all())
    assert float(b1._amax.max()) == float(a1.abs().max())
    bc, bd = p1.h3_bound()
    bound = bc * float(xd.abs().max()) + bd
    e = int(b1.exponent.item())
    assert 2.0 ** 14 <= bound * 2.0 ** e < 2.0 ** 15 and float(a1.abs().max()) <= bound
    assert float(b1.planes[0].abs().max()) < 32768.0
    # the chain: same bars against fp64 as the f32-tensor chain, and the two agree far inside them
    def ref_chain():
        t = np.transpose(x, (2, 0, 1, 3))
        r1, _ = ref_conv(t, w1, 1, "valid", s1, h1, None, "relu")
        r2, _ = ref_conv(r1.astype(np.float64), w2, 1, "same", s2, h2, None, "relu")
        r3, m3 = ref_conv(r2, w3, 1, "valid", s3, h3, np.transpose(res, (2, 0, 1, 3)), "relu")
        return r3, m3
    r3, m3 = ref_chain()
    to_nhwc = lambda t: t.permute(2, 0, 1, 3).cpu().numpy()
    e_f32, e_pl = err(to_nhwc(a3), r3, m3), err(to_nhwc(b3), r3, m3)
    print("chain error vs fp64: f32 tensors %.3g, plane tensors %.3g" % (e_f32, e_pl))
    assert e_pl <= max(1.5 * e_f32, 6e-7)
    assert float((a3 - b3).abs().max() / a3.abs().max()) < 2e-6
    # bitwise reproducible
    with ops.f32_engine("f16x3"), ops.conv_workspace(ops.NO_SPLIT_K):
        c1_ = ops.conv2d(xd, p1, 1, "valid", "relu", layout=1, planes_out=True)
        c3_ = ops.conv2d(ops.conv2d(c1_, p2, 1, "same", "relu", layout=1, planes_out=True), p3, 1, "valid", "relu", residual=rd, layout=1)
    assert torch.equal(c3_, b3) and torch.equal(c1_.planes, b1.planes)


def test_h3_block_output_as_planes_feeds_the_next_conv_and_its_shortcut():
    """Round 6 (frcnn_conv2d_fwd_h3_planes_res): a head block's output written ONCE, as planes; the next block's branch2a stages them and its
    closing 1x1 reads the shortcut back from them -- (hi + lo / 2048) * 2^-e, the tensor's elements to 22-24 bits.  Against the same two
    blocks through f32 tensors: the planes ARE the f32 block output to one unit in its last place, the second block's output agrees to
    2^-20 of the tensor's scale, and the error against fp64 stays at the f32 chain's."""
    from faster_rcnn_amd import _lib, ops
    rs = np.random.RandomState(91)
    n, c, cw = 300, 512, 1024                                                # 14 700 rows; inner width c, block width cw (>= 256 tiles of 128x128 per launch)
    x0 = np.maximum(rs.randn(n, 7, 7, cw), 0).astype(np.float32)
    mk = lambda kh, ci, co: ops.PackedConv((rs.randn(kh, kh, ci, co) * np.sqrt(2.0 / (kh * kh * ci))).astype(np.float32),
                                           (1 + 0.1 * rs.randn(co)).astype(np.float32), (0.1 * rs.randn(co)).astype(np.float32))
    blocks = [(mk(1, cw, c), mk(3, c, c), mk(1, c, cw)) for _ in range(2)]
    xd = torch.from_numpy(x0).cuda()

    def chain(block_planes):
        x = xd
        with ops.f32_engine("f16x3"), ops.conv_workspace(ops.NO_SPLIT_K):
            for i, (a, b, cc) in enumerate(blocks):
                t = ops.conv2d(x, a, 1, "valid", "relu", planes_out=True)
                t = ops.conv2d(t, b, 1, "same", "relu", planes_out=True)
                x = ops.conv2d(t, cc, 1, "valid", "relu", residual=x, planes_out=block_planes and i == 0)
        return x
    ops.CONV_PROFILE = []
    try:
        y_pl = chain(True)
        names = [r["kernel"] for r in ops.CONV_PROFILE]
        for r in ops.CONV_PROFILE:
            r["relaunch"]()
    finally:
        ops.CONV_PROFILE = None
    y_f = chain(False)
    assert isinstance(y_pl, torch.Tensor) and names[3] == "k_conv_igemm_h3_db<2,1,4,4,planes>", names     # block 2's branch2a read planes
    assert torch.equal(chain(True), y_pl)                                    # reproducible
    scale = float(y_f.abs().max())
    assert float((y_pl - y_f).abs().max()) <= 2.0 ** -20 * scale, float((y_pl - y_f).abs().max()) / scale
    # the first block's output, both ways
    with ops.f32_engine("f16x3"), ops.conv_workspace(ops.NO_SPLIT_K):
        a, b, cc = blocks[0]
        t = ops.conv2d(ops.conv2d(xd, a, 1, "valid", "relu", planes_out=True), b, 1, "same", "relu", planes_out=True)
        o_f = ops.conv2d(t, cc, 1, "valid", "relu", residual=xd)
        o_p = ops.conv2d(t, cc, 1, "valid", "relu", residual=xd, planes_out=True)
        assert isinstance(o_p, ops.PlaneTensor)
        v, f = o_p.float().double(), o_f.double()
        assert bool(((v - f).abs() <= 2.0 ** -22 * f.abs() + 2.0 ** -38 * f.abs().max()).all())
        # a residual as planes on a launch that cannot read them is refused, as a plane input is
        with pytest.raises(_lib.FrcnnError):
            ops.conv2d(xd[:8].contiguous(), cc if False else blocks[0][0], 1, "valid", "relu", residual=ops.PlaneTensor((8, 7, 7, c)))


def test_pooling_writes_the_planes_the_next_conv_reads():
    """Round 6 (frcnn_pool2d_fwd_planes): VGG's max-pools hand their map to block<n>_conv1 as the planes it multiplies -- the scale from the
    INPUT's magnitude record (a window's maximum cannot exceed it).  The planes are the f32 pooling's values to one unit in the last place,
    and the convolution over them equals the convolution over the f32 map far inside the engine's bars."""
    from faster_rcnn_amd import ops
    rs = np.random.RandomState(17)
    x = torch.from_numpy(np.maximum(rs.randn(2, 150, 250, 128), 0).astype(np.float32)).cuda()
    pc = ops.PackedConv((rs.randn(3, 3, 128, 256) * np.sqrt(2.0 / (9 * 128))).astype(np.float32), np.ones(256, np.float32), (0.1 * rs.randn(256)).astype(np.float32))
    arena = ops.AmaxArena()
    with ops.f32_engine("f16x3"), ops.conv_workspace(ops.NO_SPLIT_K), ops.amax_arena(arena):
        x._amax = ops.amax_of(x)
        p_f = ops.pool2d(x, 2, 2, True)
        p_p = ops.pool2d(x, 2, 2, True, planes_out=True)
        assert isinstance(p_p, ops.PlaneTensor) and p_p.shape == tuple(p_f.shape)
        v, f = p_p.float().double(), p_f.double()
        assert bool(((v - f).abs() <= 2.0 ** -22 * f.abs() + 2.0 ** -38 * f.abs().max()).all())
        e = int(p_p.exponent.item())
        assert 2.0 ** 14 <= float(x.abs().max()) * 2.0 ** e < 2.0 ** 15
        y_f = ops.conv2d(p_f, pc, 1, "same", "relu")
        y_p = ops.conv2d(p_p, pc, 1, "same", "relu")
        assert int(arena.status().item()) == 0
    assert float((y_f - y_p).abs().max()) <= 2e-6 * float(y_f.abs().max())
    a_p = ops.pool2d(x, 2, 2, False)                                   # outside an f16x3 scope the request is ignored
    assert isinstance(ops.pool2d(x, 2, 2, False, planes_out=True), torch.Tensor) and torch.equal(a_p, ops.pool2d(x, 2, 2, False, planes_out=True))


@pytest.mark.parametrize("layout", [0, 1])
def test_h3_ring_equals_the_double_buffer_bit_for_bit(layout):
    """Round 6: plane-input launches with long reductions walk a three-stage direct-to-LDS ring (csrc/conv_h3.hip h3_ring_tile; tile code
    86), short ones and tile code 85 the register-staged double buffer: same chunk order, same products -- the same bits, in both row
    layouts (position-major: with tap skipping), for plane and f32 outputs, with a ragged last tile."""
    from faster_rcnn_amd import ops
    rs = np.random.RandomState(71)
    n, c = 333, 256                                                          # 333 * 49 rows: not a multiple of 256
    shape = (7, 7, n, c) if layout else (n, 7, 7, c)
    x = torch.from_numpy(np.maximum(rs.randn(*shape), 0).astype(np.float32)).cuda()
    p1 = ops.PackedConv((rs.randn(1, 1, c, c) * np.sqrt(2.0 / c)).astype(np.float32), np.ones(c, np.float32), np.zeros(c, np.float32))
    p3 = ops.PackedConv((rs.randn(3, 3, c, c) * np.sqrt(2.0 / (9 * c))).astype(np.float32), (1 + 0.1 * rs.randn(c)).astype(np.float32), (0.1 * rs.randn(c)).astype(np.float32))
    res = torch.from_numpy(rs.randn(*shape).astype(np.float32)).cuda()
    with ops.f32_engine("f16x3"), ops.conv_workspace(ops.NO_SPLIT_K):
        t = ops.conv2d(x, p1, 1, "valid", "relu", layout=layout, planes_out=True, tile=86)
        assert isinstance(t, ops.PlaneTensor)
        ring_p = ops.conv2d(t, p3, 1, "same", "relu", layout=layout, planes_out=True, tile=86)      # 72 chunks: the ring
        db_p = ops.conv2d(t, p3, 1, "same", "relu", layout=layout, planes_out=True, tile=85)
        ring_f = ops.conv2d(t, p3, 1, "same", "relu", residual=res, layout=layout, tile=86)
        db_f = ops.conv2d(t, p3, 1, "same", "relu", residual=res, layout=layout, tile=85)
    assert torch.equal(ring_p.planes, db_p.planes) and int(ring_p.exponent.item()) == int(db_p.exponent.item())
    assert torch.equal(ring_f, db_f) and float(ring_f.abs().max()) > 0.0
    assert torch.equal(ring_p._amax, db_p._amax)


@pytest.mark.parametrize("size", [(160, 224), (161, 227), (600, 1000), (37, 29)])
def test_h3_fused_stem_equals_conv_plus_pool(size):
    """frcnn_stem_h3_fwd: conv1 7x7 / 2 'same' + folded BatchNorm + ReLU + MaxPooling2D((3,3), (2,2)) in one f16x3 launch (resnet.py:408-412)
    against the two-launch form (native f32 conv, then the pool) and against fp64: the conv values within the engine's bar, the pool
    exact on them -- even and odd sizes (TF 'same' pads one side only on even sizes), patches that hang over the image edge."""
    from faster_rcnn_amd import ops
    h, w = size
    rs = np.random.RandomState(h * 1000 + w)
    x = (rs.randint(0, 256, (2, h, w, 3)).astype(np.float64) - np.array([103.939, 116.779, 123.68])).astype(np.float32)
    wt = (rs.randn(7, 7, 3, 64) * np.sqrt(2.0 / 147)).astype(np.float32)
    scale = (1 + 0.1 * rs.randn(64)).astype(np.float32)
    shift = (0.5 * rs.randn(64)).astype(np.float32)
    xd = torch.from_numpy(x).cuda()
    got = ops.stem_h3(xd, ops.PackedStemH3(wt, scale, shift))
    two = ops.pool2d(ops.conv2d(xd, ops.PackedConv(wt, scale, shift), 2, "same", "relu"), 3, 2, True)
    assert got.shape == two.shape
    conv64, mag = ref_conv(x, wt, 2, "same", scale, shift, None, "relu")
    t = torch.from_numpy(conv64).permute(0, 3, 1, 2)
    pool64 = torch.nn.functional.max_pool2d(t, 3, 2).permute(0, 2, 3, 1).numpy()
    magp = torch.nn.functional.max_pool2d(torch.from_numpy(mag).permute(0, 3, 1, 2), 3, 2).permute(0, 2, 3, 1).numpy()     # (an upper bound of the selected pixel's)
    e_fused = float((np.abs(got.cpu().numpy() - pool64) / np.maximum(magp, 1e-30)).max())
    e_two = float((np.abs(two.cpu().numpy() - pool64) / np.maximum(magp, 1e-30)).max())
    print(size, "fused f16x3 stem %.3g, native conv + pool %.3g" % (e_fused, e_two))
    assert e_fused <= max(1.5 * e_two, 4e-7)
    assert float((got - two).abs().max() / two.abs().max()) < 1e-5
    assert float(got._amax.max()) == float(got.abs().max())
    assert torch.equal(ops.stem_h3(xd, ops.PackedStemH3(wt, scale, shift)), got)          # bitwise reproducible


def test_h3_roi_resampling_writes_the_planes_the_next_conv_reads():
    """frcnn_roi_crop_resize_fwd_planes: the crops of RoiResizeConv (custom_layers.py:35-56) as fp16 planes under the scale the map's
    record gives BEFORE the launch (a bilinear sample cannot exceed the map's maximum; a rejected RoI yields the fill vector): the
    planes equal the f32 crops to one unit in the last place, and the 3x3 convolution behind them gives what it gives on the f32 crops."""
    from faster_rcnn_amd import ops
    rs = np.random.RandomState(41)
    rows, cols, C, n = 38, 63, 512, 300
    with ops.f32_engine("f16x3"), ops.conv_workspace(ops.NO_SPLIT_K):
        base = torch.from_numpy(rs.randn(1, rows, cols, 64).astype(np.float32)).cuda()
        feat = ops.conv2d(base, ops.PackedConv((rs.randn(1, 1, 64, C) * 0.2).astype(np.float32)), 1, "valid")       # a producer: the map has a record
        x1 = rs.randint(0, cols - 3, n); y1 = rs.randint(0, rows - 3, n)
        rois = np.stack([x1, y1, np.minimum(cols - 1, x1 + 2 + rs.randint(0, 20, n)), np.minimum(rows - 1, y1 + 2 + rs.randint(0, 14, n))], axis=1).astype(np.float32)
        rois[7] = [5, 5, 5, 5]                                               # an empty RoI: the fill vector
        rd = torch.from_numpy(rois).cuda()
        fill = torch.from_numpy((rs.randn(C) * 3).astype(np.float32)).cuda()
        f32 = ops.roi_crop_resize(feat, rd, 7, fill=fill, relu=True, layout=1)
        pl = ops.roi_crop_resize(feat, rd, 7, fill=fill, relu=True, layout=1, planes_out=True)
        assert isinstance(pl, ops.PlaneTensor) and pl.shape == tuple(f32.shape)
        v, f = pl.float().double(), f32.double()
        assert bool(((v - f).abs() <= 2.0 ** -23 * f.abs() + 2.0 ** -40 * f.abs().max()).all())
        bound = max(float(feat.abs().max()), float(fill.abs().max()))
        assert float(pl._amax.max()) == bound and 2.0 ** 14 <= bound * 2.0 ** int(pl.exponent.item()) < 2.0 ** 15
        pc = ops.PackedConv((rs.randn(3, 3, C, C) * np.sqrt(2.0 / (9 * C))).astype(np.float32))
        assert ops.conv_accepts_planes(pl.shape, pc, 1, "same", "relu", 1)
        a = ops.conv2d(f32, pc, 1, "same", "relu", layout=1)
        b = ops.conv2d(pl, pc, 1, "same", "relu", layout=1)
        assert float((a - b).abs().max() / a.abs().max()) < 2e-6
        # without a record on the map the request falls back to the f32 tensor
        plain = ops.roi_crop_resize(feat.clone(), rd, 7, fill=fill, relu=True, layout=1, planes_out=True)
        assert isinstance(plain, torch.Tensor) and torch.equal(plain, f32)


def test_refresh_h3_planes_equals_the_single_filter_pack():
    """frcnn_refresh_h3_planes (a training step's re-derivation of every trainable filter's planes after the update): header and both
    planes of many packed filters in three launches == frcnn_pack_conv_weights_h3 per filter, bit for bit -- also after the weights
    have CHANGED in place (the header's maximum is measured again, a smaller one too); malformed jobs are refused."""
    from faster_rcnn_amd import _lib, ops
    rs = np.random.RandomState(23)
    shapes = [(3, 3, 64, 96), (1, 1, 256, 1024), (1, 1, 32, 8), (3, 3, 512, 512)] * 15          # 60 jobs: more than one table
    packs = [ops.PackedConv((rs.randn(*sh) * 0.05 * (1 + i % 5)).astype(np.float32)) for i, sh in enumerate(shapes)]
    outs = [torch.full_like(pc.h3_planes(), 7) for pc in packs]
    jobs = (_lib.X6Job * len(packs))()
    for j, pc, o in zip(jobs, packs, outs):
        j.w_packed, j.planes_bf16, j.rows, j.kpad = pc.w.data_ptr(), o.data_ptr(), pc.w.shape[0], pc.w.shape[1]
    for scale in (1.0, 0.125, 3.0):
        for pc in packs:
            pc.w.mul_(scale)                                        # in place, like the optimiser's re-pack
            pc._h3 = None                                           # (the single-filter form derives its planes afresh)
        want = [pc.h3_planes().clone() for pc in packs]
        _lib.call("frcnn_refresh_h3_planes", jobs, len(packs), None)
        torch.cuda.synchronize()
        for o, w_ in zip(outs, want):
            assert torch.equal(o, w_)
            assert float(o[:4].view(torch.float32).item()) > 0       # the header: max|w|
    jobs[1].kpad = 48
    with pytest.raises(_lib.FrcnnError):
        _lib.call("frcnn_refresh_h3_planes", jobs, 2, None)
    _lib.call("frcnn_refresh_h3_planes", None, 0, None)              # nothing to do is fine
