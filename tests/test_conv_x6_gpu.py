"""The split-bf16 fp32 conv engine (csrc/conv_x6.hip, frcnn_conv2d_fwd_x6): fp32 convolution on the bf16 matrix cores by
exact three-way operand splitting.  Held to the SAME bars as the native f32 MFMA kernel -- error against an fp64 reference
at or below the native kernel's, network outputs within 1e-4 of the oracle -- plus the properties the native engine's
tests check: layouts agree, skipped padding taps change nothing, runs are bitwise reproducible."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
F = torch.nn.functional


def ref_conv(x, w, stride, padding, scale=None, shift=None, residual=None, act=None):
    """fp64 reference: TF SAME / VALID padding, NHWC x HWIO."""
    from faster_rcnn_amd import ops
    xt = torch.from_numpy(x).double().permute(0, 3, 1, 2)
    wt = torch.from_numpy(w).double().permute(3, 2, 0, 1)
    if padding == "same":
        _, pt = ops.same_pad(x.shape[1], w.shape[0], stride)
        _, pl = ops.same_pad(x.shape[2], w.shape[1], stride)
        ho, wo = -(-x.shape[1] // stride), -(-x.shape[2] // stride)
        pb = max((ho - 1) * stride + w.shape[0] - x.shape[1] - pt, 0)
        pr = max((wo - 1) * stride + w.shape[1] - x.shape[2] - pl, 0)
        xt = F.pad(xt, (pl, pr, pt, pb))
    y = F.conv2d(xt, wt, stride=stride).permute(0, 2, 3, 1)
    mag = F.conv2d(xt.abs(), wt.abs(), stride=stride).permute(0, 2, 3, 1)
    if scale is not None:
        y, mag = y * torch.from_numpy(scale).double(), mag * torch.from_numpy(np.abs(scale)).double()
    if shift is not None:
        y = y + torch.from_numpy(shift).double()
    if residual is not None:
        y = y + torch.from_numpy(residual).double()
    if act == "relu":
        y = y.clamp(min=0)
    return y.numpy(), mag.numpy()


def err(got, ref, mag):
    return float((np.abs(got.astype(np.float64) - ref) / np.maximum(mag, 1e-30)).max())


CASES = [
    # n, h, w, cin, cout, k, stride, padding, tile
    (1, 40, 52, 64, 128, 1, 1, "valid", 71),
    (1, 40, 52, 64, 128, 1, 1, "valid", 74),
    (2, 23, 31, 64, 96, 3, 1, "same", 71),        # ragged rows / columns, batch, halo
    (2, 23, 31, 64, 96, 3, 1, "same", 73),
    (1, 33, 29, 128, 192, 3, 2, "same", 75),      # stride 2 SAME (pad on one side only for odd sizes)
    (1, 30, 44, 256, 64, 1, 2, "valid", 74),      # strided 1x1 (conv_block shortcut)
    (3, 14, 14, 96, 256, 3, 1, "same", 72),
    (2, 23, 31, 64, 96, 3, 1, "same", 76),        # 256x128 on sixteen waves, two LDS buffers
    (1, 40, 52, 64, 320, 1, 1, "valid", 76),
    (1, 33, 29, 128, 192, 3, 2, "same", 76),
    (2, 23, 31, 64, 64, 3, 1, "same", 77),        # 128x64 on four waves: the 64-column layers
    (1, 40, 52, 96, 160, 1, 1, "valid", 77),      # ragged column tiles (160 = 2.5 x 64)
]


@pytest.mark.parametrize("case", CASES)
def test_x6_matches_fp64_as_well_as_the_native_kernel(case):
    from faster_rcnn_amd import ops
    n, h, w, cin, cout, k, stride, padding, tile = case
    rs = np.random.RandomState(1000 + CASES.index(case))            # (hash() of a tuple with strings changes from process to process)
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
    e_x6, e_nat = err(got.cpu().numpy(), ref, mag), err(nat.cpu().numpy(), ref, mag)
    print(case, "x6 %.3g native %.3g" % (e_x6, e_nat))
    assert got.shape == nat.shape
    assert e_x6 <= 5e-7 and e_x6 <= max(3.0 * e_nat, 4e-7)          # an fp32 GEMM: the error level of the native f32 MFMA kernel (2-3e-7 of sum|ab|)
    assert err(got_r.cpu().numpy(), ref_r, mag) <= 6e-7
    # bitwise reproducible, and the launch does not depend on what the output buffer held
    again = ops.conv2d(xd, pc, stride, padding, "relu", tile=tile, out=torch.full_like(got, 7.0))
    assert torch.equal(again, got)


@pytest.mark.parametrize("kind", ["all_positive", "wide_exponents", "integers", "subnormal_neighbours"])
def test_x6_error_on_operands_that_do_not_cancel(kind):
    """The three-way split is exact for every finite f32 whose bf16 pieces stay normal; what the engine drops are the three partial
    products below 2^-24 of a product.  Random normal operands hide a systematic error behind cancellation: here nothing cancels
    (all-positive operands: sum|ab| = |sum ab|), magnitudes span 2^-12..2^12 inside one dot product, operands are integers whose
    products and sums are exact in f32 (the result must be EXACT), and values sit next to the bf16-subnormal range."""
    from faster_rcnn_amd import ops
    rs = np.random.RandomState({"all_positive": 1, "wide_exponents": 2, "integers": 3, "subnormal_neighbours": 4}[kind])
    n, h, w, cin, cout, k = 1, 30, 44, 128, 128, 3
    if kind == "all_positive":
        x = rs.uniform(0.5, 1.0, (n, h, w, cin)).astype(np.float32); wt = rs.uniform(0.5, 1.0, (k, k, cin, cout)).astype(np.float32)
    elif kind == "wide_exponents":
        x = (rs.randn(n, h, w, cin) * 2.0 ** rs.randint(-12, 13, (n, h, w, cin))).astype(np.float32)
        wt = (rs.randn(k, k, cin, cout) * 2.0 ** rs.randint(-12, 13, (k, k, cin, cout))).astype(np.float32)
    elif kind == "integers":
        x = rs.randint(-60, 61, (n, h, w, cin)).astype(np.float32); wt = rs.randint(-60, 61, (k, k, cin, cout)).astype(np.float32)
    else:
        x = (rs.randn(n, h, w, cin) * 1e-30).astype(np.float32); wt = (rs.randn(k, k, cin, cout) * 1e-6).astype(np.float32)
    pc = ops.PackedConv(wt)
    xd = torch.from_numpy(x).cuda()
    ref, mag = ref_conv(x, wt, 1, "same")
    for tile in (71, 74, 76):
        with ops.conv_workspace(ops.NO_SPLIT_K):                 # like for like: both sum k = 0 .. K - 1 in ONE accumulator (a split-K launch
            got = ops.conv2d(xd, pc, 1, "same", None, tile=tile).cpu().numpy()     # sums slices pairwise and would flatter either side)
            nat = ops.conv2d(xd, pc, 1, "same", None, tile=0).cpu().numpy()
        if kind == "integers":                                   # |sum| < 9 * 128 * 3600 < 2^24: every partial sum is an integer f32 holds
            assert np.array_equal(got.astype(np.float64), ref) and np.array_equal(nat.astype(np.float64), ref)
            continue
        if kind == "subnormal_neighbours":                       # products ~1e-36: f32 subnormal range starts at 1.2e-38, results are normal
            scale = 1e36
            e_x6 = float((np.abs(got.astype(np.float64) - ref) * scale).max() / (mag * scale).max())
            e_nat = float((np.abs(nat.astype(np.float64) - ref) * scale).max() / (mag * scale).max())
        else:
            e_x6, e_nat = err(got, ref, mag), err(nat, ref, mag)
        print(kind, tile, "x6 %.3g native %.3g" % (e_x6, e_nat))
        assert e_x6 <= max(1.5 * e_nat, 4e-7), (kind, tile, e_x6, e_nat)


def test_x6_position_major_layout_and_tap_skipping_are_bit_identical_to_nhwc():
    """The detector head's [7][7][roi][c] tensors: a 128-row tile covers one or two output positions and skips the taps that
    only meet zero padding -- exact zeros, so the NHWC launch of the same engine must give the same bits."""
    from faster_rcnn_amd import ops
    rs = np.random.RandomState(3)
    n, cin, cout = 300, 64, 128
    x = rs.randn(n, 7, 7, cin).astype(np.float32)
    wt = (rs.randn(3, 3, cin, cout) * 0.05).astype(np.float32)
    pc = ops.PackedConv(wt, np.ones(cout, np.float32), np.zeros(cout, np.float32))
    for tile in (71, 74, 76):
        a = ops.conv2d(torch.from_numpy(x).cuda(), pc, 1, "same", "relu", tile=tile)                                   # (n,7,7,c)
        b = ops.conv2d(torch.from_numpy(np.ascontiguousarray(x.transpose(1, 2, 0, 3))).cuda(), pc, 1, "same", "relu", tile=tile, layout=1)   # (7,7,n,c)
        assert torch.equal(a, b.permute(2, 0, 1, 3))
    ref, mag = ref_conv(x, wt, 1, "same", None, None, None, "relu")
    assert err(a.cpu().numpy(), ref, mag) <= 6e-7


def test_x6_two_layers_in_one_launch_equal_two_launches():
    from faster_rcnn_amd import ops
    rs = np.random.RandomState(4)
    x = torch.from_numpy(rs.randn(1, 38, 63, 256).astype(np.float32)).cuda()
    w1 = (rs.randn(1, 1, 256, 128) * 0.06).astype(np.float32)
    w2 = (rs.randn(1, 1, 256, 512) * 0.06).astype(np.float32)
    s = lambda c: ((1 + 0.1 * rs.randn(c)).astype(np.float32), (0.1 * rs.randn(c)).astype(np.float32))
    (s1, h1), (s2, h2) = s(128), s(512)
    both = ops.PackedConv(np.concatenate([w1, w2], axis=3), np.concatenate([s1, s2]), np.concatenate([h1, h2]))
    for tile in (71, 74, 76):
        y1, y2 = ops.conv2d_dual(x, both, 128, 1, "valid", "relu", None, tile=tile)
        a = ops.conv2d(x, ops.PackedConv(w1, s1, h1), 1, "valid", "relu", tile=tile)
        b = ops.conv2d(x, ops.PackedConv(w2, s2, h2), 1, "valid", None, tile=tile)
        assert torch.equal(y1, a) and torch.equal(y2, b)
    # a boundary that is not a tile boundary takes the per-column epilogue: 9 + 36 channels (the RPN output pair) padded to 64 k
    w3 = (rs.randn(1, 1, 256, 9) * 0.06).astype(np.float32); w4 = (rs.randn(1, 1, 256, 36) * 0.06).astype(np.float32)
    pair = ops.PackedConv(np.concatenate([w3, w4], axis=3), np.ones(45, np.float32), np.zeros(45, np.float32))
    y1, y2 = ops.conv2d_dual(x, pair, 9, 1, "valid", "sigmoid", None, tile=74)
    a = ops.conv2d(x, ops.PackedConv(w3, np.ones(9, np.float32), np.zeros(9, np.float32)), 1, "valid", "sigmoid", tile=74)
    b = ops.conv2d(x, ops.PackedConv(w4, np.ones(36, np.float32), np.zeros(36, np.float32)), 1, "valid", None, tile=74)
    assert torch.equal(y1, a) and torch.equal(y2, b)


def test_engine_policy_scope_and_refusals():
    from faster_rcnn_amd import _lib, ops
    rs = np.random.RandomState(5)
    big = torch.from_numpy(rs.randn(1, 96, 128, 64).astype(np.float32)).cuda()          # 12 288 rows
    small = torch.from_numpy(rs.randn(1, 20, 20, 64).astype(np.float32)).cuda()
    pc = ops.PackedConv((rs.randn(3, 3, 64, 128) * 0.05).astype(np.float32))
    ops.CONV_PROFILE = []
    try:
        ops.conv2d(big, pc, 1, "same")
        with ops.f32_engine("bf16x6"):
            ops.conv2d(big, pc, 1, "same")
            ops.conv2d(small, pc, 1, "same")                     # 14 tiles of 64x64, under X6_MIN_TILES: stays native
        ops.conv2d(big, pc, 1, "same")
        names = [r["kernel"] for r in ops.CONV_PROFILE]
    finally:
        ops.CONV_PROFILE = None
    assert [("x6" in n) for n in names] == [False, True, False, False], names
    # cin % 32 != 0 has no split form: an explicit tile code is not honoured silently
    odd = ops.PackedConv((rs.randn(1, 1, 48, 64) * 0.1).astype(np.float32))
    y = ops.conv2d(torch.from_numpy(rs.randn(1, 8, 8, 48).astype(np.float32)).cuda(), odd, 1, "valid", tile=71)
    assert y.shape == (1, 8, 8, 64)                              # (took the native generic kernel)
    with pytest.raises(_lib.FrcnnError):
        import ctypes
        d = ops._conv_desc((1, 8, 8, 48), 1, 1, 64, 1, "valid", 0, 0, 71)
        _lib.call("frcnn_conv2d_fwd_x6", ctypes.byref(d), y.data_ptr(), y.data_ptr(), None, None, None, None, y.data_ptr(), None, 0, None)


def test_refresh_x6_planes_equals_the_single_filter_pack():
    """frcnn_refresh_x6_planes: the three bf16 planes of many packed filters in one launch == frcnn_pack_conv_weights_x6 per
    filter, bit for bit; the pieces sum back to the f32 value exactly; malformed jobs are refused."""
    import ctypes
    from faster_rcnn_amd import _lib, ops
    rs = np.random.RandomState(21)
    shapes = [(3, 3, 64, 96), (1, 1, 256, 1024), (1, 1, 32, 8), (3, 3, 512, 512)] * 15          # 60 jobs: more than one table
    packs = [ops.PackedConv((rs.randn(*sh) * 0.05).astype(np.float32)) for sh in shapes]
    want = [pc.x6_planes().clone() for pc in packs]
    outs = [torch.full_like(w, 7.0) for w in want]
    jobs = (_lib.X6Job * len(packs))()
    for j, pc, o in zip(jobs, packs, outs):
        j.w_packed, j.planes_bf16, j.rows, j.kpad = pc.w.data_ptr(), o.data_ptr(), pc.w.shape[0], pc.w.shape[1]
    _lib.call("frcnn_refresh_x6_planes", jobs, len(packs), None)
    torch.cuda.synchronize()
    for o, w_, pc in zip(outs, want, packs):
        assert torch.equal(o.view(torch.int16), w_.view(torch.int16))
        assert torch.equal(o[0].float() + o[1].float() + o[2].float(), pc.w)
    jobs[1].kpad = 48
    with pytest.raises(_lib.FrcnnError):
        _lib.call("frcnn_refresh_x6_planes", jobs, 2, None)
    _lib.call("frcnn_refresh_x6_planes", None, 0, None)          # nothing to do is fine


def test_engine_tile_choice_is_what_the_library_reports():
    """frcnn_conv2d_x6_config: 64x64 tiles for 64-column and small-grid layers, the four-wave 128x64 tile for >= 1024 row tiles of 64
    columns, 128x128 / the sixteen-wave form otherwise; a two-layer launch whose boundary is not a multiple of 128 takes 64-wide tiles."""
    import ctypes
    from faster_rcnn_amd import _lib, ops
    lib = _lib.load()

    def cfg(shape, k, cout, n1=0, stride=1, padding="same"):
        d = ops._conv_desc(shape, k, k, cout, stride, padding, 0, 0, 0)
        return lib.frcnn_conv2d_x6_config(ctypes.byref(d), n1)
    assert cfg((1, 149, 249, 64), 3, 64) == 74                       # stage 2's 3x3: 290 row tiles of 128
    assert cfg((1, 600, 1000, 64), 3, 64) == 77                      # VGG16 conv1_2
    assert cfg((1, 75, 125, 128), 3, 128) == 74                      # under 256 tiles of 128x128
    assert cfg((300, 7, 7, 512), 1, 2048) == 71
    assert cfg((300, 7, 7, 512), 3, 512) == 76                       # long k, >= 200 tiles of 256x128
    assert cfg((1, 149, 249, 64), 1, 320, n1=64, padding="valid") == 74
    assert cfg((1, 38, 63, 1024), 1, 2560, n1=512, padding="valid") == 76
    d = ops._conv_desc((300, 7, 7, 512), 3, 3, 512, 1, "same", 0, 0, 73)
    assert lib.frcnn_conv2d_x6_config(ctypes.byref(d), 0) == 73      # an explicit code is returned as asked
    assert lib.frcnn_conv2d_x6_config(None, 0) < 0


def test_x6_split_k_small_grid_long_k():
    """rpn_conv1 / stage-4 shapes (2 394 rows, k = 9 216 / 2 304): the engine's 64x64 split-K form -- partial tiles summed in slice
    order by the last arriver -- against fp64, bitwise reproducible, tickets left zero (a second launch on the same workspace)."""
    from faster_rcnn_amd import ops
    rs = np.random.RandomState(8)
    for (h, w, cin, cout, k) in ((38, 63, 1024, 512, 3), (38, 63, 256, 256, 3), (19, 31, 2048, 192, 1), (56, 56, 512, 512, 3)):      # (the last: 3 136 rows, 100 tiles of 128x128 -> the eight-wave split-K form)
        x = rs.randn(1, h, w, cin).astype(np.float32)
        wt = (rs.randn(k, k, cin, cout) * np.sqrt(2.0 / (k * k * cin))).astype(np.float32)
        scale = (1 + 0.1 * rs.randn(cout)).astype(np.float32); shift = (0.1 * rs.randn(cout)).astype(np.float32)
        pc = ops.PackedConv(wt, scale, shift)
        xd = torch.from_numpy(x).cuda()
        ws = ops.ConvWorkspace()
        ops.CONV_PROFILE = []
        try:
            with ops.f32_engine("bf16x6"), ops.conv_workspace(ws):
                got = ops.conv2d(xd, pc, 1, "same", "relu")
                again = ops.conv2d(xd, pc, 1, "same", "relu", out=torch.full_like(got, 3.0))
            with ops.f32_engine("bf16x6"), ops.conv_workspace(ops.NO_SPLIT_K):
                plain = ops.conv2d(xd, pc, 1, "same", "relu")
            names = [r["kernel"] for r in ops.CONV_PROFILE]
        finally:
            ops.CONV_PROFILE = None
        assert names[0].endswith("split-K") and "x6" in names[0], names
        tiles64 = -(-(h * w) // 64) * -(-cout // 64)
        assert ("x6" in names[2]) == (tiles64 >= ops.X6_MIN_TILES), names      # without a workspace a grid under X6_MIN_TILES stays native
        ref, mag = ref_conv(x, wt, 1, "same", scale, shift, None, "relu")
        assert err(got.cpu().numpy(), ref, mag) <= 5e-7
        assert torch.equal(got, again)
        assert err(plain.cpu().numpy(), ref, mag) <= 5e-7
        assert not ws.buf[:16384].any().item()                   # tickets back to zero
