"""GPU parity of the MFMA implicit-GEMM conv engine vs the torch-CPU restatement.
fp32 tolerance: |got - want_f64| <= 1e-4 * max(1, |want|)  (north_star: fp32 within 1e-4)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def ops():
    from faster_rcnn_amd import ops as o
    return o


def check(got, want64, tol=1e-4):
    got = got.cpu().double()
    err = (got - want64).abs() / want64.abs().clamp(min=1.0)
    assert err.max().item() <= tol, err.max().item()


CASES = [
    # n, h, w, cin, cout, k, stride, padding, act, residual, tile
    (1, 20, 31, 64, 64, 1, 1, "valid", "relu", False, 0),
    (1, 20, 31, 64, 256, 1, 1, "valid", None, True, 0),
    (1, 21, 33, 64, 64, 3, 1, "same", "relu", False, 0),
    (1, 37, 50, 256, 128, 1, 2, "valid", "relu", False, 0),       # stride-2 1x1 VALID
    (1, 15, 22, 128, 128, 3, 1, "same", "relu", False, 1),
    (1, 15, 22, 128, 128, 3, 1, "same", "relu", False, 2),
    (1, 15, 22, 128, 96, 3, 1, "same", "relu", False, 3),         # cout not a tile multiple
    (1, 15, 22, 128, 256, 3, 1, "same", "relu", True, 4),
    (1, 15, 22, 128, 128, 3, 1, "same", "relu", False, 11),       # v2 kernels (buffer loads, pinned interleave)
    (1, 15, 22, 128, 128, 3, 1, "same", "relu", True, 12),
    (1, 15, 22, 128, 96, 3, 1, "same", "relu", False, 13),
    (1, 15, 22, 128, 256, 3, 1, "same", "relu", True, 14),
    (1, 37, 50, 256, 128, 1, 2, "valid", "relu", False, 12),
    (2, 9, 11, 64, 64, 3, 2, "same", None, False, 13),
    (3, 7, 7, 512, 512, 3, 1, "same", "relu", False, 11),
    (1, 15, 22, 128, 128, 3, 1, "same", "relu", True, 21),        # v2 late-store variant
    (1, 15, 22, 128, 96, 3, 1, "same", "relu", False, 22),
    (1, 20, 31, 32, 64, 1, 1, "valid", "relu", False, 21),
    (1, 20, 31, 64, 64, 1, 1, "valid", "relu", False, 22),
    (1, 15, 22, 128, 96, 3, 1, "same", "relu", True, 23),         # mid-chunk-barrier main loop: 64x64 (residual prefetched before the loop)
    (1, 20, 31, 32, 64, 1, 1, "valid", "relu", False, 23),        #   one k-chunk (the prologue alone over-reads four)
    (1, 20, 31, 64, 64, 1, 1, "valid", None, True, 23),           #   two chunks
    (1, 20, 31, 96, 72, 1, 1, "valid", "relu", True, 23),         #   three chunks (odd count: the peeled tail), cout % 64 != 0
    (1, 21, 33, 64, 200, 3, 1, "same", "relu", True, 24),         #   64x128 tile
    (2, 9, 11, 64, 64, 3, 2, "same", None, True, 25),             #   128x64 tile, stride 2
    (1, 15, 22, 128, 256, 3, 1, "same", "relu", True, 26),        #   128x128 tile (one staging set)
    (3, 7, 7, 512, 512, 3, 1, "same", "relu", False, 26),
    (1, 15, 22, 128, 96, 3, 1, "same", "relu", True, 323),        #   split-K, 3 slices of 12 chunks
    (2, 9, 11, 64, 64, 3, 2, "same", "relu", False, 1823),        #   one chunk per slice
    (1, 38, 63, 512, 36, 1, 1, "valid", None, False, 423),        #   cout = 36: 16-byte epilogue with a partial column tile
    (1, 38, 63, 512, 9, 1, 1, "valid", "sigmoid", False, 423),    #   cout = 9: the 4-byte epilogue fallback
    (1, 15, 22, 128, 128, 3, 1, "same", "relu", True, 41),        # 8-wave 128x128 variants
    (1, 15, 22, 128, 200, 3, 1, "same", "relu", False, 42),
    (2, 9, 11, 64, 64, 3, 2, "same", None, False, 43),
    (3, 7, 7, 512, 512, 3, 1, "same", "relu", False, 41),
    (1, 15, 22, 128, 96, 3, 1, "same", "relu", True, 322),        # split-K, 3 slices: slices start mid-filter (36 chunks)
    (1, 15, 22, 128, 128, 3, 1, "same", None, False, 722),        # 7 slices of 36 chunks: uneven slice lengths
    (2, 9, 11, 64, 64, 3, 2, "same", "relu", False, 1822),        # as many slices as chunks (18): one chunk each
    (1, 38, 63, 512, 9, 1, 1, "valid", "sigmoid", False, 422),    # rpn_out_cls, 4 slices
    (5, 1, 1, 2048, 101, 1, 1, "valid", None, False, 1622),       # dense, 16 slices, 5 valid rows of 64
    (1, 38, 63, 256, 256, 3, 1, "same", "relu", True, 0),         # stage-4 3x3 at full size: auto picks split-K
    (1, 38, 63, 1024, 256, 1, 1, "valid", "relu", False, 0),      # stage-4 1x1 reduce
    (1, 38, 63, 256, 256, 3, 1, "same", "relu", True, 122),       # same shape, split-K forced off
    (1, 20, 31, 32, 64, 1, 1, "valid", "relu", False, 11),        # single k-chunk
    (1, 20, 31, 64, 64, 1, 1, "valid", "relu", False, 12),        # two k-chunks
    (3, 7, 7, 512, 512, 3, 1, "same", "relu", False, 0),          # head: RoIs as batch
    (1, 38, 63, 512, 9, 1, 1, "valid", "sigmoid", False, 0),      # rpn_out_cls
    (1, 38, 63, 512, 36, 1, 1, "valid", None, False, 0),          # rpn_out_bbreg
    (5, 1, 1, 2048, 101, 1, 1, "valid", None, False, 0),          # dense
    (1, 61, 83, 3, 64, 7, 2, "same", "relu", False, 0),           # stem (odd size)
    (1, 60, 80, 3, 64, 7, 2, "same", "relu", False, 0),           # stem (even size: pad 2/3)
    (1, 24, 30, 3, 64, 3, 1, "same", "relu", False, 0),           # vgg block1_conv1
    (2, 9, 11, 64, 64, 3, 2, "same", None, False, 0),             # SAME with stride 2
    (1, 20, 23, 32, 64, 7, 1, "same", "relu", False, 0),          # 49 taps: beyond the v2 kernels' 32-bit tap mask
    (1, 20, 23, 64, 64, 7, 2, "same", None, True, 22),
    (1, 14, 15, 32, 32, 6, 1, "valid", None, False, 41),
    (2, 60, 80, 3, 64, 7, 2, "same", "relu", False, 0),           # stem kernel, two images (49 taps -> 7 chunks of 8)
    (1, 24, 30, 3, 96, 3, 1, "valid", None, True, 0),             # stem kernel: cout not a tile multiple, residual, 7 dead taps
    (1, 9, 9, 3, 32, 1, 1, "valid", None, False, 0),              # stem kernel: one tap
    (1, 61, 83, 4, 64, 7, 2, "same", "relu", False, 0),           # other small cin: the generic gather kernel
    (1, 24, 30, 5, 96, 3, 1, "valid", None, True, 0),
    (1, 24, 30, 20, 64, 3, 1, "same", "relu", False, 2),
]


@pytest.mark.parametrize("case", CASES)
def test_conv2d(ops, case):
    from oracle import keras_ref
    n, h, w, cin, cout, k, stride, padding, act, use_res, tile = case
    rs = np.random.RandomState(hash(case) % (2 ** 31))
    x = rs.randn(n, h, w, cin).astype(np.float32)
    wt = (rs.randn(k, k, cin, cout) * np.sqrt(2.0 / (k * k * cin))).astype(np.float32)
    scale = (1 + 0.1 * rs.randn(cout)).astype(np.float32)
    shift = (0.1 * rs.randn(cout)).astype(np.float32)
    want = keras_ref.conv2d(x, wt, None, stride, padding, dtype=torch.float64)
    want = want * torch.from_numpy(scale).double() + torch.from_numpy(shift).double()
    res = None
    if use_res:
        res = rs.randn(*want.shape).astype(np.float32)
        want = want + torch.from_numpy(res).double()
    if act == "relu":
        want = want.clamp(min=0)
    elif act == "sigmoid":
        want = torch.sigmoid(want)
    pc = ops.PackedConv(wt, scale, shift)
    got = ops.conv2d(torch.from_numpy(x).cuda(), pc, stride, padding, act,
                     None if res is None else torch.from_numpy(res).cuda(), tile=tile)
    assert tuple(got.shape) == tuple(want.shape)
    check(got, want)


@pytest.mark.parametrize("case", [
    # n, h, w, cin, cout, k, stride, padding, act, tile
    (1, 38, 63, 256, 1024, 1, 1, "valid", "relu", 23),           # stage-4 2c: residual prefetched before the main loop
    (1, 38, 63, 256, 1024, 1, 1, "valid", "relu", 22),
    (1, 21, 33, 64, 200, 3, 1, "same", None, 24),                # partial column tile (200 = 128 + 72)
    (3, 7, 7, 512, 512, 3, 1, "same", "relu", 26),               # 128x128: sixteen passes in groups of four
    (1, 38, 63, 256, 256, 3, 1, "same", "relu", 323),            # split-K reducer
    (1, 38, 63, 512, 36, 1, 1, "valid", "sigmoid", 23),
])
def test_vector_epilogue_is_bitwise_the_scalar_epilogue(ops, case):
    """The 16-byte epilogue (tile through LDS, b128 residual / mask loads and stores) performs the scalar epilogue's
    arithmetic per element in the same order.  An output tensor that is NOT 16-byte aligned forces the 4-byte
    epilogue (frcnn_conv2d_fwd_ws checks the alignment), so the two are compared bit for bit, residual included."""
    n, h, w, cin, cout, k, stride, padding, act, tile = case
    rs = np.random.RandomState(abs(hash(case)) % (2 ** 31))
    x = torch.from_numpy(rs.randn(n, h, w, cin).astype(np.float32)).cuda()
    wt = (rs.randn(k, k, cin, cout) * np.sqrt(2.0 / (k * k * cin))).astype(np.float32)
    pc = ops.PackedConv(wt, (1 + 0.1 * rs.randn(cout)).astype(np.float32), (0.1 * rs.randn(cout)).astype(np.float32))
    vec = ops.conv2d(x, pc, stride, padding, act, tile=tile)
    res = torch.from_numpy(rs.randn(*vec.shape).astype(np.float32)).cuda()
    vec_r = ops.conv2d(x, pc, stride, padding, act, res, tile=tile)
    odd = torch.empty(vec.numel() + 4, dtype=torch.float32, device="cuda")[1:1 + vec.numel()].view(vec.shape)
    assert odd.data_ptr() % 16 == 4 and odd.is_contiguous()
    assert torch.equal(ops.conv2d(x, pc, stride, padding, act, out=odd, tile=tile), vec)
    assert torch.equal(ops.conv2d(x, pc, stride, padding, act, res, out=odd, tile=tile), vec_r)
    assert not torch.equal(vec, vec_r)


def test_conv_identity_asymmetric(ops):
    """A = I against an ASYMMETRIC filter catches a transposed C write (cdna guide s3)."""
    cin = cout = 64
    x = np.zeros((1, 1, 64, cin), np.float32)
    x[0, 0, np.arange(64), np.arange(64)] = 1.0                # pixel m has a one in channel m
    wt = (np.arange(cin)[:, None] * 100 + np.arange(cout)[None, :]).astype(np.float32).reshape(1, 1, cin, cout)
    got = ops.conv2d(torch.from_numpy(x).cuda(), ops.PackedConv(wt)).cpu().numpy()[0, 0]
    assert np.array_equal(got, wt[0, 0])


def test_pool_and_softmax(ops):
    from oracle import keras_ref
    rs = np.random.RandomState(1)
    x = rs.randn(2, 17, 23, 64).astype(np.float32)
    for k, s, mx in ((3, 2, True), (2, 2, True), (7, 7, False)):
        want = keras_ref.pool2d(torch.from_numpy(x), k, s, mx)
        got = ops.pool2d(torch.from_numpy(x).cuda(), k, s, mx).cpu()
        assert got.shape == want.shape
        assert torch.allclose(got, want, rtol=0, atol=1e-6 if not mx else 0)
    z = rs.randn(300, 101).astype(np.float32) * 3
    got = ops.softmax_rows(torch.from_numpy(z).cuda(), 21).cpu()
    want = torch.softmax(torch.from_numpy(z[:, :21]).double(), dim=1)
    assert (got.double() - want).abs().max().item() < 1e-6


def test_split_k_workspace_reuse_and_determinism(ops):
    """One workspace serves launch after launch (tickets return to zero), results do not depend on what
    the slabs held before, and the fixed slice order makes two runs bitwise equal."""
    rs = np.random.RandomState(7)
    pcs, xs = [], []
    for cin, cout, k in [(256, 256, 3), (1024, 256, 1), (512, 36, 1)]:
        wt = (rs.randn(k, k, cin, cout) * np.sqrt(2.0 / (k * k * cin))).astype(np.float32)
        pcs.append(ops.PackedConv(wt, np.ones(cout, np.float32), np.zeros(cout, np.float32)))
        xs.append(torch.from_numpy(rs.randn(1, 38, 63, cin).astype(np.float32)).cuda())
    ws = ops.ConvWorkspace()
    with ops.conv_workspace(ws):
        first = [ops.conv2d(x, pc, 1, "same", "relu").clone() for x, pc in zip(xs, pcs)]
        assert ws.buf is not None, "these shapes must take the split-K launch"
        for _ in range(20):
            again = [ops.conv2d(x, pc, 1, "same", "relu") for x, pc in zip(xs, pcs)]
        torch.cuda.synchronize()
        for a, b in zip(first, again):
            assert torch.equal(a, b)
        assert int(ws.buf[:16384].sum().item()) == 0            # tickets left zero
    for x, pc, a in zip(xs, pcs, first):                          # vs the plain launch (different summation order)
        plain = ops.conv2d(x, pc, 1, "same", "relu", tile=122)
        assert (plain - a).abs().max().item() <= 2e-5 * max(1.0, plain.abs().max().item())


@pytest.mark.parametrize("case", [
    # n, h, w, cin, cout, k, padding, tile
    (300, 7, 7, 64, 128, 3, "same", 0),          # the detector head's geometry: tiles span 1-2 positions, border taps skipped
    (300, 7, 7, 64, 128, 3, "same", 21),
    (300, 7, 7, 64, 96, 3, "same", 22),
    (300, 7, 7, 64, 96, 3, "same", 23),
    (300, 7, 7, 64, 128, 3, "same", 26),
    (300, 7, 7, 64, 128, 3, "same", 323),
    (300, 7, 7, 64, 128, 3, "same", 42),
    (300, 7, 7, 64, 128, 3, "same", 322),        # split-K over the compacted (channel group, needed tap) sequence
    (64, 7, 7, 128, 64, 3, "same", 0),
    (37, 5, 9, 32, 40, 3, "same", 12),
    (300, 7, 7, 256, 64, 1, "valid", 0),         # 1x1: one tap, nothing to skip
    (3, 7, 7, 64, 64, 3, "same", 0),             # few images: a tile spans > 8 positions -> all taps kept
    (50, 6, 6, 32, 64, 5, "same", 0),            # 25 taps
    (20, 9, 9, 64, 64, 3, "valid", 0),
])
def test_conv2d_position_major_layout(ops, case):
    """layout=1 ([h][w][n][c] tensors, taps that only meet padding skipped) is the SAME arithmetic as the
    NHWC launch: skipped chunks would add exact zeros, so the two results are bitwise equal."""
    n, h, w, cin, cout, k, padding, tile = case
    rs = np.random.RandomState(abs(hash(case)) % (2 ** 31))
    x = torch.from_numpy(rs.randn(n, h, w, cin).astype(np.float32)).cuda()
    wt = (rs.randn(k, k, cin, cout) * np.sqrt(2.0 / (k * k * cin))).astype(np.float32)
    pc = ops.PackedConv(wt, (1 + 0.1 * rs.randn(cout)).astype(np.float32), (0.1 * rs.randn(cout)).astype(np.float32))
    no_split = 100 + tile % 100                              # same tile code, split-K forced off
    want = ops.conv2d(x, pc, 1, padding, "relu", tile=no_split)
    res = torch.from_numpy(rs.randn(*want.shape).astype(np.float32)).cuda()
    want = ops.conv2d(x, pc, 1, padding, "relu", res, tile=no_split)
    xp = x.permute(1, 2, 0, 3).contiguous()
    got = ops.conv2d(xp, pc, 1, padding, "relu", res.permute(1, 2, 0, 3).contiguous(), tile=tile if tile >= 100 else no_split, layout=1)
    assert got.shape == (want.shape[1], want.shape[2], n, cout)
    got_nhwc = got.permute(2, 0, 1, 3)
    if tile >= 100:                                          # split-K changes the summation order
        assert (got_nhwc - want).abs().max().item() <= 2e-5 * max(1.0, want.abs().max().item())
    else:
        assert torch.equal(got_nhwc, want)


@pytest.mark.parametrize("case", [
    # n, h, w, cin, cout, k, padding, tile, layout
    (1, 48, 48, 128, 1024, 3, "same", 62, 0),        # 576 64x64 tiles on 1024 workgroups: ranges of 20.25 chunks, tiles of 36
    (2, 48, 48, 128, 1024, 3, "same", 61, 0),        # 288 128x128 tiles on 512 workgroups
    (150, 7, 7, 128, 512, 3, "same", 62, 1),         # position-major: tiles of 16..36 chunks (padding-only taps skipped)
    (300, 7, 7, 128, 512, 3, "same", 61, 1),
    (300, 7, 7, 1024, 500, 1, "valid", 61, 1),       # 1x1, cout tail inside the last column tile
    (3, 50, 47, 512, 520, 1, "valid", 62, 0),        # ragged rows and columns (999 tiles, 16 chunks each)
])
def test_conv2d_balanced_launch(ops, case):
    """The balanced (stream-K) launch form: workgroups take equal runs of k-chunks across tile boundaries, tiles met by
    several runs are summed from partial slots in slot order.  Same values as the plain launch up to the regrouped
    f32 sums, bit-identical from run to run, and the workspace tickets return to zero."""
    import ctypes
    from faster_rcnn_amd import _lib
    n, h, w, cin, cout, k, padding, tile, layout = case
    rs = np.random.RandomState(abs(hash(case)) % (2 ** 31))
    x = torch.from_numpy(rs.randn(n, h, w, cin).astype(np.float32)).cuda()
    wt = (rs.randn(k, k, cin, cout) * np.sqrt(2.0 / (k * k * cin))).astype(np.float32)
    pc = ops.PackedConv(wt, (1 + 0.1 * rs.randn(cout)).astype(np.float32), (0.1 * rs.randn(cout)).astype(np.float32))
    plain = ops.conv2d(x, pc, 1, padding, None, tile=100 + tile - 40)            # 121 / 122: the same tile, one workgroup per tile
    res = torch.from_numpy(rs.randn(*plain.shape).astype(np.float32)).cuda()
    want = ops.conv2d(x, pc, 1, padding, "relu", res, tile=100 + tile - 40)
    if layout:
        xin, rin = x.permute(1, 2, 0, 3).contiguous(), res.permute(1, 2, 0, 3).contiguous()
    else:
        xin, rin = x, res
    ho, wo = want.shape[1], want.shape[2]
    d = _lib.ConvDesc(n=n, h=h, w=w, cin=cin, cout=cout, kh=k, kw=k, stride=1, pad_top=(k - 1) // 2 if padding == "same" else 0,
                      pad_left=(k - 1) // 2 if padding == "same" else 0, ho=ho, wo=wo, act=1, ldy=0, ldres=0, tile=tile, layout=layout)
    assert _lib.load().frcnn_conv2d_config(ctypes.byref(d)) == tile               # the shape is eligible: the balanced kernel runs
    ws = ops.ConvWorkspace()
    with ops.conv_workspace(ws):
        got = ops.conv2d(xin, pc, 1, padding, "relu", rin, tile=tile, layout=layout)
        again = ops.conv2d(xin, pc, 1, padding, "relu", rin, tile=tile, layout=layout)
    assert ws.buf is not None and not ws.buf[:16384].any()                         # tickets back at zero
    assert torch.equal(got, again)
    g = got.permute(2, 0, 1, 3) if layout else got
    assert (g - want).abs().max().item() <= 2e-5 * max(1.0, want.abs().max().item())
    # slots hold stale partials: poison them
    ws.buf[16384:].view(torch.float32).fill_(float("nan"))
    with ops.conv_workspace(ws):
        third = ops.conv2d(xin, pc, 1, padding, "relu", rin, tile=tile, layout=layout)
    assert torch.equal(third, got)                                                  # every slot that is read was written first


def test_balanced_launch_is_chosen_for_the_head_shapes(ops):
    """Auto tile selection: alone on the chip, the detector head's 460-tile launches (300 RoIs) take the balanced form
    (given a workspace); short-k launches, launches that share the chip and "never split" descriptors do not."""
    import ctypes
    from faster_rcnn_amd import _lib
    cfg = lambda **kw: _lib.load().frcnn_conv2d_config(ctypes.byref(_lib.ConvDesc(stride=1, act=1, ldy=0, ldres=0, **kw)))
    head = dict(n=300, h=7, w=7, ho=7, wo=7, layout=1)
    assert cfg(cin=2048, cout=512, kh=1, kw=1, pad_top=0, pad_left=0, tile=0, **head) == 61       # 460 tiles of 128x128, 64 chunks
    assert cfg(cin=512, cout=512, kh=3, kw=3, pad_top=1, pad_left=1, tile=0, **head) == 61        # alone on the chip
    assert cfg(cin=512, cout=512, kh=3, kw=3, pad_top=1, pad_left=1, tile=50, **head) == 21       # beside other images' launches: plain
    assert cfg(cin=512, cout=2048, kh=1, kw=1, pad_top=0, pad_left=0, tile=0, **head) == 23       # 16 chunks: too short
    assert cfg(cin=512, cout=512, kh=3, kw=3, pad_top=1, pad_left=1, tile=100, **head) == 23      # "never split" turns it off


# ----------------------------------------------------------------------------------------------------------------
# Two layers on one input in ONE launch (frcnn_conv2d_fwd_dual): conv_block's branch2a + shortcut (resnet.py:218-241)
# and rpn_out_cls + rpn_out_bbreg (resnet.py:464-474).  The bar is bit-identity with the two single-layer launches.
DUAL_CASES = [
    # n, h, w, cin, n1, n2, k, stride, padding, act1, act2, tile, layout
    (1, 40, 50, 64, 64, 256, 1, 1, "valid", "relu", None, 0, 0),          # res2a: 64x64 tiles, boundary on a tile edge (16-byte epilogue)
    (1, 37, 50, 256, 128, 512, 1, 2, "valid", "relu", None, 0, 0),        # res3a: stride 2
    (1, 38, 63, 512, 256, 1024, 1, 1, "valid", "relu", None, 0, 0),       # res4a shapes
    (1, 38, 63, 1024, 512, 2048, 1, 1, "valid", None, None, 0, 0),        # the hoisted res5a pair on the conv4 map
    (1, 38, 63, 512, 9, 36, 1, 1, "valid", "sigmoid", None, 0, 0),        # RPN outputs: boundary INSIDE a tile (4-byte epilogue), split-K
    (1, 38, 63, 512, 18, 72, 1, 1, "valid", "sigmoid", None, 0, 0),       # 18 anchors (configs[3])
    (1, 21, 33, 64, 64, 72, 3, 1, "same", "relu", "relu", 23, 0),         # a 3x3 pair, second layer with a partial column tile
    (1, 21, 33, 64, 128, 128, 1, 1, "valid", "relu", None, 26, 0),        # 128x128 tiles, boundary on a tile edge
    (1, 21, 33, 64, 64, 192, 1, 1, "valid", "relu", None, 26, 0),         # 128x128 tiles, boundary inside a tile -> 4-byte epilogue
    (1, 21, 33, 64, 64, 192, 1, 1, "valid", "relu", None, 21, 0),
    (20, 7, 7, 1024, 512, 2048, 1, 1, "valid", "relu", None, 0, 1),       # position-major (the TimeDistributed res5a pair, reference order)
]


@pytest.mark.parametrize("case", DUAL_CASES, ids=[str(c) for c in DUAL_CASES])
@pytest.mark.parametrize("split_k", [False, True])
def test_dual_launch_equals_two_launches(ops, case, split_k):
    n, h, w, cin, n1, n2, k, stride, padding, act1, act2, tile, layout = case
    rs = np.random.RandomState(n1 + n2 + h)
    shape = (h, w, n, cin) if layout else (n, h, w, cin)
    x = torch.from_numpy(rs.randn(*shape).astype(np.float32)).cuda()
    wa = (rs.randn(k, k, cin, n1) / np.sqrt(k * k * cin)).astype(np.float32)
    wb = (rs.randn(k, k, cin, n2) / np.sqrt(k * k * cin)).astype(np.float32)
    sa, ha, sb, hb = (rs.rand(n1) + 0.5).astype(np.float32), rs.randn(n1).astype(np.float32), (rs.rand(n2) + 0.5).astype(np.float32), rs.randn(n2).astype(np.float32)
    pa, pb = ops.PackedConv(wa, sa, ha), ops.PackedConv(wb, sb, hb)
    pcat = ops.PackedConv(np.concatenate([wa, wb], axis=3), np.concatenate([sa, sb]), np.concatenate([ha, hb]))
    assert torch.equal(pcat.w, torch.cat([pa.w, pb.w]))                   # packed rows are per output channel
    ws = ops.ConvWorkspace() if split_k else ops.NO_SPLIT_K
    with ops.conv_workspace(ws):
        ya = ops.conv2d(x, pa, stride, padding, act1, tile=tile, layout=layout)
        yb = ops.conv2d(x, pb, stride, padding, act2, tile=tile, layout=layout)
        for _ in range(3):                                                # (repeat: split-K tickets must come back to zero)
            y1, y2 = ops.conv2d_dual(x, pcat, n1, stride, padding, act1, act2, layout, tile)
    torch.cuda.synchronize()
    assert y1.shape == ya.shape and y2.shape == yb.shape
    same_split = not split_k or (n1 + n2 <= 64)       # tiny grids: one column tile, the same slices either way
    if same_split:
        assert torch.equal(y1, ya) and torch.equal(y2, yb)
    else:
        # a single-layer launch on a small grid may have cut K into slices where the pair's larger grid does not:
        # same products, another summation tree
        for got, want in ((y1, ya), (y2, yb)):
            assert float(((got - want).abs() / want.abs().clamp(min=1)).max()) < 1e-5


def test_dual_rejects_bad_arguments(ops):
    from faster_rcnn_amd import _lib
    x = torch.zeros((1, 8, 8, 64), dtype=torch.float32, device="cuda")
    pc = ops.PackedConv(np.zeros((1, 1, 64, 96), np.float32), np.ones(96, np.float32), np.zeros(96, np.float32))
    with pytest.raises(AssertionError):
        ops.conv2d_dual(x, pc, 96)
    # raw C ABI: strided outputs and an undersized workspace are refused before any launch
    import ctypes
    d = _lib.ConvDesc(n=1, h=38, w=63, cin=512, cout=45, kh=1, kw=1, stride=1, pad_top=0, pad_left=0, ho=38, wo=63, act=0, ldy=0, ldres=0, tile=0, layout=0)
    need = _lib.load().frcnn_conv2d_dual_workspace_bytes(ctypes.byref(d))
    assert need > 0                                                        # 38 tiles, 16 chunks: split-K
    xs = torch.zeros((1, 38, 63, 512), dtype=torch.float32, device="cuda")
    pcs = ops.PackedConv(np.zeros((1, 1, 512, 45), np.float32), np.ones(45, np.float32), np.zeros(45, np.float32))
    y1, y2 = torch.empty((2394, 9), device="cuda"), torch.empty((2394, 36), device="cuda")
    small = torch.zeros(1024, dtype=torch.uint8, device="cuda")
    rc = _lib.load().frcnn_conv2d_fwd_dual(ctypes.byref(d), xs.data_ptr(), pcs.w.data_ptr(), pcs.scale.data_ptr(), pcs.shift.data_ptr(),
                                           y1.data_ptr(), 9, 2, y2.data_ptr(), 0, small.data_ptr(), small.numel(), None)
    assert rc == -2 and b"workspace" in _lib.load().frcnn_last_error()       # FRCNN_E_WORKSPACE
    d.ldy = 64
    rc = _lib.load().frcnn_conv2d_fwd_dual(ctypes.byref(d), xs.data_ptr(), pcs.w.data_ptr(), None, None, y1.data_ptr(), 9, 2, y2.data_ptr(), 0, None, 0, None)
    assert rc == -1                                                         # FRCNN_E_ARG: dense outputs only
    x48 = torch.zeros((1, 8, 8, 48), dtype=torch.float32, device="cuda")
    pc48 = ops.PackedConv(np.zeros((1, 1, 48, 96), np.float32), np.ones(96, np.float32), np.zeros(96, np.float32))
    with pytest.raises(_lib.FrcnnError):
        ops.conv2d_dual(x48, pc48, 32)                                    # cin % 32 != 0


def test_networks_with_paired_launches_equal_unpaired(ops):
    """ResNet-50 base + RPN head + hoisted detector head with the pairs in one launch == every layer on its own launch,
    bit for bit (plain launches), and to rounding with split-K on (the pair's grid is larger: other slice counts)."""
    from faster_rcnn_amd import nets, resnet, util
    from faster_rcnn_amd.weights import synthetic_resnet
    anchors = util.get_anchors([128, 256, 512])
    w = synthetic_resnet(50, anchors_per_loc=9, num_classes=21, seed=11)
    x = torch.from_numpy((np.random.RandomState(1).randint(0, 256, (1, 160, 224, 3)) - 110.0).astype(np.float32)).cuda()
    rois = torch.tensor([[0, 0, 5, 4], [2, 1, 9, 8], [3, 3, 13, 9], [0, 2, 4, 9]], dtype=torch.float32, device="cuda")

    def run(fuse, ws):
        nets.FUSE_PAIRS = fuse
        try:
            base = resnet.resnet50_base(weights=w)
            rpn = resnet.resnet50_rpn(base, include_conv=True, anchors_per_loc=9)
            det = resnet.resnet50_classifier(4, 21, weights=w)
            with ops.conv_workspace(ws):
                cls, reg, feat = rpn.forward_dev(x)
                dc, dr = det.forward_dev(feat, rois)
            torch.cuda.synchronize()
            return cls, reg, feat, dc, dr
        finally:
            nets.FUSE_PAIRS = True
    a = run(True, ops.NO_SPLIT_K)
    b = run(False, ops.NO_SPLIT_K)
    for s, t in zip(a, b):
        assert torch.equal(s, t)
    c = run(True, ops.ConvWorkspace())
    d = run(False, ops.ConvWorkspace())
    for s, t in zip(c, d):
        assert float(((s - t).abs() / t.abs().clamp(min=1)).max()) < 1e-5


def test_dual_launch_random_shapes(ops):
    """Randomised sweep of the two-layer launch against the two single-layer launches (plain launches: bit-identical):
    channel counts off and on tile boundaries, strides, kernel sizes, both row layouts, tile codes."""
    rs = np.random.RandomState(2024)
    for it in range(40):
        k = int(rs.choice([1, 1, 3]))
        stride = int(rs.choice([1, 2])) if k == 1 else 1
        padding = "valid" if k == 1 else "same"
        cin = int(rs.choice([32, 64, 96, 160]))
        n1 = int(rs.choice([4, 9, 36, 64, 72, 128, 200]))
        n2 = int(rs.choice([4, 16, 45, 64, 128, 136, 256]))
        layout = int(rs.rand() < 0.3)
        n, h, w = (int(rs.randint(3, 9)), 7, 7) if layout else (1, int(rs.randint(5, 40)), int(rs.randint(5, 50)))
        tile = int(rs.choice([0, 0, 23, 26, 21, 50]))
        act1, act2 = rs.choice([None, "relu", "sigmoid"]), rs.choice([None, "relu"])
        x = torch.from_numpy(rs.randn(*((h, w, n, cin) if layout else (n, h, w, cin))).astype(np.float32)).cuda()
        wa = (rs.randn(k, k, cin, n1) / np.sqrt(k * k * cin)).astype(np.float32)
        wb = (rs.randn(k, k, cin, n2) / np.sqrt(k * k * cin)).astype(np.float32)
        sa, ha, sb, hb = (rs.rand(n1) + 0.5).astype(np.float32), rs.randn(n1).astype(np.float32), (rs.rand(n2) + 0.5).astype(np.float32), rs.randn(n2).astype(np.float32)
        pa, pb = ops.PackedConv(wa, sa, ha), ops.PackedConv(wb, sb, hb)
        pcat = ops.PackedConv(np.concatenate([wa, wb], axis=3), np.concatenate([sa, sb]), np.concatenate([ha, hb]))
        with ops.conv_workspace(ops.NO_SPLIT_K):
            ya = ops.conv2d(x, pa, stride, padding, act1, tile=tile, layout=layout)
            yb = ops.conv2d(x, pb, stride, padding, act2, tile=tile, layout=layout)
            y1, y2 = ops.conv2d_dual(x, pcat, n1, stride, padding, act1, act2, layout, tile)
        torch.cuda.synchronize()
        case = (it, k, stride, cin, n1, n2, layout, n, h, w, tile, act1, act2)
        assert torch.equal(y1, ya) and torch.equal(y2, yb), case
