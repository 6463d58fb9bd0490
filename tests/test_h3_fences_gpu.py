"""Fences around the f16x3 engine (csrc/conv_h3.hip): what happens OUTSIDE the operating range the parity
tests walk -- channels of one tensor 2^30 apart, Inf / NaN inputs, a magnitude record that is not an upper bound (stale or wrong), a
record of an earlier pass.  Each case: the measured behaviour against fp64 / the native kernel, and the status word
(include/frcnn_hip.h FRCNN_H3_*) that makes a bad pass visible with its outputs."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

from tests.test_conv_x6_gpu import err, ref_conv      # noqa: E402


def _status(rec):
    return int(rec.view(torch.int32)[1].item())


def _spread_case(compensate, log2_spread=30.0, seed=5):
    """x (1,128,128,256): channel c scaled by 2^(-spread c / 255) -- the statistics a checkpoint's folded normalisation leaves in an
    activation tensor; ``compensate``: the 1x1 filter's input channel c scaled by the inverse, so every channel weighs the same in the sum."""
    rs = np.random.RandomState(seed)
    cin, cout = 256, 128
    ch = 2.0 ** (-log2_spread * np.arange(cin) / (cin - 1))
    x = (rs.randn(1, 128, 128, cin) * ch).astype(np.float32)
    wt = (rs.randn(1, 1, cin, cout) * 0.05).astype(np.float32)
    if compensate:
        wt = (wt * (1.0 / ch)[None, None, :, None]).astype(np.float32)
    return x, wt


def test_channel_spread_in_the_activations_alone_costs_nothing():
    """Activations whose channels lie 2^30 apart, ordinary weights: the small channels' lost bits are small in the sum too.  The engine's
    usual bar against fp64 holds, next to the native kernel and the exact bf16 split."""
    from faster_rcnn_amd import ops
    x, wt = _spread_case(False)
    pc = ops.PackedConv(wt)
    assert pc._h3_spread_log2 < 8
    xd = torch.from_numpy(x).cuda()
    ref, mag = ref_conv(x, wt, 1, "valid")
    with ops.conv_workspace(ops.NO_SPLIT_K):
        with ops.f32_engine("f16x3"):
            d = ops._conv_desc(tuple(xd.shape), 1, 1, 128, 1, "valid", 0, 0, 0)
            assert ops._split_engine(d, pc, 0) == "h3"               # the policy keeps it on the engine
            got = ops.conv2d(xd, pc, 1, "valid")
        nat = ops.conv2d(xd, pc, 1, "valid", tile=0)
        x6 = ops.conv2d(xd, pc, 1, "valid", tile=74)
    e_h3, e_nat, e_x6 = err(got.cpu().numpy(), ref, mag), err(nat.cpu().numpy(), ref, mag), err(x6.cpu().numpy(), ref, mag)
    print("activation spread 2^30: f16x3 %.3g native %.3g bf16x6 %.3g" % (e_h3, e_nat, e_x6))
    assert e_h3 <= max(1.5 * e_nat, 4e-7)


@pytest.mark.parametrize("log2_spread", [30.0, 44.0])
def test_channel_spread_compensated_by_the_filter_is_routed_to_the_exact_split(log2_spread):
    """The adversarial form: the filter's input channels carry the INVERSE scales (the spread sits inside the filter tensor too), so the
    channels whose activations lost bits under the one-scale-per-tensor rule weigh as much as any other.  Forced onto the engine (an
    explicit tile code) the error against fp64 is measured -- at 2^30 fp16's subnormals and the low plane still hold ~19 bits and the
    sum is as good as the native kernel's; at 2^44 the smallest channels are down to a few bits and it shows -- while the POLICY
    (ops.H3_MAX_SPREAD_LOG2, from max|w| per input channel at lowering) sends the layer to the exact bf16 split, which holds the bar
    at either spread."""
    from faster_rcnn_amd import ops
    x, wt = _spread_case(True, log2_spread)
    pc = ops.PackedConv(wt)
    assert pc._h3_spread_log2 > log2_spread - 1
    xd = torch.from_numpy(x).cuda()
    ref, mag = ref_conv(x, wt, 1, "valid")
    with ops.conv_workspace(ops.NO_SPLIT_K):
        forced = ops.conv2d(xd, pc, 1, "valid", tile=84)
        with ops.f32_engine("f16x3"):
            d = ops._conv_desc(tuple(xd.shape), 1, 1, 128, 1, "valid", 0, 0, 0)
            assert ops._split_engine(d, pc, 0) == "x6"
            routed = ops.conv2d(xd, pc, 1, "valid")
            assert not ops.conv_accepts_planes(tuple(xd.shape), pc)
        nat = ops.conv2d(xd, pc, 1, "valid", tile=0)
    e_forced, e_routed, e_nat = err(forced.cpu().numpy(), ref, mag), err(routed.cpu().numpy(), ref, mag), err(nat.cpu().numpy(), ref, mag)
    print("compensated spread 2^%d: f16x3 forced %.3g, routed (bf16x6) %.3g, native %.3g" % (log2_spread, e_forced, e_routed, e_nat))
    assert e_routed <= max(1.5 * e_nat, 4e-7)
    assert _status(forced._amax) == 0 and torch.isfinite(forced).all()      # no fence trips: this is precision, not range
    if log2_spread <= 30.0:
        assert e_forced <= max(1.5 * e_nat, 4e-7)


def test_nan_and_inf_inputs_are_reported_not_carried():
    """The native kernel carries a NaN / Inf element through its receptive field (9 x 128 outputs here).  This engine cannot promise that
    -- one power of two cannot serve a tensor with an infinity in it, and under MODE.FP16_OVFL the conversion clamps a NaN like an
    overflow (measured on gfx950) -- so it REPORTS instead: the status word of the input's record is non-zero (NaN: clamped =
    FRCNN_H3_SATURATED; Inf: FRCNN_H3_NONFINITE), the outputs are finite and not to be used.  A documented, tested difference."""
    from faster_rcnn_amd import _lib, ops
    rs = np.random.RandomState(7)
    x = rs.randn(1, 24, 24, 64).astype(np.float32)
    wt = (rs.randn(3, 3, 64, 128) * 0.05).astype(np.float32)
    pc = ops.PackedConv(wt)
    for tile in (84, 81, 86):
        for bad in (np.nan, np.inf, -np.inf):
            xb = x.copy()
            xb[0, 10, 11, 5] = bad
            xd = torch.from_numpy(xb).cuda()
            rec = ops.amax_of(xd)
            xd._amax = rec
            with ops.conv_workspace(ops.NO_SPLIT_K):
                got = ops.conv2d(xd, pc, 1, "same", tile=tile)
                nat = ops.conv2d(xd, pc, 1, "same", tile=0)
            n = nat.cpu().numpy()
            assert (~np.isfinite(n)).sum() == 9 * 128                    # the native kernel: the receptive field
            st = _status(rec)
            print(tile, bad, "status", st, "finite outputs", bool(torch.isfinite(got).all()))
            if np.isnan(bad):
                assert np.isfinite(float(rec.max())) and st & _lib.H3_SATURATED
            else:
                assert np.isinf(float(rec.max())) and st & _lib.H3_NONFINITE
    # clean data, same shapes: nothing trips
    xd = torch.from_numpy(x).cuda()
    rec = ops.amax_of(xd)
    xd._amax = rec
    ops.conv2d(xd, pc, 1, "same", tile=84)
    assert _status(rec) == 0


@pytest.mark.parametrize("tile", [84, 81, 86])
def test_an_underestimated_record_is_clamped_and_flagged(tile):
    """A record that is NOT an upper bound.  Values that land in [2^15, 65504) after scaling still fit fp16: the result is right and the
    status word says FRCNN_H3_UNDER.  Under by 2^4 and more the conversion would overflow -- it is clamped to +-65504 (MODE.FP16_OVFL)
    instead of turning into Inf / NaN: a finite, wrong result and FRCNN_H3_UNDER | FRCNN_H3_SATURATED.  A zeroed record (what a stale arena
    slot looks like) under non-zero data: the same two bits."""
    from faster_rcnn_amd import _lib, ops
    rs = np.random.RandomState(8)
    x = rs.randn(1, 40, 52, 64).astype(np.float32)
    wt = (rs.randn(3, 3, 64, 128) * 0.05).astype(np.float32)
    pc = ops.PackedConv(wt)
    xd = torch.from_numpy(x).cuda()
    ref, mag = ref_conv(x, wt, 1, "same")
    true_max = float(np.abs(x).max())
    k = int(np.floor(np.log2(true_max)))

    def run(bound):
        rec = torch.zeros(_lib.load().frcnn_amax_record_floats(), dtype=torch.float32, device="cuda")
        rec[0] = bound
        xd._amax = rec
        with ops.conv_workspace(ops.NO_SPLIT_K):
            y = ops.conv2d(xd, pc, 1, "same", tile=tile)
        return y.cpu().numpy(), _status(rec)

    y, st = run(true_max)
    assert st == 0 and err(y, ref, mag) <= 5e-7
    y, st = run(2.0 ** (k + 9))                                       # 2^8 too large: nothing to report, nothing lost
    assert st == 0 and err(y, ref, mag) <= 5e-7
    ratio = true_max / 2.0 ** (k - 1)                                 # a bound of 2^(k-1) scales to 2^14: the largest value lands at ratio * 2^14, ratio in [2, 4)
    y, st = run(2.0 ** (k - 1))
    if ratio < 3.99:
        assert st == _lib.H3_UNDER and err(y, ref, mag) <= 5e-7       # flagged, and still right
    else:
        assert st & _lib.H3_UNDER
    y, st = run(2.0 ** (k - 4))                                       # the largest values at >= 2^18: clamped
    assert st == (_lib.H3_UNDER | _lib.H3_SATURATED) and np.isfinite(y).all() and err(y, ref, mag) > 1e-4
    y, st = run(0.0)
    assert st == (_lib.H3_UNDER | _lib.H3_SATURATED) and np.isfinite(y).all()


def test_a_record_of_an_earlier_pass_is_measured_again():
    """records live in a per-pass arena that the next amax_begin() zeroes; a tensor that outlives its pass (a cached conv map,
    an ``out=`` buffer) must not be read under the zeroed / re-used record.  amax_of sees the arena's generation and measures again;
    a native launch into an ``out=`` buffer drops the buffer's old record."""
    from faster_rcnn_amd import ops
    rs = np.random.RandomState(9)
    x = rs.randn(1, 40, 52, 64).astype(np.float32)
    wt = (rs.randn(1, 1, 64, 128) * 0.1).astype(np.float32)
    wt2 = (rs.randn(3, 3, 128, 128) * 0.03).astype(np.float32)
    pc, pc2 = ops.PackedConv(wt), ops.PackedConv(wt2)
    xd = torch.from_numpy(x).cuda()
    arena = ops.AmaxArena(16)
    with ops.amax_arena(arena), ops.conv_workspace(ops.NO_SPLIT_K):
        ops.amax_begin()
        mid = ops.conv2d(xd, pc, 1, "valid", "relu", tile=84)            # carries a record of pass 1
        want = ops.conv2d(mid, pc2, 1, "same", tile=84).clone()
        assert int(arena.status().item()) == 0
        ops.amax_begin()                                                  # pass 2: pass 1's records are cleared
        measured0 = ops.AMAX_MEASURED
        again = ops.conv2d(mid, pc2, 1, "same", tile=84)                  # `mid` outlived its pass
        assert ops.AMAX_MEASURED == measured0 + 1
        assert torch.equal(again, want)
        assert int(arena.status().item()) == 0
        # an out= buffer on the native path: the record it carried is dropped, not kept
        buf = ops.conv2d(xd, pc, 1, "valid", "relu", tile=84)
        assert buf._amax is not None
        ops.conv2d(xd * 1000.0, pc, 1, "valid", "relu", tile=2, out=buf)
        assert getattr(buf, "_amax", None) is None
        y = ops.conv2d(buf, pc2, 1, "same", tile=84)
        assert torch.isfinite(y).all() and int(arena.status().item()) == 0


def test_pass_status_travels_with_the_detections():
    """pipeline.InferencePipeline puts the pass's status word into det_packed[2]; a clean f16x3 pass reads 0, and entry.DetectionEntry raises
    on anything else."""
    from faster_rcnn_amd import ops, resnet, util
    from faster_rcnn_amd.pipeline import InferencePipeline
    from faster_rcnn_amd.weights import synthetic_resnet
    anchors = util.get_anchors([128, 256, 512])
    w = synthetic_resnet(50, anchors_per_loc=len(anchors), num_classes=21, seed=3)
    base = resnet.resnet50_base(weights=w)
    rpn = resnet.resnet50_rpn(base, include_conv=True, anchors_per_loc=len(anchors))
    det = resnet.resnet50_classifier(300, 21, base_model=None, weights=w)
    pipe = InferencePipeline(rpn, det, anchors)
    pipe.capture(320, 480, f32_engine="f16x3")
    rs = np.random.RandomState(4)
    img = (rs.randint(0, 256, (1, 320, 480, 3)).astype(np.float32) - 110.0)
    out = pipe.replay(torch.from_numpy(img).cuda())
    torch.cuda.synchronize()
    assert "h3_status" in out and int(out["h3_status"].item()) == 0 and int(out["det_packed"][2].item()) == 0
    bad = img.copy()
    bad[0, 100, 100, 1] = np.inf
    out = pipe.replay(torch.from_numpy(bad).cuda())
    torch.cuda.synchronize()
    assert int(out["det_packed"][2].item()) & 4
    out = pipe.replay(torch.from_numpy(img).cuda())                    # the word is per pass: clean again
    torch.cuda.synchronize()
    assert int(out["det_packed"][2].item()) == 0
    pipe.close()
