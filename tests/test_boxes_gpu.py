"""GPU parity: box-side HIP kernels (through the C ABI) vs the numpy oracle and the golden
vectors captured from the reference.  Integer / mask / index results must be bit exact."""
import numpy as np
import pytest

from tests import synth

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def ops():
    from faster_rcnn_amd import ops as o
    return o


@pytest.fixture(scope="module")
def ref():
    from oracle import np_ref
    return np_ref


def dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a)).cuda()
    return t if dtype is None else t.to(dtype)


def dirty(*nbytes):
    """Leave stale non-zero bytes in the caching allocator's free blocks of these sizes: the calls below get their
    outputs from torch.empty and must define every element themselves."""
    import torch
    keep = [torch.full((max(int(n), 1),), 0x5A, dtype=torch.uint8, device="cuda") for n in nbytes]
    torch.cuda.synchronize()
    del keep


def test_anchors(ops, ref, golden):
    a9, a18 = golden["anchors9"], golden["anchors18"]
    for rows, cols, anc in ((3, 4, a9), (38, 63, a9), (38, 94, a18), (1, 1, a18)):
        got = ops.anchors_image(rows, cols, anc, 16).cpu().numpy()
        assert np.array_equal(got, ref.anchors_image(rows, cols, anc, 16))
        got = ops.anchors_conv(rows, cols, anc // 16).cpu().numpy()
        assert np.array_equal(got, ref.anchors_conv(rows, cols, anc // 16))
    assert np.array_equal(ops.anchors_image(38, 63, a9, 16).cpu().numpy(), golden["anc_img_c2_i16"].astype(np.float32))


def test_cross_ious(ops, ref, golden):
    c2 = golden["anc_img_c2_i16"].astype(np.float32)
    got = ops.cross_ious(dev(c2), dev(golden["gt5"])).cpu().numpy()
    assert np.array_equal(got, golden["iou_c2_gt5"])
    got = ops.cross_ious(dev(golden["iou_i16_boxes"]), dev(golden["iou_i16_gt"])).cpu().numpy()
    assert np.array_equal(got, golden["iou_i16"])
    # random float boxes incl. degenerate and disjoint ones
    rs = np.random.RandomState(11)
    b1 = (rs.rand(5000, 4) * 500).astype(np.float32)
    b1[:, 2:] += b1[:, :2]
    b2 = (rs.rand(37, 4) * 500).astype(np.float32)
    b2[:, 2:] += b2[:, :2]
    got = ops.cross_ious(dev(b1), dev(b2)).cpu().numpy()
    assert np.array_equal(got, ref.cross_ious(b1, b2))
    # empty sides
    assert ops.cross_ious(dev(b1[:0]), dev(b2)).shape == (0, 37)
    assert ops.cross_ious(dev(b1), dev(b2[:0])).shape == (5000, 0)


RPN_CASES = {
    "rpn_img5_vgg": ("img5_gt", 0, "vgg", "anchors9"),
    "rpn_img5rs_vgg": ("img5_rs_gt", 2, "vgg", "anchors9"),
    "rpn_img5rs_res": ("img5_rs_gt", 2, "res", "anchors9"),
    "rpn_c2": ("gt5", (1000, 600), "res", "anchors9"),
    "rpn_c4": ("gt5", (1500, 600), "res", "anchors18"),
}


@pytest.mark.parametrize("name", list(RPN_CASES))
def test_rpn_assign_golden(ops, ref, golden, name):
    gt_key, dims, net, anc_key = RPN_CASES[name]
    w, h = (golden["img5_dims"][dims:dims + 2] if isinstance(dims, int) else dims)
    w, h = int(w), int(h)
    anc = golden[anc_key]
    rows, cols = (ref.conv_dims_vgg if net == "vgg" else ref.conv_dims_resnet)(h, w)
    can_use, is_pos, bbreg, argmax = ops.rpn_assign(rows, cols, anc, 16, golden[gt_key], w, h)
    can_use, is_pos, bbreg = can_use.cpu().numpy().astype(bool), is_pos.cpu().numpy().astype(bool), bbreg.cpu().numpy()
    assert (np.nonzero(is_pos)[0] == golden[name + "_is_pos"]).all()
    assert (np.packbits(can_use) == golden[name + "_can_use"]).all()
    assert not bbreg[~is_pos].any()
    assert np.array_equal(bbreg[is_pos], golden[name + "_bbreg_rows"])
    o_cu, o_ip, o_bb, o_arg = ref.rpn_assign(golden[gt_key], rows, cols, anc, 16, w, h)
    assert np.array_equal(argmax.cpu().numpy(), o_arg)


def test_rpn_assign_random_gt(ops, ref, golden):
    """many GT boxes, ties between anchors (identical GT boxes) and GT outside the image."""
    rs = np.random.RandomState(5)
    gt = (rs.rand(40, 4) * np.array([900, 500, 300, 300])).astype(np.float32)
    gt[:, 2:] += gt[:, :2] + 8
    gt[7] = gt[3]                       # duplicate GT -> arg-max ties resolve to the first
    gt[11] = [2000, 2000, 2100, 2100]   # no overlap with any anchor
    anc = golden["anchors9"]
    got = ops.rpn_assign(38, 63, anc, 16, gt, 1000, 600)
    want = ref.rpn_assign(gt, 38, 63, anc, 16, 1000, 600)
    assert np.array_equal(got[0].cpu().numpy().astype(bool), want[0])
    assert np.array_equal(got[1].cpu().numpy().astype(bool), want[1])
    assert np.array_equal(got[2].cpu().numpy(), want[2])
    assert np.array_equal(got[3].cpu().numpy(), want[3])


def test_rpn_assign_no_gt(ops, golden):
    can_use, is_pos, bbreg, _ = ops.rpn_assign(5, 6, golden["anchors9"], 16, np.zeros((0, 4), np.float32), 96, 80)
    assert not is_pos.any() and not bbreg.any()


def decode_boundary_mask(ref, anc_conv, regr):
    """proposals whose pre-round decode sits within 1e-3 of a .5 boundary: there a 1-ulp
    difference between numpy's SIMD expf and the device exp may legally flip np.round."""
    pre = ref.decode_preround(anc_conv, regr.reshape(-1, 4) / ref.BBREG_MULTIPLIERS)
    frac = np.abs(pre - np.floor(pre) - 0.5)
    return (frac < 1e-3 * np.maximum(1.0, np.abs(pre))).any(axis=1)


@pytest.mark.parametrize("tag", ["tiny", "c2", "c4"])
def test_decode_proposals(ops, ref, golden, tag):
    rows, cols, A = synth.SHAPES[tag]
    anc = golden["anchors9"] if A == 9 else golden["anchors18"]
    regr, _ = synth.rpn_outputs(tag)
    rois, valid = ops.decode_proposals(dev(regr), anc // 16)
    rois, valid = rois.cpu().numpy(), valid.cpu().numpy().astype(bool)
    want = golden[f"prop_{tag}_rois_i16"].astype(np.float32)
    diff = (rois != want).any(axis=1)
    boundary = decode_boundary_mask(ref, ref.anchors_conv(rows, cols, anc // 16).reshape(-1, 4), regr)
    assert not (diff & ~boundary).any(), f"{(diff & ~boundary).sum()} non-boundary mismatches"
    assert diff.sum() <= 4, f"{diff.sum()} boundary flips"          # expected ~0.35 per 21 546 anchors
    assert np.array_equal(valid[~diff], ref.valid_mask(want)[~diff])


def test_transform_inplace(ops, golden):
    c = dev(np.array([[0, 0, 8, 8], [3, 2, 8, 13]], dtype=np.float32))
    d = dev(np.array([[.1, -.2, .3, -.4], [0, 0, 0, 0]], dtype=np.float32))
    out = ops.transform_inplace(c, d)
    assert out.data_ptr() == c.data_ptr()
    assert np.array_equal(c.cpu().numpy(), golden["kat_transform_np"])


@pytest.mark.parametrize("tag", ["tiny", "c2", "c4"])
def test_topk_order(ops, ref, golden, tag):
    rows, cols, A = synth.SHAPES[tag]
    _, cls = synth.rpn_outputs(tag)
    rois = golden[f"prop_{tag}_rois_i16"].astype(np.float32)
    valid = ref.valid_mask(rois)
    for K in (8000, 12000):
        order, n = ops.topk_order(dev(cls.reshape(-1)), dev(valid.astype(np.uint8)), K)
        n = int(n.item())
        want = golden[f"prop_{tag}_{K}_order"]
        assert n == len(want)
        assert np.array_equal(order.cpu().numpy()[:n], want)
        assert (order.cpu().numpy()[n:] == -1).all()


def test_topk_ties_and_negative(ops):
    s = np.array([0.5, -1.0, 0.5, 2.0, -0.0, 0.0, 0.5, -3.5], dtype=np.float32)
    order, n = ops.topk_order(dev(s), None, 8)
    want = np.argsort(-s, kind="stable")
    # -0.0 and 0.0 compare equal in numpy but order by bit pattern here: positions 4,5 may swap
    got = order.cpu().numpy()
    assert int(n.item()) == 8
    assert list(got[:4]) == list(want[:4]) and set(got[4:6]) == {4, 5} and list(got[6:]) == list(want[6:])
    dirty(4 * 2000, 4)
    order, n = ops.topk_order(dev(s), None, 2000)            # K far beyond N: every slot past n is written (-1)
    got = order.cpu().numpy()
    assert int(n.item()) == 8 and list(got[:4]) == list(want[:4]) and (got[8:] == -1).all()


@pytest.mark.parametrize("N", [21546, 64296])
def test_topk_in_a_replayed_graph(ops, N):
    """Both top-K paths captured into a hipGraph between other kernels and replayed back to back: every replay starts
    from cleared counters (a lone hipMemsetAsync node did not guarantee that, see frcnn_topk_order)."""
    import torch
    rs = np.random.RandomState(5)
    s = dev(rs.rand(N).astype(np.float32))
    rois = dev((rs.rand(N, 4) * 500).astype(np.float32))
    K = 8000
    want_order, want_n = ops.topk_order(s, None, K)
    want_cand, _ = ops.gather_candidates(rois, s, want_order, want_n, K)
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        ops.topk_order(s * 1.0, None, K)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        s2 = s * 1.0                                            # a kernel node in front of the reset
        order, n = ops.topk_order(s2, None, K)
        cand, _ = ops.gather_candidates(rois, s2, order, n, K)
    for _ in range(40):
        g.replay()
    torch.cuda.synchronize()
    assert int(n.item()) == int(want_n.item())
    assert torch.equal(order, want_order) and torch.equal(cand, want_cand)


@pytest.mark.parametrize("case", ["all_equal", "three_values", "saturated", "one_binade", "wide", "k_ge_n", "none_valid"])
def test_topk_distributions(ops, case):
    """The histogram prefilter must never change the answer: ties by ascending index, whatever the bins hold."""
    rs = np.random.RandomState(11)
    N, K = 64296, 8000
    valid = rs.rand(N) < 0.9
    if case == "all_equal":
        s = np.full(N, 0.5, np.float32)                              # every key in ONE bin: degrades to the full sweep
    elif case == "three_values":
        s = rs.choice(np.array([0.25, 0.5, 0.75], np.float32), N)
    elif case == "saturated":
        s = np.where(rs.rand(N) < 0.3, 1.0, rs.rand(N)).astype(np.float32)      # 30 % exactly 1.0 (sigmoid saturation)
    elif case == "one_binade":
        s = (0.5 + 0.001 * rs.rand(N)).astype(np.float32)            # a handful of histogram bins
    elif case == "wide":
        s = (rs.randn(N) * 1e3).astype(np.float32)                   # negative and positive, many binades
    elif case == "k_ge_n":
        N, K = 5000, 8000
        valid = valid[:N]
        s = rs.rand(N).astype(np.float32)
    else:
        s = rs.rand(N).astype(np.float32)
        valid = np.zeros(N, bool)
    order, n = ops.topk_order(dev(s), dev(valid.astype(np.uint8)), K)
    idx = np.flatnonzero(valid)
    want = idx[np.argsort(-s[idx].astype(np.float64), kind="stable")][:K]          # score descending, index ascending
    n = int(n.item())
    assert n == len(want)
    got = order.cpu().numpy()
    assert np.array_equal(got[:n], want)
    assert (got[n:] == -1).all()


@pytest.mark.parametrize("tag", ["tiny", "c2", "c4"])
def test_nms_golden(ops, golden, tag):
    _, cls = synth.rpn_outputs(tag)
    rois = golden[f"prop_{tag}_rois_i16"].astype(np.float32)
    probs = cls.reshape(-1)
    for pre, post in ((8000, 300), (12000, 2000)):
        order = golden[f"prop_{tag}_{pre}_order"]
        K = pre
        o = np.full(K, -1, dtype=np.int32)
        o[:len(order)] = order
        n = dev(np.array([len(order)], dtype=np.int32))
        cand, cs = ops.gather_candidates(dev(rois), dev(probs), dev(o), n, K)
        assert np.array_equal(cand.cpu().numpy()[:len(order)], rois[order].astype(np.int16))
        assert np.array_equal(cs.cpu().numpy()[:len(order)], probs[order])
        dirty(post * 4, 4)
        keep, n_keep = ops.nms_sorted(cand, n, 0.7, post)
        nk = int(n_keep.item())
        want = golden[f"prop_{tag}_{pre}_pick"]
        assert nk == len(want)
        assert np.array_equal(keep.cpu().numpy()[:nk], want)
        assert (keep.cpu().numpy()[nk:] == -1).all()
        out = ops.gather_rois(cand, keep, n_keep, 64, ((post + 63) // 64) * 64).cpu().numpy()
        assert np.array_equal(out[:nk], golden[f"prop_{tag}_{pre}_kept"].astype(np.float32))
        pad_to = (nk + 63) // 64 * 64
        if pad_to > nk:
            assert (out[nk:pad_to] == out[nk // 64 * 64]).all()


def test_nms_kats_and_edges(ops, ref, golden):
    kb = golden["kat_boxes"].astype(np.int16)
    ks = np.array([.9, .8, .95, .5, .6], dtype=np.float32)
    for th, key in ((0.7, "kat_nms_7"), (0.5, "kat_nms_5")):
        order = np.argsort(-ks, kind="stable")
        keep, nk = ops.nms_sorted(dev(kb[order]), dev(np.array([5], np.int32)), th, 300)
        got = kb[order][keep.cpu().numpy()[:int(nk.item())]]
        assert np.array_equal(got, golden[key])
    # empty input (the reference returns [] there)
    dirty(40, 4)
    keep, nk = ops.nms_sorted(dev(np.zeros((8, 4), np.int16)), dev(np.array([0], np.int32)), 0.7, 10)
    assert int(nk.item()) == 0 and (keep.cpu().numpy() == -1).all()
    keep, nk = ops.nms_sorted(dev(np.zeros((0, 4), np.int16)), dev(np.array([0], np.int32)), 0.7, 10)      # K == 0: no kernel runs
    assert int(nk.item()) == 0 and (keep.cpu().numpy() == -1).all()
    # exact-threshold case: overlap == 0.7 exactly must be KEPT (<=): boxes of area 10 with 7 shared...
    a = np.array([[0, 0, 9, 16], [0, 0, 9, 6], [0, 0, 9, 9]], dtype=np.int16)   # inter/union = 70/170, 100/170 ...
    for th in (70 / 170, 100 / 170, 0.7):
        want = ref.nms(a, np.array([3., 2., 1.], np.float32), th, 10)[2]
        keep, nk = ops.nms_sorted(dev(a), dev(np.array([3], np.int32)), th, 10)
        assert list(keep.cpu().numpy()[:int(nk.item())]) == list(want)


def test_nms_random_int16_vs_oracle(ops, ref):
    rs = np.random.RandomState(21)
    for n, max_boxes, th in ((1, 5, 0.7), (63, 10, 0.3), (64, 300, 0.7), (65, 300, 0.5), (3000, 100, 0.6), (12288, 2000, 0.7)):
        xy = rs.randint(0, 80, (n, 2))
        wh = rs.randint(1, 40, (n, 2))
        b = np.concatenate([xy, xy + wh], axis=1).astype(np.int16)
        s = ((rs.permutation(n) + 0.5) / n).astype(np.float32)
        order = np.argsort(-s, kind="stable")
        want = ref.nms(b[order], s[order], th, max_boxes)[2]
        keep, nk = ops.nms_sorted(dev(b[order]), dev(np.array([n], np.int32)), th, max_boxes)
        nk = int(nk.item())
        assert nk == len(want), (n, nk, len(want))
        assert np.array_equal(keep.cpu().numpy()[:nk], want)


def test_nms_f64(ops, golden):
    fb, fs = golden["nms_f64_boxes"], golden["nms_f64_scores"]
    order = np.argsort(-fs, kind="stable")
    keep, nk = ops.nms_sorted(dev(fb[order]), dev(np.array([len(fb)], np.int32)), 0.5, 2000)
    nk = int(nk.item())
    got = fb[order][keep.cpu().numpy()[:nk]]
    assert np.array_equal(got, golden["nms_f64_kept"])
    assert np.array_equal(fs[order][keep.cpu().numpy()[:nk]], golden["nms_f64_kept_scores"])


def test_roi_targets(ops, ref, golden):
    kept = golden["prop_c2_12000_kept"]
    gt64 = np.array([[c * (1 / 16) for c in b] for b in synth.GT5], dtype=np.float64)
    gt32 = gt64.astype(np.float32)
    elig, cls, tg = ops.roi_targets(dev(kept), dev(gt32), dev(gt64), dev(golden["truth_gt_cls"].astype(np.int32)), 20)
    elig, cls, tg = elig.cpu().numpy().astype(bool), cls.cpu().numpy(), tg.cpu().numpy()
    assert np.array_equal(kept[elig], golden["truth_c2_rois"])
    assert np.array_equal(cls[elig], golden["truth_c2_cls"])
    rows = golden["truth_c2_pos_rows"]
    assert np.array_equal(tg[elig][rows], golden["truth_c2_pos_targets"])
    neg = np.ones(elig.sum(), bool)
    neg[rows] = False
    assert not tg[elig][neg].any()


def test_roi_crop_resize(ops):
    from oracle import keras_ref
    rs = np.random.RandomState(2)
    feat = rs.randn(38, 63, 64).astype(np.float32)
    rois = np.array([[0, 0, 62, 37], [5, 5, 6, 6], [10, 3, 17, 10], [10, 3, 24, 8], [61, 36, 62, 37], [3, 0, 5, 30],
                     [20, 20, 34, 34], [0, 0, 1, 37]], dtype=np.float32)
    got = ops.roi_crop_resize(dev(feat), dev(rois), 7).cpu().numpy()
    want = keras_ref.roi_resize(feat, rois, 7)
    assert np.array_equal(got, want)
    # backward: adjoint test  <dout, fwd(x)> == <bwd(dout), x>
    dout = rs.randn(*got.shape).astype(np.float32)
    dfeat = ops.roi_crop_resize_bwd(dev(dout), dev(rois), 38, 63).cpu().numpy()
    lhs = float((dout.astype(np.float64) * want.astype(np.float64)).sum())
    rhs = float((dfeat.astype(np.float64) * feat.astype(np.float64)).sum())
    assert abs(lhs - rhs) <= 1e-4 * max(1.0, abs(lhs))
    # ... and bit for bit the sequential f32 scatter in the order TF's CPU ResizeBilinearGrad walks the samples
    # (roi, py, px; top-left, top-right, bottom-left, bottom-right): the gather sums every cell's taps in that order
    seq = np.zeros_like(feat)
    one = np.float32(1)
    for r, (x1, y1, x2, y2) in enumerate(rois.astype(np.int64)):
        h, w = y2 - y1, x2 - x1
        sy, sx = np.float32(h) / np.float32(7), np.float32(w) / np.float32(7)
        for py in range(7):
            fy = np.float32(py) * sy
            ly = int(fy); ty = fy - np.float32(ly); yl, yh = y1 + ly, y1 + min(ly + 1, h - 1)
            for px in range(7):
                fx = np.float32(px) * sx
                lx = int(fx); tx = fx - np.float32(lx); xl, xh = x1 + lx, x1 + min(lx + 1, w - 1)
                gv = dout[r, py, px]
                dtop, dbot = (one - ty) * gv, ty * gv
                seq[yl, xl] += dtop * (one - tx)
                seq[yl, xh] += dtop * tx
                seq[yh, xl] += dbot * (one - tx)
                seq[yh, xh] += dbot * tx
    assert np.array_equal(dfeat, seq)
    assert not ops.roi_crop_resize_bwd(dev(dout[:0]), dev(rois[:0]), 38, 63).any().item()      # no RoIs: zeros


def test_detections_device(ops):
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "dets_legacy.npz"))
    rois = g["rois"].astype(np.float32)
    n = len(rois)
    for tag in ("t0", "t5", "t0r", "tb"):     # tb: threshold one f64 step above a row's f32 score (the f64 comparison of the pinned numpy)
        thr, ratio = g[tag + "_args"]
        for rows in (n, 320):                 # unpadded (fast path) and with the reference's padded duplicates
            r = np.zeros((rows, 4), np.float32)
            r[:n] = rois
            if rows > n:
                r[n:] = rois[256]
            dirty(rows * 4, rows * 4, rows * 16, rows * 4, 4)
            out = ops.detections(dev(r), dev(np.array([rows], np.int32)), dev(g["out_cls"][:rows]), dev(g["out_reg"][:rows]),
                                 64, 20, float(thr), 16.0, float(ratio))
            nd = int(out["n_dets"].item())
            assert (out["det_cls"].cpu().numpy()[nd:] == -1).all() and (out["det_roi"].cpu().numpy()[nd:] == -1).all()
            assert not out["det_prob"].cpu().numpy()[nd:].any() and not out["det_bbox"].cpu().numpy()[nd:].any()
            assert nd == len(g[tag + "_cls"])
            assert np.array_equal(out["det_cls"].cpu().numpy()[:nd], g[tag + "_cls"])
            assert np.array_equal(out["det_prob"].cpu().numpy()[:nd], g[tag + "_prob"])
            assert np.array_equal(out["det_bbox"].cpu().numpy()[:nd], g[tag + "_bbox"])
            # the captured-pass form: the two scalars from device memory; roi_batch 64 scores the padded list out of the live count
            dyn = dev(np.array([ratio, thr], np.float64))
            for n_live, batch in ((rows, 0), (n, 64)):
                if n_live + (-n_live % max(batch, 1)) > rows:
                    continue
                dirty(rows * 4, rows * 4, rows * 16, rows * 4, 4)
                o2 = ops.detections_dyn(dev(r), dev(np.array([n_live], np.int32)), dev(g["out_cls"][:rows]), dev(g["out_reg"][:rows]), batch, 20, 16.0, dyn)
                assert o2["det_packed"][:2].tolist() == [nd, n_live]
                for k in ("det_cls", "det_prob", "det_bbox", "det_roi"):
                    assert torch.equal(o2[k], out[k]), (tag, rows, batch, k)
    # nothing above threshold
    out = ops.detections(dev(rois), dev(np.array([n], np.int32)), dev(g["out_cls"][:n]), dev(g["out_reg"][:n]), 64, 20, 2.0, 16.0, 1.0)
    assert int(out["n_dets"].item()) == 0 and (out["det_cls"].cpu().numpy() == -1).all()


def test_preprocess_u8_is_bit_identical_to_the_host_path(ops):
    from faster_rcnn_amd import resnet
    rs = np.random.RandomState(0)
    img = rs.randint(0, 256, (37, 53, 3)).astype(np.uint8)
    want = resnet.preprocess(img)[None].astype(np.float32)             # float64(img) - mean, then the f32 input cast
    got = ops.preprocess_u8(img, (103.939, 116.779, 123.68)).cpu().numpy()
    assert got.shape == want.shape and np.array_equal(got, want)
