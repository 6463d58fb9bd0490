#!/usr/bin/env python3
"""Randomised bit-exact sweep of the integer / box kernels against the numpy oracle on shapes and value patterns the
fixed tests do not enumerate: anchors, cross IoU, RPN target assignment, proposal decode (outside libm boundaries),
top-K order (ties included), NMS (int16 and f64, duplicates, tiny and huge candidate lists), RoI targets and the
detection post-process.  Dev tool; exits non-zero on the first class of mismatch it sees."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from faster_rcnn_amd import ops
from oracle import np_ref as ref


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    fails = 0

    def check(ok, what):
        nonlocal fails
        if not ok:
            fails += 1
            print("FAIL", what)

    for it in range(n_cases):
        scales = sorted(rs.choice([16, 32, 64, 128, 256, 512], size=int(rs.randint(1, 5)), replace=False).tolist())
        anc = ref.get_anchors(scales)
        A = len(anc)
        rows, cols = int(rs.randint(3, 45)), int(rs.randint(3, 100))
        H, W = rows * 16 - int(rs.randint(0, 16)), cols * 16 - int(rs.randint(0, 16))
        tag = "it=%d rows=%d cols=%d A=%d" % (it, rows, cols, A)
        # anchors
        check(np.array_equal(ops.anchors_image(rows, cols, anc, 16).cpu().numpy(), ref.anchors_image(rows, cols, anc, 16)), "anchors_image " + tag)
        check(np.array_equal(ops.anchors_conv(rows, cols, anc // 16).cpu().numpy(), ref.anchors_conv(rows, cols, anc // 16)), "anchors_conv " + tag)
        # RPN assignment with random GT (duplicates, boxes outside the image, degenerate counts)
        G = int(rs.choice([1, 2, 5, 17, 40]))
        gt = (rs.rand(G, 4) * np.array([W, H, W / 2, H / 2])).astype(np.float32)
        gt[:, 2:] += gt[:, :2] + 4
        if G > 3:
            gt[2] = gt[0]
        got = ops.rpn_assign(rows, cols, anc, 16, gt, W, H)
        want = ref.rpn_assign(gt, rows, cols, anc, 16, W, H)
        check(np.array_equal(got[0].cpu().numpy().astype(bool), want[0]) and np.array_equal(got[1].cpu().numpy().astype(bool), want[1])
              and np.array_equal(got[2].cpu().numpy(), want[2]) and np.array_equal(got[3].cpu().numpy(), want[3]), "rpn_assign " + tag)
        # cross IoU, f32 and int16 first operand
        b1 = (rs.rand(int(rs.randint(1, 400)), 4) * 60).astype(np.float32)
        b1[:, 2:] += b1[:, :2]
        check(np.array_equal(ops.cross_ious(dev(b1), dev(gt / 16)).cpu().numpy(), ref.cross_ious(b1, gt / 16)), "cross_ious f32 " + tag)
        b1i = b1.astype(np.int16)
        check(np.array_equal(ops.cross_ious(dev(b1i), dev(gt / 16)).cpu().numpy(), ref.cross_ious(b1i, gt / 16)), "cross_ious i16 " + tag)
        # proposal path: decode -> top-K -> NMS -> gather, oracle fed the same RPN outputs
        regr = (rs.randn(1, rows, cols, 4 * A) * rs.choice([0.1, 0.5, 2.0])).astype(np.float32)
        n = rows * cols * A
        cls = rs.rand(1, rows, cols, A).astype(np.float32)
        if it % 3 == 0:
            cls = np.round(cls * 8) / 8                       # heavy ties: the device orders ties by ascending index
        rois_d, valid_d = ops.decode_proposals(dev(regr), anc // 16)
        want_rois = ref.get_rois(regr, anc, 16)
        pre = ref.decode_preround(ref.anchors_conv(rows, cols, anc // 16).reshape(-1, 4), regr.reshape(-1, 4) / ref.BBREG_MULTIPLIERS)
        frac = np.abs(pre - np.floor(pre) - 0.5)
        boundary = (frac < 1e-4).any(axis=1)                   # x.5 within libm rounding: either neighbour is legitimate
        same = (rois_d.cpu().numpy() == want_rois).all(axis=1)
        check(bool((same | boundary).all()), "decode " + tag)
        rois_h = rois_d.cpu().numpy()
        valid_h = ref.valid_mask(rois_h)
        check(np.array_equal(valid_d.cpu().numpy().astype(bool), valid_h), "valid " + tag)
        K = int(rs.choice([50, 300, 2000, 8000, 12000]))
        post = int(rs.choice([1, 17, 64, 300, 2000]))
        order, n_out = ops.topk_order(dev(cls.reshape(-1)), valid_d, K)
        idx = np.flatnonzero(valid_h)
        want_order = idx[np.argsort(-cls.reshape(-1)[idx].astype(np.float64), kind="stable")][:K]
        no = int(n_out.item())
        check(no == len(want_order) and np.array_equal(order.cpu().numpy()[:no], want_order) and (order.cpu().numpy()[no:] == -1).all(), "topk " + tag + " K=%d" % K)
        cand, cs = ops.gather_candidates(rois_d, dev(cls.reshape(-1)), order, n_out, K)
        keep, n_keep = ops.nms_sorted(cand, n_out, 0.7, post)
        nk = int(n_keep.item())
        cand_h = rois_h[want_order].astype(np.int16)
        # the reference re-sorts inside nms with numpy's unstable argsort: with tied scores its order is
        # implementation-defined, so the oracle is given strictly decreasing stand-in scores in the candidates' order
        stand_in = np.linspace(1.0, 0.0, len(want_order), endpoint=False).astype(np.float64)
        want_pick = ref.nms(cand_h, stand_in, 0.7, post)[2] if len(want_order) else np.zeros(0, np.int64)
        check(nk == len(want_pick) and np.array_equal(keep.cpu().numpy()[:nk], want_pick) and (keep.cpu().numpy()[nk:] == -1).all(), "nms i16 " + tag + " K=%d post=%d" % (K, post))
        out_rows = -(-post // 64) * 64
        got_rois = ops.gather_rois(cand, keep, n_keep, 64, out_rows).cpu().numpy()
        check(np.array_equal(got_rois[:nk], cand_h[want_pick].astype(np.float32)), "gather_rois " + tag)
        # RoI targets
        if nk:
            gt64 = gt.astype(np.float64) / 16
            gcls = rs.randint(0, 20, G).astype(np.int32)
            elig, tcls, tg = ops.roi_targets(dev(cand_h[want_pick]), dev(gt64.astype(np.float32)), dev(gt64), dev(gcls), 20)
            w_rois, w_cls, w_reg = ref.rois_to_truth(cand_h[want_pick], gt.astype(np.float64), gcls, 21)[:3]
            e = elig.cpu().numpy().astype(bool)
            check(np.array_equal(cand_h[want_pick][e], w_rois), "roi_targets eligibility " + tag)
        # detections on random detector outputs
        C = int(rs.choice([2, 10, 21]))
        m = int(rs.randint(1, 300))
        rois_f = np.sort(rs.randint(0, 60, (m, 4)), axis=1).astype(np.float32)[:, [0, 1, 2, 3]]
        rois_f[:, 2:] = np.maximum(rois_f[:, 2:], rois_f[:, :2] + 1)
        logits = rs.randn(m, C).astype(np.float32) * 3
        probs = np.exp(logits - logits.max(1, keepdims=True))
        probs = (probs / probs.sum(1, keepdims=True)).astype(np.float32)
        oreg = (rs.randn(m, 4 * (C - 1)) * 0.5).astype(np.float32)
        thr = float(rs.choice([0.0, 0.3, 0.7]))
        ratio = float(rs.choice([1.0, 1.6, 0.625]))
        out = ops.detections(dev(rois_f), dev(np.array([m], np.int32)), dev(probs), dev(oreg), 64, C - 1, thr, 16.0, ratio)
        want = ref.detections(rois_f, probs, oreg, C - 1, ratio, det_threshold=thr)
        nd = int(out["n_dets"].item())
        got = [(int(out["det_cls"][i]), float(out["det_prob"][i]), tuple(int(v) for v in out["det_bbox"][i])) for i in range(nd)]
        exp = [(int(w[0]), float(w[1]), tuple(int(v) for v in w[2])) for w in want]
        runs = lambda seq: [sorted(b for c, p, b in seq if (c, p) == key) for key in dict.fromkeys((c, p) for c, p, _ in seq)]
        check(nd == len(want) and [g[:2] for g in got] == [e_[:2] for e_ in exp] and runs(got) == runs(exp), "detections " + tag + " m=%d C=%d" % (m, C))
        # RoI crop + bilinear resize (bit-exact), with the fill / ReLU / position-major variants of the hoisted head
        if it % 2 == 0:
            from oracle import keras_ref
            fr, fc, Cf = int(rs.randint(2, 20)), int(rs.randint(2, 24)), int(rs.choice([4, 8, 64]))
            feat = rs.randn(fr, fc, Cf).astype(np.float32)
            nr = int(rs.randint(1, 12))
            x1 = rs.randint(0, fc, nr); y1 = rs.randint(0, fr, nr)
            x2 = np.minimum(fc, x1 + rs.randint(0, fc, nr)); y2 = np.minimum(fr, y1 + rs.randint(0, fr, nr))      # some empty
            rr = np.stack([x1, y1, x2, y2], 1).astype(np.float32)
            pool = int(rs.choice([3, 7]))
            want_r = keras_ref.roi_resize(feat, rr, pool)
            got_r = ops.roi_crop_resize(dev(feat), dev(rr), pool).cpu().numpy()
            check(np.array_equal(got_r, want_r), "roi_crop_resize " + tag + " map %dx%dx%d" % (fr, fc, Cf))
            fill = rs.randn(Cf).astype(np.float32)
            empty = (x2 <= x1) | (y2 <= y1)
            want_f = want_r.copy()
            want_f[empty] = fill
            want_f = np.maximum(want_f, 0)
            got_f = ops.roi_crop_resize(dev(feat), dev(rr), pool, fill=dev(fill), relu=True, layout=1).cpu().numpy()
            check(np.array_equal(got_f.transpose(2, 0, 1, 3), want_f), "roi_crop_resize fill/relu/pos-major " + tag)
    print("cases %d  failures %d" % (n_cases, fails))
    sys.exit(1 if fails else 0)


if __name__ == "__main__":
    main()
