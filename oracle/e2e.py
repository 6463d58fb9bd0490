"""TEST INFRASTRUCTURE (checker only; never on the product path): the oracle END TO END beside the device END TO END.

Every other full-size parity check in this repository is stage-wise -- each stage of the HIP pipeline is compared with the
CPU restatement fed with the DEVICE's own input to that stage, which holds each kernel to its 1e-4 bar but says nothing
about how far 1e-5 of feature noise moves the final detection set.  Here the oracle runs on its own from the image to the
detections (resnet.py:395-548 -> det_util.py:136-158, 370-380 -> voc_dets.py:20-88, restated in keras_ref.py / np_ref.py),
the device runs on its own (InferencePipeline.forward_dev), and the two detection sets are compared directly and through the
reference's own scorer: both are written with voc_dets.write_dets (voc_dets.py:114-129) and scored with eval_dets.voc_eval
(eval_dets.py:37-127) -- SURVEY 8(d)'s stand-in for "box mAP delta vs ref".

Two ground truths are scored (random-init weights detect nothing meaningful, so the number of interest is the DELTA of the pair):
  fixed   the 5 annotated boxes of tests/golden/VOC_test/000005 (class "chair"), used as they are on the real image and
          scaled to the frame on the synthetic ones;
  pseudo  the oracle's own most confident detection of each of an image's five most confident classes, taken as ground
          truth: the oracle then scores high by construction and the device's AP drops as soon as its confident
          detections move, which makes the delta sensitive.
"""
import contextlib
import os
import sys
import tempfile

import numpy as np

from . import np_ref

# (x1, y1, x2, y2, difficult) as the annotation loader returns them: xmin-1 .. (voc_data_helpers.py:111-114)
FIXED_GT_000005 = [(262, 210, 323, 338, 0), (164, 263, 252, 371, 0), (4, 243, 66, 373, 1), (240, 193, 294, 298, 0), (276, 185, 311, 219, 1)]


def oracle_detect(graphs, x, anchors, num_classes, depth=50, proposals=300, resize_ratio=1.0, alt_dense_class=None):
    """One image through the CPU restatement, image -> detections.  Returns (kept proposals (n,4), detections list of
    (cls_idx, prob f32, bbox int64[4])).  ``alt_dense_class`` = [kernel, bias]: a third element, the detections the same
    pass emits with THAT dense_class layer (the layer is the last step: the pooled features are shared)."""
    import torch
    with torch.no_grad():
        feat = graphs.resnet_base(x, depth)
        cls, reg = graphs.rpn(feat)
        kept = np_ref.proposals(reg.numpy(), cls.numpy(), anchors, 16, 8000, proposals)[0]
        rois = np_ref.pad_rois(kept.astype(np.float32), 64)
        out_cls, out_reg, pooled = graphs.resnet_classifier(feat, rois, num_classes, depth, return_pooled=True)
        dets = np_ref.detections(kept, out_cls.numpy(), out_reg.numpy(), num_classes - 1, resize_ratio)
        if alt_dense_class is None:
            return kept, dets
        k, b = (torch.as_tensor(np.asarray(a), dtype=pooled.dtype) for a in alt_dense_class)
        alt_cls = torch.softmax(pooled @ k + b, dim=1)
        return kept, dets, np_ref.detections(kept, alt_cls.numpy(), out_reg.numpy(), num_classes - 1, resize_ratio)


def device_detect(pipe, x, resize_ratio=1.0):
    """The same image through the HIP pipeline on its own.  Same return shape as oracle_detect."""
    import torch
    out = pipe.forward_dev(torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).cuda(), resize_ratio)
    torch.cuda.synchronize()
    n, nd = int(out["n_rois"].item()), int(out["n_dets"].item())
    rois = out["rois"].cpu().numpy()[:n]
    c, p, b = out["det_cls"].cpu().numpy()[:nd], out["det_prob"].cpu().numpy()[:nd], out["det_bbox"].cpu().numpy()[:nd]
    return rois, [(int(c[i]), np.float32(p[i]), b[i].astype(np.int64)) for i in range(nd)]


def _write_voc(root, names, sizes, gts):
    """A minimal VOC-style tree eval_dets.voc_eval can read: Annotations/<name>.xml + ImageSets/Main/test.txt."""
    os.makedirs(os.path.join(root, "Annotations"), exist_ok=True)
    os.makedirs(os.path.join(root, "ImageSets", "Main"), exist_ok=True)
    for name, (w, h), objs in zip(names, sizes, gts):
        body = "".join("<object><name>%s</name><difficult>%d</difficult><bndbox><xmin>%d</xmin><ymin>%d</ymin><xmax>%d</xmax><ymax>%d</ymax></bndbox></object>"
                       % (cls, b[4] if len(b) > 4 else 0, b[0] + 1, b[1] + 1, b[2] + 1, b[3] + 1) for cls, b in objs)
        with open(os.path.join(root, "Annotations", name + ".xml"), "w") as f:
            f.write("<annotation><filename>%s.jpg</filename><size><width>%d</width><height>%d</height><depth>3</depth></size>%s</annotation>" % (name, w, h, body))
    with open(os.path.join(root, "ImageSets", "Main", "test.txt"), "w") as f:
        f.write("\n".join(names) + "\n")


def _score(voc_root, dets_by_image, rev_classes, classes_with_gt, tag):
    """write_dets + voc_eval over the classes that have ground truth -> mean AP (the reference's VOC07 11-point rule)."""
    from faster_rcnn_amd import eval_dets, voc_dets
    by_cls = {}
    for name, dets in dets_by_image.items():
        for c, p, b in dets:
            by_cls.setdefault(rev_classes[c], {}).setdefault(name, []).append({"bbox": np.asarray(b, np.int64), "cls_name": rev_classes[c], "prob": p})
    out_dir = os.path.join(voc_root, "dets_" + tag)
    voc_dets.write_dets(by_cls, out_dir)
    aps = []
    imageset = os.path.join(voc_root, "ImageSets", "Main", "test.txt")
    for cls in classes_with_gt:
        path = eval_dets.get_voc_results_filename(out_dir, cls)
        if not os.path.exists(path):
            aps.append(0.0)
            continue
        with contextlib.redirect_stdout(sys.stderr):            # voc_eval prints its progress; a bench line owns stdout
            aps.append(float(eval_dets.voc_eval(voc_root, path, imageset, cls)[2]))
    return float(np.mean(aps)) if aps else 0.0


def _iou(a, b):
    """IoU of two integer boxes in the reference's +1 convention (eval_dets.py:118-127)."""
    iw = min(a[2], b[2]) - max(a[0], b[0]) + 1.0
    ih = min(a[3], b[3]) - max(a[1], b[1]) + 1.0
    if iw <= 0 or ih <= 0:
        return 0.0
    inter = iw * ih
    return inter / ((a[2] - a[0] + 1.0) * (a[3] - a[1] + 1.0) + (b[2] - b[0] + 1.0) * (b[3] - b[1] + 1.0) - inter)


def match_detections(oracle_dets, device_dets, iou_thresh=0.5):
    """Greedy one-to-one matching the way a detection scorer pairs boxes (eval_dets.py:100-140): the device's detections in
    descending score, each to the unmatched oracle detection of the SAME class with the highest IoU, accepted at
    IoU >= iou_thresh.  Returns the list of (oracle score, device score, iou) of the matched pairs."""
    free = [True] * len(oracle_dets)
    pairs = []
    for c, p, b in sorted(device_dets, key=lambda d: -float(d[1])):
        best, best_j = 0.0, -1
        for j, (oc, op, ob) in enumerate(oracle_dets):
            if free[j] and oc == c:
                v = _iou(ob, b)
                if v > best:
                    best, best_j = v, j
        if best_j >= 0 and best >= iou_thresh:
            free[best_j] = False
            pairs.append((float(oracle_dets[best_j][1]), float(p), best))
    return pairs


def compare(items, class_mapping):
    """items: list of dicts {name, size (w, h) of the ORIGINAL frame, oracle: (kept, dets), device: (rois, dets)}.
    Returns the ``e2e`` object of bench.py's parity section."""
    rev = {v: k for k, v in class_mapping.items()}
    prop_same = prop_total = det_same = det_total = 0
    max_score = 0.0
    matched, matched_total, score_diffs = 0, 0, []
    classes_fired = set()
    for it in items:
        pairs = match_detections(it["oracle"][1], it["device"][1])
        matched += len(pairs)
        matched_total += max(len(it["oracle"][1]), len(it["device"][1]))
        score_diffs += [abs(a - b) for a, b, _ in pairs]
        classes_fired |= {c for c, _, _ in it["oracle"][1]}
        ok, od = it["oracle"]
        dk, dd = it["device"]
        oset = set(map(tuple, np.asarray(ok, np.float32).tolist()))
        prop_same += sum(1 for r in np.asarray(dk, np.float32).tolist() if tuple(r) in oset)
        prop_total += max(len(ok), len(dk))
        omap = {}
        for c, p, b in od:
            omap.setdefault((c,) + tuple(int(v) for v in b), []).append(float(p))
        for c, p, b in dd:
            cand = omap.get((c,) + tuple(int(v) for v in b))
            if cand:
                j = int(np.argmin([abs(q - float(p)) for q in cand]))
                max_score = max(max_score, abs(cand.pop(j) - float(p)))
                det_same += 1
        det_total += max(len(od), len(dd))
    res = {"images": len(items), "proposals_identical": "%d/%d" % (prop_same, prop_total), "detections_identical": "%d/%d" % (det_same, det_total),
           "max_score_diff": float("%.3g" % max_score),
           # the looser pairing a scorer makes: same class, IoU >= 0.5, one to one (what survives a change of precision)
           "detections_matched_iou50": "%d/%d" % (matched, matched_total),
           "matched_score_diff": {"max": float("%.3g" % (max(score_diffs) if score_diffs else 0.0)),
                                  "mean": float("%.3g" % (float(np.mean(score_diffs)) if score_diffs else 0.0))},
           "classes_detected_by_oracle": len(classes_fired)}
    names = [it["name"] for it in items]
    sizes = [it["size"] for it in items]
    with tempfile.TemporaryDirectory() as tmp:
        # (1) the fixed ground truth: 000005's five chairs, scaled from 500x375 to each frame
        fixed_cls = "chair" if "chair" in class_mapping else rev[0]          # (KITTI's map has no chair: its first class stands in)
        gts = [[(fixed_cls, (b[0] * w / 500.0, b[1] * h / 375.0, b[2] * w / 500.0, b[3] * h / 375.0, b[4])) for b in FIXED_GT_000005] for (w, h) in sizes]
        root = os.path.join(tmp, "fixed")
        _write_voc(root, names, sizes, gts)
        m_o = _score(root, {it["name"]: it["oracle"][1] for it in items}, rev, [fixed_cls], "oracle")
        m_d = _score(root, {it["name"]: it["device"][1] for it in items}, rev, [fixed_cls], "device")
        res["map_fixed_gt"] = {"oracle": round(m_o, 6), "device": round(m_d, 6)}
        # (2) the oracle's five most confident detections per image as ground truth
        gts, classes = [], []
        for it in items:
            # the most confident detection of each of the image's five most confident CLASSES (the five most confident
            # detections overall tend to share one class, and an mAP over one class says little)
            best = {}
            for d in sorted(it["oracle"][1], key=lambda d: -float(d[1])):
                best.setdefault(d[0], d)
            top = list(best.values())[:5]
            gts.append([(rev[c], tuple(int(v) for v in b)) for c, _, b in top])
            classes += [rev[c] for c, _, _ in top]
        classes = sorted(set(classes))
        root = os.path.join(tmp, "pseudo")
        _write_voc(root, names, sizes, gts)
        p_o = _score(root, {it["name"]: it["oracle"][1] for it in items}, rev, classes, "oracle")
        p_d = _score(root, {it["name"]: it["device"][1] for it in items}, rev, classes, "device")
        res["map_pseudo_gt"] = {"oracle": round(p_o, 6), "device": round(p_d, 6), "classes": len(classes)}
    res["map_pair_delta"] = float("%.3g" % max(abs(m_o - m_d), abs(p_o - p_d)))
    return res
