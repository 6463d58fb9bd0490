#!/usr/bin/env python3
"""Generate golden vectors by IMPORTING the reference's numpy half (build container only).

Run:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py
Needs /root/reference (absent on the GPU box); writes tests/golden/*.npz, which are
committed.  Only data (inputs/outputs) is stored -- never reference source.

The Keras half of the reference (resnet/vgg/custom_layers/loss_functions) cannot be
imported here (no TensorFlow/Keras), so no goldens exist for it: see oracle/keras_ref.py.
"""
import hashlib
import os
import random
import sys
import types

import numpy as np

REF = "/root/reference/faster_rcnn"
OUT = os.path.dirname(os.path.abspath(__file__))

sys.dont_write_bytecode = True
sys.modules["cv2"] = types.ModuleType("cv2")      # shapes.py / voc_data_helpers import cv2 at module level only
sys.path.insert(0, REF)

import det_util        # noqa: E402
import rpn_util        # noqa: E402
import shapes          # noqa: E402
import util            # noqa: E402
from data import voc_data_helpers  # noqa: E402
import eval_dets       # noqa: E402

VOC = "/root/reference/test_data/VOC_test"

GT5 = [[100, 100, 300, 400], [400, 50, 900, 550], [10, 10, 60, 80], [500, 300, 620, 420], [700, 100, 990, 590]]
GT5_CLS = ["cat", "person", "chair", "cat", "bicycle"]


def sha(*arrs):
    h = hashlib.sha1()
    for a in arrs:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def synth_image(w, h, boxes=GT5, classes=GT5_CLS):
    gts = [shapes.GroundTruthBox(obj_cls=c, difficult=False, box=shapes.Box(*b)) for b, c in zip(boxes, classes)]
    md = shapes.Metadata("synth", width=w, height=h, gt_boxes=gts, image_path="none")
    return shapes.Image(md)


def conv_dims_resnet(height, width):   # resnet.py is not importable; formula restated, pinned by KATs below
    dims = [height, width]
    for i in range(2):
        dims[i] += 6
        for f in (7, 3, 1, 1):
            dims[i] = (dims[i] - f) // 2 + 1
    return dims


def conv_dims_vgg(height, width):
    return height // 16, width // 16


def main():
    g = {}
    # ---- a1 anchors table
    g["anchors9"] = util.get_anchors([128, 256, 512])
    g["anchors18"] = util.get_anchors([16, 32, 64, 128, 256, 512])

    # ---- a2/a3 image-space anchors
    a9 = g["anchors9"]
    a18 = g["anchors18"]
    g["anc_img_3x4"] = rpn_util._get_all_anchor_coords(3, 4, a9, 16)
    c2 = rpn_util._get_all_anchor_coords(38, 63, a9, 16)
    c4 = rpn_util._get_all_anchor_coords(38, 94, a18, 16)
    g["anc_img_c2_i16"] = c2.astype(np.int16)          # integer valued, exact in int16
    g["anc_img_c4_i16"] = c4.astype(np.int16)
    assert (g["anc_img_c2_i16"] == c2).all() and (g["anc_img_c4_i16"] == c4).all()
    g["oob_c2"] = rpn_util._get_out_of_bounds_idxs(c2, 1000, 600)
    g["oob_c4"] = rpn_util._get_out_of_bounds_idxs(c4, 1500, 600)

    # ---- a4 IoU
    gt5 = np.array(GT5, dtype=np.float32)
    g["gt5"] = gt5
    g["iou_c2_gt5"] = util.cross_ious(c2, gt5)
    kat_boxes = np.array([[0, 0, 10, 10], [1, 1, 11, 11], [0, 0, 10, 9], [20, 20, 30, 30], [5, 5, 15, 15]], dtype=np.float32)
    g["kat_boxes"] = kat_boxes
    g["kat_iou"] = util.cross_ious(kat_boxes, np.array([[0, 0, 10, 10], [4, 4, 12, 12]], dtype=np.float32))
    rs = np.random.RandomState(7)
    rb = np.sort(rs.randint(0, 60, (500, 2, 2)), axis=1).transpose(0, 2, 1).reshape(500, 4)[:, [0, 2, 1, 3]]
    rb = np.stack([rb[:, 0], rb[:, 1], rb[:, 0] + 1 + rs.randint(0, 30, 500), rb[:, 1] + 1 + rs.randint(0, 30, 500)], axis=1).astype(np.int16)
    gtf = (rs.rand(7, 4) * 30).astype(np.float32)
    gtf[:, 2:] += gtf[:, :2] + 1
    g["iou_i16_boxes"] = rb
    g["iou_i16_gt"] = gtf
    g["iou_i16"] = util.cross_ious(rb, gtf)

    # ---- a5/a6/a8 RPN targets on the real annotation (000005) and on synthetic GT at C2/C4
    img5 = voc_data_helpers.extract_img_data(VOC, "000005")
    (img5_rs,), (ratio5,) = util.resize_imgs([img5], 600, 1000)
    g["img5_gt"] = util.get_bbox_coords(img5.gt_boxes)
    g["img5_rs_gt"] = util.get_bbox_coords(img5_rs.gt_boxes)
    g["img5_rs_gt_f64"] = np.array([b.corners for b in img5_rs.gt_boxes], dtype=np.float64)
    g["img5_rs_gt_cls"] = np.array([voc_data_helpers.VOC_CLASS_MAPPING[b.obj_cls] for b in img5_rs.gt_boxes])
    g["img5_dims"] = np.array([img5.width, img5.height, img5_rs.width, img5_rs.height])
    g["img5_ratio"] = np.array([ratio5])
    cases = {
        "rpn_img5_vgg": (img5, conv_dims_vgg, a9),
        "rpn_img5rs_vgg": (img5_rs, conv_dims_vgg, a9),
        "rpn_img5rs_res": (img5_rs, conv_dims_resnet, a9),
        "rpn_c2": (synth_image(1000, 600), conv_dims_resnet, a9),
        "rpn_c4": (synth_image(1500, 600), conv_dims_resnet, a18),
    }
    for name, (img, dims_fn, anc) in cases.items():
        mgr = rpn_util.RpnTrainingManager(dims_fn, 16, lambda x: x, anc)
        mgr._process(img)
        res = mgr._cache[img.cache_key]
        g[name + "_can_use"] = np.packbits(res["can_use"])
        g[name + "_is_pos"] = np.nonzero(res["is_pos"])[0]
        g[name + "_bbreg_rows"] = res["bbreg_targets"][res["is_pos"]]
        assert not res["bbreg_targets"][~res["is_pos"]].any()
        g[name + "_sha"] = np.frombuffer(bytes.fromhex(sha(res["can_use"], res["is_pos"], res["bbreg_targets"])), dtype=np.uint8)
        random.seed(1)
        y_class, y_bbreg = mgr.rpn_y_true(img)
        g[name + "_ycls_sha"] = np.frombuffer(bytes.fromhex(sha(y_class, y_bbreg)), dtype=np.uint8)
        g[name + "_ycls_shape"] = np.array(y_class.shape + y_bbreg.shape)
        g[name + "_ycls_use"] = np.nonzero(y_class[0, :, :, :len(anc)].reshape(-1))[0]

    # ---- a9-a13 proposal pipeline on seeded synthetic RPN outputs
    for tag, rows, cols, anc in (("c2", 38, 63, a9), ("c4", 38, 94, a18), ("tiny", 5, 7, a9)):
        rs = np.random.RandomState(0)
        regr = (rs.randn(1, rows, cols, 4 * len(anc)) * 0.5).astype(np.float32)
        n_anc = rows * cols * len(anc)
        # tie-free scores: f32 rand() collides (~14 equal pairs at N=21546) and the reference's
        # argsort tie order is unspecified, so use a permutation of distinct values instead
        cls = ((rs.permutation(n_anc).astype(np.float32) + 0.5) / n_anc).reshape(1, rows, cols, len(anc))
        anc_conv = det_util._get_anchor_coords(rows, cols, anc // 16)
        g[f"prop_{tag}_anc_conv_sha"] = np.frombuffer(bytes.fromhex(sha(anc_conv)), dtype=np.uint8)
        if tag == "tiny":
            g["prop_tiny_anc_conv"] = anc_conv
        rois = det_util._get_rois(regr, anc, 16)
        g[f"prop_{tag}_rois_i16"] = rois.astype(np.int16)
        assert (rois.astype(np.int16) == rois).all()
        probs = cls.reshape(-1)
        valid = det_util._get_valid_box_idxs(rois)
        g[f"prop_{tag}_nvalid"] = np.array([len(valid)])
        r, p = rois[valid], probs[valid]
        for pre, post in ((8000, 300), (12000, 2000)):
            order = p.argsort()[::-1][:pre]
            cand = r[order].astype("int16")
            cp = p[order]
            kept, kp = det_util.nms(cand, cp, max_boxes=post, overlap_thresh=0.7)
            # recover pick indices: candidates are unique in (score) so match by prob
            pos = {float(v): i for i, v in enumerate(cp)}
            pick = np.array([pos[float(v)] for v in kp], dtype=np.int32)
            assert (cand[pick] == kept).all()
            g[f"prop_{tag}_{pre}_order"] = valid[order].astype(np.int32)
            g[f"prop_{tag}_{pre}_pick"] = pick
            g[f"prop_{tag}_{pre}_kept"] = kept
            if (tag, pre) == ("c2", 12000):
                # a14 detector targets from those 2000 rois against the synthetic GT (conv units)
                img = synth_image(1000, 600)
                e_rois, onehot, bb = det_util._rois_to_truth(kept, img, voc_data_helpers.VOC_CLASS_MAPPING, stride=16)
                g["truth_c2_rois"] = e_rois
                g["truth_c2_cls"] = onehot.argmax(axis=1).astype(np.int32)
                assert (onehot.sum(axis=1) == 1).all()
                g["truth_c2_bbreg_sha"] = np.frombuffer(bytes.fromhex(sha(bb)), dtype=np.uint8)
                nzr = np.nonzero(bb[:, :80].any(axis=1))[0]
                g["truth_c2_pos_rows"] = nzr
                g["truth_c2_pos_targets"] = np.stack([bb[i, 80 + 4 * c:84 + 4 * c] for i, c in zip(nzr, g["truth_c2_cls"][nzr])])
                g["truth_gt_cls"] = np.array([voc_data_helpers.VOC_CLASS_MAPPING[c] for c in GT5_CLS])
                np.random.seed(1337)
                found = onehot[:, -1] == 0
                g["truth_c2_samples"] = np.array(det_util._get_det_samples(found, 64))

    # ---- nms KATs (int16 and float64 inputs)
    kb = kat_boxes.astype(np.int16)
    ks = np.array([.9, .8, .95, .5, .6], dtype=np.float32)
    for th in (0.7, 0.5):
        kept, kp = det_util.nms(kb, ks, overlap_thresh=th, max_boxes=300)
        g[f"kat_nms_{int(th * 10)}"] = kept
    rs = np.random.RandomState(3)
    fb = rs.rand(400, 4) * 300
    fb[:, 2:] = fb[:, :2] + 20 + rs.rand(400, 2) * 200
    fs = rs.rand(400).astype(np.float32)
    kept, kp = det_util.nms(fb, fs, overlap_thresh=0.5, max_boxes=2000)
    g["nms_f64_boxes"] = fb
    g["nms_f64_scores"] = fs
    g["nms_f64_kept"] = kept
    g["nms_f64_kept_scores"] = kp

    # ---- scalar KATs
    g["kat_reg_params"] = np.array(util.get_reg_params([0, 0, 10, 10], [2, 3, 8, 13]), dtype=np.float64)
    g["kat_transform"] = np.array(util.transform([0, 0, 8, 8], [.1, -.2, .3, -.4]), dtype=np.float64)
    g["kat_transform_np"] = util.transform_np_inplace(
        np.array([[0, 0, 8, 8], [3, 2, 8, 13]], dtype=np.float32), np.array([[.1, -.2, .3, -.4], [0, 0, 0, 0]], dtype=np.float32))
    g["kat_rois_zero"] = det_util._get_rois(np.zeros((1, 2, 3, 36), dtype=np.float32), a9, 16)
    g["kat_conv_dims_in"] = np.array([600, 1000, 1500, 800, 375, 500])
    g["kat_conv_dims_resnet"] = np.array([38, 63, 94, 50, 24, 31])    # SURVEY 8(c) [probe]

    # ---- VOC07 AP (eval_dets.py:8-36) for the f3 row
    rs = np.random.RandomState(5)
    rec = np.sort(rs.rand(50))
    prec = np.sort(rs.rand(50))[::-1].copy()
    g["ap_rec"], g["ap_prec"] = rec, prec
    g["ap_val"] = np.array([eval_dets.voc_ap(rec, prec)])

    np.savez_compressed(os.path.join(OUT, "numpy_half.npz"), **g)
    tot = os.path.getsize(os.path.join(OUT, "numpy_half.npz"))
    print(f"wrote {len(g)} arrays, {tot/1e6:.2f} MB")


if __name__ == "__main__":
    main()
