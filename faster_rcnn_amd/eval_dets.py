"""Mirror of the reference's eval_dets (eval_dets.py:8-151): PASCAL-VOC detection scoring of the
``comp3_det_test_<cls>.txt`` files voc_dets.write_dets produces -- the metric behind "box mAP delta"
(SURVEY 8(f) f3).  Host code; pinned by goldens captured from the imported reference
(tests/golden/make_golden.py ``ap_val``, tests/golden/make_golden_eval.py).

Reference behaviour kept on purpose: boxes use the +1 pixel convention; a detection is a match only if its best
IoU is strictly above the threshold; 'difficult' objects are neither rewarded nor punished; ``voc_eval`` always
scores with the VOC07 11-point rule even though ``voc_ap`` defaults to the area rule (eval_dets.py:125).
"""
import os

import numpy as np

from .data.voc_data_helpers import extract_img_data


def voc_ap(rec, prec, use_07_metric=False):
    """eval_dets.py:8-36.  11-point rule: mean over t = 0, .1, ..., 1 of the best precision at recall >= t.
    Area rule: area under the monotone (right-to-left running max) precision envelope at the recall steps."""
    rec, prec = np.asarray(rec, dtype=np.float64), np.asarray(prec, dtype=np.float64)
    if use_07_metric:
        total = 0.0
        for t in np.arange(0.0, 1.1, 0.1):
            reached = rec >= t
            total = total + (np.max(prec[reached]) if reached.any() else 0) / 11.0
        return total
    r = np.concatenate(([0.0], rec, [1.0]))
    envelope = np.maximum.accumulate(np.concatenate(([0.0], prec, [0.0]))[::-1])[::-1]
    steps = np.nonzero(r[1:] != r[:-1])[0]
    return np.sum((r[steps + 1] - r[steps]) * envelope[steps + 1])


def _iou_plus_one(box, gts):
    """IoU of one box against (G,4) boxes with the +1 convention (eval_dets.py:93-107)."""
    iw = np.maximum(np.minimum(gts[:, 2], box[2]) - np.maximum(gts[:, 0], box[0]) + 1.0, 0.0)
    ih = np.maximum(np.minimum(gts[:, 3], box[3]) - np.maximum(gts[:, 1], box[1]) + 1.0, 0.0)
    inter = iw * ih
    union = (box[2] - box[0] + 1.0) * (box[3] - box[1] + 1.0) + (gts[:, 2] - gts[:, 0] + 1.0) * (gts[:, 3] - gts[:, 1] + 1.0) - inter
    return inter / union


def voc_eval(voc_path, det_file, imageset_path, cls_name, ovthresh=0.5):
    """eval_dets.py:37-127 -> (recall, precision, ap) over the detections of one class, best score first."""
    with open(imageset_path) as f:
        names = [line.strip() for line in f]
    truth, n_positive = {}, 0
    for i, name in enumerate(names):
        if i % 100 == 0:
            print("Reading annotation for image {}/{}".format(i, len(names)))
        objs = [b for b in extract_img_data(voc_path, name).gt_boxes if b.obj_cls == cls_name]
        hard = np.array([bool(b.difficult) for b in objs], dtype=bool)
        truth[name] = {"boxes": np.array([b.corners for b in objs], dtype=np.float64).reshape(-1, 4), "hard": hard,
                       "claimed": np.zeros(len(objs), dtype=bool)}
        n_positive += int((~hard).sum())

    with open(det_file) as f:
        rows = [line.strip().split(" ") for line in f if line.strip()]
    conf = np.array([float(r[1]) for r in rows])
    boxes = np.array([[float(v) for v in r[2:]] for r in rows], dtype=np.float64).reshape(-1, 4)
    order = np.argsort(-conf)
    hit, miss = np.zeros(len(rows)), np.zeros(len(rows))
    for rank, d in enumerate(order):
        rec_ = truth[rows[d][0]]
        best, j = -np.inf, -1
        if len(rec_["boxes"]):
            ov = _iou_plus_one(boxes[d], rec_["boxes"])
            j = int(np.argmax(ov))
            best = ov[j]
        if best > ovthresh:
            if not rec_["hard"][j]:
                if rec_["claimed"][j]:
                    miss[rank] = 1.0                    # a second detection of an object already found
                else:
                    hit[rank] = 1.0
                    rec_["claimed"][j] = True
        else:
            miss[rank] = 1.0
    tp, fp = np.cumsum(hit), np.cumsum(miss)
    with np.errstate(divide="ignore", invalid="ignore"):
        rec = tp / float(n_positive)
    prec = tp / np.maximum(tp + fp, np.finfo(np.float64).eps)
    return rec, prec, voc_ap(rec, prec, use_07_metric=True)


def get_voc_results_filename(dets_path, cls_name):
    return os.path.join(dets_path, "comp3_det_test_{}.txt".format(cls_name))


def eval_all(dets_path, voc_path, class_mapping, img_set="val"):
    """eval_dets.py:134-151: per-class AP and the running mean over the classes scored so far; returns the APs."""
    aps = []
    imageset_file = os.path.join(voc_path, "ImageSets", "Main", img_set + ".txt")
    for cls_name, _ in sorted(class_mapping.items()):
        print(cls_name)
        if cls_name == "bg":
            continue
        _, _, ap = voc_eval(voc_path, get_voc_results_filename(dets_path, cls_name), imageset_file, cls_name, ovthresh=0.5)
        aps.append(ap)
        print("AP for {} = {:.4f}".format(cls_name, ap))
        print("Mean AP = {:.4f}".format(np.mean(aps)))
        print("~~~~~~~~")
        print("Results:")
        for a in aps:
            print("{:.3f}".format(a))
        print("{:.3f}".format(np.mean(aps)))
    return aps


def main(argv=None):
    import argparse
    from .data.voc_data_helpers import KITTI_CLASS_MAPPING, VOC_CLASS_MAPPING
    p = argparse.ArgumentParser(description="Score detection files against VOC-style annotations")
    p.add_argument("--voc_path", dest="voc_path", required=True, help="base path of the VOC-style test dataset")
    p.add_argument("--dets_path", dest="dets_path", default="./tmpout", help="directory holding comp3_det_test_<cls>.txt")
    p.add_argument("--kitti", dest="kitti", action="store_true", help="KITTI classes instead of Pascal VOC")
    p.add_argument("--img_set", dest="img_set", choices=("val", "test", "trainval"), default="val")
    args = p.parse_args(argv)
    eval_all(args.dets_path, args.voc_path, KITTI_CLASS_MAPPING if args.kitti else VOC_CLASS_MAPPING, img_set=args.img_set)


if __name__ == "__main__":
    main()
