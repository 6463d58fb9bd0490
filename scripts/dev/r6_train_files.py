#!/usr/bin/env python3
"""train_util.train_rpn / train_detector_step2 over FILE-backed images (the reference's use: VOC JPEGs from disk) against the same frames held
in memory: what the JPEG decode on the loop's thread costs an iteration.  Mixed bf16, device feed.  Dev tool."""
import contextlib, io, os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from faster_rcnn_amd import det_util, resnet, rpn_util, shapes, train, train_util, util
from faster_rcnn_amd.data.voc_data_helpers import VOC_CLASS_MAPPING, extract_img_data
from faster_rcnn_amd.weights import synthetic_resnet

dtype = sys.argv[1] if len(sys.argv) > 1 else "bf16"
anchors = util.get_anchors([128, 256, 512])
A, C = len(anchors), 21
root = os.path.join(ROOT, "tests", "golden", "VOC_test")
def file_frames(n):
    out = []
    for i in range(n):
        b = extract_img_data(root, "000005")
        (r,), _ = util.resize_imgs([b], min_size=600, max_size=1000)
        r.metadata.name = "file%03d" % i
        out.append(r)
    return out
def memory_twins(frames):
    return [shapes.Image(shapes.Metadata(f.name, f.width, f.height, f.gt_boxes, "none"), f.raw) for f in frames]
reg = dict(weight_regularizer=resnet.WEIGHT_REGULARIZER, bias_regularizer=resnet.BIAS_REGULARIZER)
for kind in ("rpn_step1", "det_step2"):
    for where in ("memory", "files"):
        frames = file_frames(32)
        imgs = frames if where == "files" else memory_twins(frames)
        random.seed(1); np.random.seed(1337)
        if kind == "rpn_step1":
            model = resnet.resnet50_rpn(resnet.resnet50_base(weights=synthetic_resnet(50, anchors_per_loc=A, num_classes=C, seed=1), dtype=dtype, **reg), anchors_per_loc=A)
            mgr = rpn_util.RpnTrainingManager(resnet.get_conv_rows_cols, 16, resnet.preprocess, anchors)
            loop = train_util.train_rpn
        else:
            frozen = resnet.resnet50_rpn(resnet.resnet50_base(weights=synthetic_resnet(50, anchors_per_loc=A, num_classes=C, seed=1)), anchors_per_loc=A)
            model = resnet.resnet50_classifier(64, C, resnet.resnet50_base(weights=synthetic_resnet(50, anchors_per_loc=A, num_classes=C, seed=2), dtype=dtype, **reg))
            mgr = det_util.DetTrainingManager(frozen, VOC_CLASS_MAPPING, resnet.preprocess, anchor_dims=anchors)
            loop = train_util.train_detector_step2
        opt = train.optimizer_from_str("sgd")
        with contextlib.redirect_stdout(io.StringIO()):
            loop(model, imgs, mgr, opt, phases=[[64, 1e-3]])
            train.finish_pending_updates(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            loop(model, imgs, mgr, opt, phases=[[96, 1e-3]])
            train.finish_pending_updates(); torch.cuda.synchronize()
            el = time.perf_counter() - t0
        print("%s %s, frames in %s: %.3f ms per iteration" % (kind, dtype, where, 1e3 * el / 96))
