import os, sys, traceback, io, contextlib
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.argv = ["bench.py", "--config", "c4"]
import numpy as np, torch
import bench
bench.configure(["--config", "c4"]) if hasattr(bench, "configure") else None
from faster_rcnn_amd import resnet, shapes, voc_dets, util
from faster_rcnn_amd.weights import synthetic_resnet
from faster_rcnn_amd.data.voc_data_helpers import KITTI_CLASS_MAPPING
from faster_rcnn_amd.det_util import DetTrainingManager
anchors = util.get_anchors([16, 32, 64, 128, 256, 512])
w = synthetic_resnet(101, anchors_per_loc=len(anchors), num_classes=10, seed=1)
base = resnet.resnet101_base(weights=w, dtype="bf16")
rpn = resnet.resnet101_rpn(base, include_conv=True, anchors_per_loc=len(anchors))
det = resnet.resnet101_classifier(300, 10, weights=w, dtype="bf16")
mgr = DetTrainingManager(rpn_model=rpn, class_mapping=KITTI_CLASS_MAPPING, preprocess_func=resnet.preprocess, anchor_dims=anchors)
rs = np.random.RandomState(0)
imgs = [shapes.Image(shapes.Metadata("s%d" % i, 1500, 600, [], "none"), rs.randint(0, 256, (600, 1500, 3)).astype(np.uint8)) for i in range(6)]
for fast in (True, False):
    voc_dets.FAST_ENTRY = fast
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            d = voc_dets.get_dets_by_cls(mgr, det, [1.0] * len(imgs), imgs)
        print("fast" if fast else "eager", "ok:", {k: sum(len(v) for v in d[k].values()) for k in d})
    except Exception:
        print("fast" if fast else "eager", "FAILED"); traceback.print_exc()
