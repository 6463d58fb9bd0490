"""dev: soak of the captured entry path -- 240 frames of 12 sizes in random order through voc_dets.get_dets_by_cls under a graph-cache
budget small enough to evict, f32 ResNet-50 and bf16 ResNet-101 models, twice: both walks must return the same bits."""
import contextlib, io, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
os.environ["FRCNN_GRAPH_CACHE_BYTES"] = str(6 << 30)
import numpy as np, torch
from faster_rcnn_amd import entry, resnet, shapes, util, voc_dets
from faster_rcnn_amd.data.voc_data_helpers import KITTI_CLASS_MAPPING, VOC_CLASS_MAPPING
from faster_rcnn_amd.det_util import DetTrainingManager
from faster_rcnn_amd.weights import synthetic_resnet
rs = np.random.RandomState(0)
sizes = [(320, 480), (352, 480), (320, 512), (288, 448), (384, 512), (320, 544), (256, 416), (336, 496), (304, 464), (368, 528), (272, 432), (400, 560)]
frames = {s: [rs.randint(0, 256, s + (3,)).astype(np.uint8) for _ in range(3)] for s in sizes}
def walk(mgr, det, order):
    imgs = [shapes.Image(shapes.Metadata("f%03d" % i, s[1], s[0], [], "none"), frames[s][k]) for i, (s, k) in enumerate(order)]
    with contextlib.redirect_stdout(io.StringIO()):
        t0 = time.perf_counter(); d = voc_dets.get_dets_by_cls(mgr, det, [1.0] * len(imgs), imgs, det_threshold=0.05); dt = time.perf_counter() - t0
    return d, dt
for name, depth, dtype, mapping, scales in (("f32 ResNet-50", 50, "f32", VOC_CLASS_MAPPING, [128, 256, 512]), ("bf16 ResNet-101", 101, "bf16", KITTI_CLASS_MAPPING, [16, 32, 64, 128, 256, 512])):
    anchors = util.get_anchors(scales)
    w = synthetic_resnet(depth, anchors_per_loc=len(anchors), num_classes=len(mapping), seed=1)
    base = (resnet.resnet50_base if depth == 50 else resnet.resnet101_base)(weights=w, dtype=dtype)
    rpn = (resnet.resnet50_rpn if depth == 50 else resnet.resnet101_rpn)(base, include_conv=True, anchors_per_loc=len(anchors))
    det = (resnet.resnet50_classifier if depth == 50 else resnet.resnet101_classifier)(64, len(mapping), weights=w, dtype=dtype)
    mgr = DetTrainingManager(rpn_model=rpn, class_mapping=mapping, preprocess_func=resnet.preprocess, anchor_dims=anchors)
    # runs of equal sizes (batched passes for the bf16 models) mixed with single frames
    order = []
    while len(order) < 240:
        s = sizes[rs.randint(len(sizes))]
        for _ in range(rs.choice([1, 1, 2, 5, 9, 17])):
            order.append((s, rs.randint(3)))
    order = order[:240]
    a, ta = walk(mgr, det, order)
    b, tb = walk(mgr, det, order)
    same = list(a) == list(b) and all(list(a[c]) == list(b[c]) and all(len(a[c][i]) == len(b[c][i]) and all(np.array_equal(x["bbox"], y["bbox"]) and x["prob"] == y["prob"] for x, y in zip(a[c][i], b[c][i])) for i in a[c]) for c in a)
    eng = entry.for_models(mgr, det, 64, 16, entry.default_in_flight(dtype))
    st = eng.stats()
    print("%s: 240 frames %.2f s then %.2f s; identical walks: %s; graphs %d, sizes %d, captures %d, evictions %d, hits %d, images per pass %d" % (
        name, ta, tb, same, st["graphs"], st["sizes"], st["captures"], st["evictions"], st["hits"], st["images_per_pass"]), flush=True)
    assert same
    del mgr, det, rpn, base
print("soak ok")
