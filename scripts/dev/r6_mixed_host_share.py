#!/usr/bin/env python3
"""cProfile of voc_dets.get_dets_by_cls over bench.py's shuffled mixed-size list (second call: passes cached).  Dev tool."""
import cProfile, contextlib, io, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from faster_rcnn_amd import entry, resnet, shapes, util, voc_dets
from faster_rcnn_amd.data.voc_data_helpers import VOC_CLASS_MAPPING
from faster_rcnn_amd.det_util import DetTrainingManager
pipe, w, anchors = bench.build_pipeline()
mgr = DetTrainingManager(rpn_model=pipe.rpn, class_mapping=VOC_CLASS_MAPPING, preprocess_func=resnet.preprocess, anchor_dims=anchors)
rs = np.random.RandomState(77)
n = 256
sizes = []
for (w_, h_), share in bench.MIXED_SIZES:
    sizes += [(w_, h_)] * int(round(share * n))
while len(sizes) < n:
    sizes.append((500, int(rs.randint(250, 500))))
sizes = sizes[:n]
rs.shuffle(sizes)
pool, raw = {}, []
for i, (w_, h_) in enumerate(sizes):
    if (w_, h_) not in pool:
        pool[(w_, h_)] = rs.randint(0, 256, (h_, w_, 3)).astype(np.uint8)
    raw.append(shapes.Image(shapes.Metadata("mixed%03d" % i, w_, h_, [], "none"), pool[(w_, h_)]))
images, ratios = util.resize_imgs(raw, min_size=600, max_size=1000)
sink = io.StringIO()
def run():
    with contextlib.redirect_stdout(sink):
        t0 = time.perf_counter(); d = voc_dets.get_dets_by_cls(mgr, pipe.det, ratios, images); return time.perf_counter() - t0, d
run(); run()
ts = [run()[0] for _ in range(3)]
print("mixed list, %d frames: %s ms -> %.1f img/s" % (n, ["%.1f" % (t * 1e3) for t in ts], n / min(ts)))
pr = cProfile.Profile(); pr.enable(); t, _ = run(); pr.disable()
print("profiled call: %.1f ms" % (t * 1e3))
st = pstats.Stats(pr, stream=sys.stdout); st.sort_stats("tottime").print_stats(16)
