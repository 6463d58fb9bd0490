#!/usr/bin/env python3
"""Where does voc_dets.get_dets_by_cls spend the wall clock of a 256-frame list of one geometry (default engine, 4 x 4 in flight)?
cProfile top functions; the time inside Event.synchronize is the time the host WAITS for the device.  Dev tool."""
import cProfile, contextlib, io, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from faster_rcnn_amd import resnet, shapes, voc_dets
from faster_rcnn_amd.data.voc_data_helpers import VOC_CLASS_MAPPING
from faster_rcnn_amd.det_util import DetTrainingManager
pipe, w, anchors = bench.build_pipeline()
mgr = DetTrainingManager(rpn_model=pipe.rpn, class_mapping=VOC_CLASS_MAPPING, preprocess_func=resnet.preprocess, anchor_dims=anchors)
rs = np.random.RandomState(2000)
base = [rs.randint(0, 256, (bench.HEIGHT, bench.WIDTH, 3)).astype(np.uint8) for i in range(32)]
images = [shapes.Image(shapes.Metadata("synth%03d" % i, bench.WIDTH, bench.HEIGHT, [], "none"), base[i % 32]) for i in range(256)]
ratios = [1.0] * len(images)
if len(sys.argv) > 1 and sys.argv[1] == "files":              # the VOC frame as voc_dets.main feeds it: a 500x375 JPEG on disk, resized to 800x600 on the device
    from faster_rcnn_amd import util
    from faster_rcnn_amd.data.voc_data_helpers import extract_img_data
    images, ratios = [], []
    for i in range(128):
        b0 = extract_img_data(os.path.join(ROOT, "tests", "golden", "VOC_test"), "000005")
        (r,), (ratio,) = util.resize_imgs([b0], min_size=600, max_size=1000)
        r.metadata.name = "file%03d" % i
        images.append(r); ratios.append(ratio)
sink = io.StringIO()
def run():
    with contextlib.redirect_stdout(sink):
        t0 = time.perf_counter(); d = voc_dets.get_dets_by_cls(mgr, pipe.det, ratios, images); return time.perf_counter() - t0, d
run(); run()
ts = [run()[0] for _ in range(3)]
print("get_dets_by_cls over %d frames: %s ms -> %.1f img/s" % (len(images), ["%.1f" % (t * 1e3) for t in ts], len(images) / min(ts)))
pr = cProfile.Profile(); pr.enable(); t, _ = run(); pr.disable()
print("profiled call: %.1f ms" % (t * 1e3))
st = pstats.Stats(pr, stream=sys.stdout); st.sort_stats("tottime").print_stats(18)
