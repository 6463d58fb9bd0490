"""dev: cProfile of voc_dets.get_dets_by_cls over 32 copies of the VOC test JPEG (500x375 -> 800x600) on the captured path."""
import cProfile, contextlib, io, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np, torch
import bench
from faster_rcnn_amd import resnet, shapes, voc_dets, ops, util
from faster_rcnn_amd.data.voc_data_helpers import VOC_CLASS_MAPPING, extract_img_data
from faster_rcnn_amd.det_util import DetTrainingManager
ops.F32_ENGINE = "bf16x6"
pipe, w, anchors = bench.build_pipeline()
mgr = DetTrainingManager(rpn_model=pipe.rpn, class_mapping=VOC_CLASS_MAPPING, preprocess_func=resnet.preprocess, anchor_dims=anchors)
base = extract_img_data(os.path.join(ROOT, "tests", "golden", "VOC_test"), "000005")
frames = []
for i in range(32):
    (r,), (ratio,) = util.resize_imgs([base], min_size=600, max_size=1000)
    r.metadata.name = "file%03d" % i
    frames.append(r)
fr = [ratio] * 32
sink = io.StringIO()
def run():
    with contextlib.redirect_stdout(sink):
        t0 = time.perf_counter(); d = voc_dets.get_dets_by_cls(mgr, pipe.det, fr, frames); return time.perf_counter() - t0, d
run(); run()
ts = [run()[0] for _ in range(3)]
print("get_dets_by_cls over 32 files: %s ms -> %.1f img/s" % (["%.1f" % (t * 1e3) for t in ts], 32 / min(ts)))
t0 = time.perf_counter()
for f in frames[:8]: f.raw
print("PIL decode of one frame on the calling thread: %.2f ms" % ((time.perf_counter() - t0) / 8 * 1e3))
pr = cProfile.Profile(); pr.enable(); run(); run(); pr.disable()
st = pstats.Stats(pr, stream=sys.stdout); st.sort_stats("tottime").print_stats(16)
for nt in (0, 2, 4, 8):
    voc_dets.DECODE_THREADS = nt
    run()
    ts = [run()[0] for _ in range(3)]
    print("decode threads %d: %.1f img/s" % (nt, 32 / min(ts)))
import threading
print("threads alive:", threading.active_count())
