"""dev: cProfile of voc_dets.get_dets_by_cls over 32 synthetic 600x1000 frames on the captured path (where the host time per image goes)."""
import cProfile, contextlib, io, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np, torch
import bench
from faster_rcnn_amd import resnet, shapes, voc_dets, ops
from faster_rcnn_amd.data.voc_data_helpers import VOC_CLASS_MAPPING
from faster_rcnn_amd.det_util import DetTrainingManager
ops.F32_ENGINE = "bf16x6"
pipe, w, anchors = bench.build_pipeline()
mgr = DetTrainingManager(rpn_model=pipe.rpn, class_mapping=VOC_CLASS_MAPPING, preprocess_func=resnet.preprocess, anchor_dims=anchors)
rs = np.random.RandomState(2000)
images = [shapes.Image(shapes.Metadata("synth%03d" % i, bench.WIDTH, bench.HEIGHT, [], "none"), rs.randint(0, 256, (bench.HEIGHT, bench.WIDTH, 3)).astype(np.uint8)) for i in range(32)]
ratios = [1.0] * 32
sink = io.StringIO()
def run():
    with contextlib.redirect_stdout(sink):
        t0 = time.perf_counter(); d = voc_dets.get_dets_by_cls(mgr, pipe.det, ratios, images); return time.perf_counter() - t0, d
run(); run()
ts = [run()[0] for _ in range(3)]
print("get_dets_by_cls over 32 frames: %s ms -> %.1f img/s" % (["%.1f" % (t * 1e3) for t in ts], 32 / min(ts)))
pr = cProfile.Profile(); pr.enable(); run(); run(); pr.disable()
st = pstats.Stats(pr, stream=sys.stdout); st.sort_stats("tottime").print_stats(22)
