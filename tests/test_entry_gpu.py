"""The reference's inference entry points (voc_dets.get_dets / get_dets_by_cls, voc_dets.py:20-111) on the captured
device pipeline (faster_rcnn_amd/entry.py) against the same calls on the eager path, which sequences the stages exactly
as the reference does (get_det_inputs -> numpy -> detector.predict) and is itself held to the oracle / the goldens in
test_pipeline_gpu.py and test_configs_full_size_gpu.py.

Bar: classes and boxes identical, list order identical, scores within 1e-4 (north_star's bar for class scores).  A
captured pass replays the launch forms chosen for its in-flight depth and the eager path those of an eager launch: another
split-K partition of a small-grid layer, or a head GEMM over 64 instead of 320 RoIs, rounds differently in the last bit,
and the fixture's classifier is re-scaled so that many classes fire, which multiplies logit differences by its gain
(measured: <= 3e-5 between eight-in-flight graphs and eager launches, <= 3e-6 between a 64-RoI and a 320-RoI head pass).  With the same launch forms the results are bit-identical (tol=0.0 where the tests re-run a path)."""
import contextlib
import io
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def models():
    from faster_rcnn_amd import resnet, util
    from faster_rcnn_amd.data.voc_data_helpers import VOC_CLASS_MAPPING
    from faster_rcnn_amd.det_util import DetTrainingManager
    from faster_rcnn_amd.pipeline import InferencePipeline
    from faster_rcnn_amd.weights import calibrate_classifier, synthetic_resnet
    anchors = util.get_anchors([128, 256, 512])
    w = synthetic_resnet(50, anchors_per_loc=9, num_classes=21, seed=1)
    rpn = resnet.resnet50_rpn(resnet.resnet50_base(weights=w), include_conv=True, anchors_per_loc=9)
    det = resnet.resnet50_classifier(64, 21, weights=w)
    # random-init heads put every RoI in one class: re-centre dense_class on a calibration frame so that many classes fire
    x = resnet.preprocess(synth_pixels(320, 480, 99))[None].astype(np.float32)
    out = InferencePipeline(rpn, det, anchors).forward_dev(torch.from_numpy(x).cuda())
    n = int(out["n_rois"].item())
    det.get_layer("dense_class_21").set_weights(calibrate_classifier(w, 21, out["cls"][:n].cpu().numpy()))
    mgr = DetTrainingManager(rpn_model=rpn, class_mapping=VOC_CLASS_MAPPING, preprocess_func=resnet.preprocess, anchor_dims=anchors)
    return mgr, det, rpn, w


def named_image(name, pixels):
    from faster_rcnn_amd import shapes
    h, w = pixels.shape[:2]
    return shapes.Image(shapes.Metadata(name, w, h, [], "none"), pixels)


def synth_pixels(h, w, seed):
    return np.random.RandomState(seed).randint(0, 256, (h, w, 3)).astype(np.uint8)


def voc_frame(resized):
    from faster_rcnn_amd import util
    from faster_rcnn_amd.data.voc_data_helpers import extract_img_data
    img = extract_img_data(os.path.join(ROOT, "tests", "golden", "VOC_test"), "000005")
    if not resized:
        return img, 1.0
    (r,), (ratio,) = util.resize_imgs([img], min_size=600, max_size=1000)
    return r, ratio


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()) as buf:
        res = fn(*a, **k)
    return res, buf.getvalue()


def both_paths(fn, *a, **k):
    from faster_rcnn_amd import voc_dets
    fast, out_fast = quiet(fn, *a, **k)
    voc_dets.FAST_ENTRY = False
    try:
        eager, out_eager = quiet(fn, *a, **k)
    finally:
        voc_dets.FAST_ENTRY = True
    return fast, eager, out_fast, out_eager


def same_dets(a, b, tol=1e-4):
    assert len(a) == len(b)
    for x, y in zip(a, b):
        assert x["cls_name"] == y["cls_name"] and np.array_equal(x["bbox"], y["bbox"]), (x, y)
        assert x["bbox"].dtype == y["bbox"].dtype and type(x["prob"]) is type(y["prob"])
        assert abs(float(x["prob"]) - float(y["prob"])) <= tol, (x, y)


def test_get_dets_three_sizes_match_the_eager_path(models):
    """600x1000 synthetic, the real VOC frame resized to 600x800, the same frame un-resized (375x500)."""
    from faster_rcnn_amd import entry, voc_dets
    mgr, det, _, _ = models
    cases = [(named_image("synth", synth_pixels(600, 1000, 7)), 1.0)]
    cases += [voc_frame(True), voc_frame(False)]
    classes = set()
    for image, ratio in cases:
        for thr in (0.0, 0.3):
            fast, eager, out_fast, out_eager = both_paths(voc_dets.get_dets, mgr, det, image, ratio, det_threshold=thr)
            assert len(eager) > 0 or thr > 0
            same_dets(fast, eager)
            classes |= {d["cls_name"] for d in eager}
            assert out_fast == out_eager and out_fast.startswith("num rois: ")           # the reference's progress line, same count
    eng = entry.for_models(mgr, det, 64, 16, 1)
    st = eng.stats()
    assert st["captures"] == 3 and st["sizes"] == 3 and st["hits"] == 3 and st["device_preprocess"]
    assert len(classes) >= 5, classes                             # the calibrated head is not degenerate


def test_get_dets_by_cls_pipelined_equals_one_by_one(models, monkeypatch):
    """A list of mixed sizes with several images in flight: the same dict, keys and per-image lists in the same order.
    (CAPTURE_MIN = 1: every geometry is captured, as before round 6; the rare-geometry policy has a test of its own below.)"""
    from faster_rcnn_amd import entry, voc_dets
    monkeypatch.setattr(voc_dets, "CAPTURE_MIN", 1)
    mgr, det, _, _ = models
    frame, ratio = voc_frame(True)
    images, ratios = [], []
    for i in range(11):
        if i % 4 == 3:
            images.append(named_image("voc%02d" % i, frame.data)); ratios.append(ratio)
        elif i % 4 == 1:
            images.append(named_image("small%02d" % i, synth_pixels(352, 480, 40 + i))); ratios.append(0.8)
        else:
            images.append(named_image("synth%02d" % i, synth_pixels(600, 1000, 20 + i))); ratios.append(1.0 + 0.05 * i)
    fast, eager, out_fast, out_eager = both_paths(voc_dets.get_dets_by_cls, mgr, det, ratios, images, det_threshold=0.1)
    assert list(fast) == list(eager) and len(fast) >= 2
    for cls_name in eager:
        assert list(fast[cls_name]) == list(eager[cls_name])
        for img_name in eager[cls_name]:
            same_dets(fast[cls_name][img_name], eager[cls_name][img_name])
    strip = lambda s: [ln.split(" ran in ")[0] for ln in s.splitlines()]
    assert strip(out_fast) == strip(out_eager)
    depth = entry.default_in_flight("f32")
    eng = entry.for_models(mgr, det, 64, 16, depth)
    st = eng.stats()
    assert st["sizes"] == 3 and st["in_flight"] == depth >= 4
    # six 600x1000 frames, three 352x480, two VOC frames, gathered per geometry into four-image passes: 4 + 2 (padded), 3 (padded), 2 (padded)
    assert st["captures"] <= 3 * depth and st["hits"] + st["captures"] == 4
    # a second walk over the list re-uses every captured pass and returns the same bits
    again, _ = quiet(voc_dets.get_dets_by_cls, mgr, det, ratios, images, det_threshold=0.1)
    assert eng.stats()["captures"] == st["captures"]
    for cls_name in fast:
        for img_name in fast[cls_name]:
            same_dets(again[cls_name][img_name], fast[cls_name][img_name], tol=0.0)


def test_slow_image_sources_are_fetched_on_threads_with_the_same_result(models):
    """get_dets_by_cls decodes inline until two fetches in a row take longer than DECODE_INLINE_MS; from then on the next images'
    pixels come from a few threads.  Same dict, same order, same bits either way."""
    import threading
    import time
    from faster_rcnn_amd import shapes, voc_dets
    mgr, det, _, _ = models
    seen = set()

    class SlowImage(shapes.Image):
        @property
        def raw(self):
            seen.add(threading.current_thread().name)
            time.sleep(0.006)
            return self._pixels
    images = [SlowImage(shapes.Metadata("slow%02d" % i, 480, 352, [], "none"), synth_pixels(352, 480, 60 + i)) for i in range(9)]
    ratios = [1.0] * len(images)
    threaded, _ = quiet(voc_dets.get_dets_by_cls, mgr, det, ratios, images, det_threshold=0.1)
    assert len(seen) >= 2, seen                                   # the calling thread first, pool threads later
    prev, seen_inline = voc_dets.DECODE_THREADS, None
    voc_dets.DECODE_THREADS = 0
    try:
        seen.clear()
        inline, _ = quiet(voc_dets.get_dets_by_cls, mgr, det, ratios, images, det_threshold=0.1)
        seen_inline = set(seen)
    finally:
        voc_dets.DECODE_THREADS = prev
    assert seen_inline == {threading.current_thread().name}
    assert list(threaded) == list(inline)
    for cls_name in inline:
        assert list(threaded[cls_name]) == list(inline[cls_name])
        for img_name in inline[cls_name]:
            same_dets(threaded[cls_name][img_name], inline[cls_name][img_name], tol=0.0)


def test_bf16_models_through_both_entry_paths(monkeypatch):
    """configs[3]'s models (ResNet-101, bf16 conv, KITTI classes) through voc_dets.get_dets_by_cls: the captured path, and the eager
    path whose conv map travels to the host and back as float32 numpy (numpy has no bf16: the widening is exact and
    DetModel.forward_dev narrows it back) -- same detections."""
    from faster_rcnn_amd import resnet, util, voc_dets
    from faster_rcnn_amd.data.voc_data_helpers import KITTI_CLASS_MAPPING
    from faster_rcnn_amd.det_util import DetTrainingManager
    from faster_rcnn_amd.weights import synthetic_resnet
    monkeypatch.setattr(voc_dets, "CAPTURE_MIN", 1)                 # (every geometry is captured: the rare-geometry policy has its own test)
    anchors = util.get_anchors([16, 32, 64, 128, 256, 512])
    w = synthetic_resnet(101, anchors_per_loc=len(anchors), num_classes=len(KITTI_CLASS_MAPPING), seed=1)
    rpn = resnet.resnet101_rpn(resnet.resnet101_base(weights=w, dtype="bf16"), include_conv=True, anchors_per_loc=len(anchors))
    det = resnet.resnet101_classifier(64, len(KITTI_CLASS_MAPPING), weights=w, dtype="bf16")
    mgr = DetTrainingManager(rpn_model=rpn, class_mapping=KITTI_CLASS_MAPPING, preprocess_func=resnet.preprocess, anchor_dims=anchors)
    # thirteen frames of one size then two of another: one whole eight-image pass, one padded pass of five (>= half a batch), two single passes
    images = [named_image("k%02d" % i, synth_pixels(320, 480 if i < 13 else 544, 70 + i)) for i in range(15)]
    ratios = [1.0 + 0.01 * i for i in range(len(images))]
    from oracle.e2e import match_detections
    fast, eager, out_fast, out_eager = both_paths(voc_dets.get_dets_by_cls, mgr, det, ratios, images, det_threshold=0.0)
    assert list(fast) == list(eager) and sum(len(v) for c in eager.values() for v in c.values()) > 0
    # bf16 activations: the captured passes' launch forms (split-K partitions, a 320-row head pass) round a few pre-activations to the
    # neighbouring bf16 value, a box edge moves by a pixel, NMS keeps a neighbour -- the sets are compared the way a scorer pairs boxes
    n_eager = n_fast = n_matched = 0
    worst = 0.0
    for cls_name in eager:
        assert list(fast[cls_name]) == list(eager[cls_name])
        for img_name in eager[cls_name]:
            a = [(cls_name, d["prob"], d["bbox"]) for d in eager[cls_name][img_name]]
            b = [(cls_name, d["prob"], d["bbox"]) for d in fast[cls_name][img_name]]
            pairs = match_detections(a, b, 0.5)
            n_eager += len(a); n_fast += len(b); n_matched += len(pairs)
            worst = max([worst] + [abs(x - y) for x, y, _ in pairs])
    print("bf16 entry: eager %d, captured %d, matched %d, worst score difference %.3g" % (n_eager, n_fast, n_matched, worst))
    assert n_matched >= 0.95 * max(n_eager, n_fast) and worst <= 5e-3      # measured: 1041 of 1079 matched, worst 2.1e-4
    conv_out, rois = mgr.get_det_inputs(images[0])
    assert conv_out.dtype == np.float32 and rois.dtype == np.int16 and conv_out.shape[-1] == 1024
    from faster_rcnn_amd import entry
    st = entry.for_models(mgr, det, 64, 16, entry.default_in_flight("bf16")).stats()
    assert st["images_per_pass"] == 8 and st["sizes"] == 2 and st["captures"] == 4, st       # two eight-image passes of (320, 480), two single passes of (320, 544)
    # strip the timing from the reference's progress lines: same lines, same order
    strip = lambda s_: [ln.split(" ran in ")[0] for ln in s_.splitlines() if not ln.startswith("num rois")]
    assert strip(out_fast) == strip(out_eager)
    # an image that cannot be submitted, in the middle of a batched run: the error surfaces, no pass stays marked busy, the next call works
    bad = named_image("bad", synth_pixels(320, 480, 3).astype(np.float32))
    with pytest.raises(TypeError):
        quiet(voc_dets.get_dets_by_cls, mgr, det, [1.0] * 12, images[:10] + [bad, images[11]])
    eng = entry.for_models(mgr, det, 64, 16, entry.default_in_flight("bf16"))
    assert not any(sl.busy for v in eng.cache._slots.values() for sl in v)
    again, _ = quiet(voc_dets.get_dets_by_cls, mgr, det, ratios[:9], images[:9])
    for cls_name in again:
        for img_name in again[cls_name]:
            same_dets(again[cls_name][img_name], fast[cls_name][img_name], tol=0.0)       # a whole pass of eight + one single pass: the same bits as before


def test_graph_cache_budget_evicts_least_recently_used(models):
    from faster_rcnn_amd import entry
    mgr, det, _, _ = models
    eng = entry.DetectionEntry(mgr, det, 64, 16, in_flight=1, byte_budget=1)        # nothing fits: every new size evicts the idle ones
    sizes = [(160, 224), (176, 240), (160, 224), (192, 256)]
    ref = {}
    for k, (h, w) in enumerate(sizes):
        img = named_image("s%d" % k, synth_pixels(h, w, 90 + h))
        n, dets = eng.collect(eng.submit(img, 1.0, 0.0))
        if (h, w) in ref:
            same_dets(dets, ref[(h, w)], tol=0.0)                                   # a re-captured size returns the same bits
        ref[(h, w)] = dets
        assert len(eng.cache) == 1 and eng.cache.keys() == [(h, w)]
    st = eng.stats()
    assert st["captures"] == 4 and st["evictions"] == 3 and st["hits"] == 0
    big = entry.DetectionEntry(mgr, det, 64, 16, in_flight=1, byte_budget=1 << 40)
    for k, (h, w) in enumerate(sizes):
        big.collect(big.submit(named_image("s%d" % k, synth_pixels(h, w, 90 + h)), 1.0, 0.0))
    assert big.stats()["captures"] == 3 and big.stats()["hits"] == 1 and big.stats()["evictions"] == 0
    assert 0 < big.stats()["bytes"] < 8 << 30


def test_changed_weights_drop_the_captured_passes(models):
    from faster_rcnn_amd import voc_dets
    from faster_rcnn_amd import entry
    mgr, det, rpn, w = models
    image = named_image("w", synth_pixels(192, 256, 5))
    before, _ = quiet(voc_dets.get_dets, mgr, det, image, 1.0)
    eng = entry.for_models(mgr, det, 64, 16, 1)
    caps = eng.stats()["captures"]
    name = "dense_class_21"
    old = det.get_layer(name).get_weights()
    try:
        det.get_layer(name).set_weights([old[0][:, ::-1].copy(), old[1][::-1].copy()])     # classes reversed
        fast, eager, _, _ = both_paths(voc_dets.get_dets, mgr, det, image, 1.0)
        same_dets(fast, eager)
        assert eng.stats()["captures"] == caps + 1
        assert [d["cls_name"] for d in fast] != [d["cls_name"] for d in before]
    finally:
        det.get_layer(name).set_weights(old)
    after, _ = quiet(voc_dets.get_dets, mgr, det, image, 1.0)
    same_dets(after, before, tol=0.0)


def test_foreign_preprocess_and_foreign_detector(models):
    from faster_rcnn_amd import entry, resnet, voc_dets
    from faster_rcnn_amd.det_util import DetTrainingManager
    mgr, det, rpn, _ = models
    image = named_image("f", synth_pixels(192, 256, 6))
    want, _ = quiet(voc_dets.get_dets, mgr, det, image, 1.25)
    # a preprocess function the entry does not know: called on the host (det_util.py:36), its float image uploaded
    mgr2 = DetTrainingManager(rpn_model=rpn, class_mapping=mgr.class_mapping, preprocess_func=lambda d: resnet.preprocess(d), anchor_dims=mgr.anchor_dims)
    got, _ = quiet(voc_dets.get_dets, mgr2, det, image, 1.25)
    same_dets(got, want, tol=0.0)
    assert not entry.for_models(mgr2, det, 64, 16, 1).stats()["device_preprocess"]

    class Foreign:                                  # the Keras predict() contract only: the eager path serves it
        def __init__(self):
            self.calls = 0

        def predict(self, inputs):
            self.calls += 1
            return det.predict(inputs)

    f = Foreign()
    assert entry.for_models(mgr, f, 64, 16, 1) is None
    got, printed = quiet(voc_dets.get_dets, mgr, f, image, 1.25)
    n_rois = int(printed.split("num rois: ")[1].split()[0])
    assert f.calls == -(-n_rois // 64) >= 2          # one predict() per batch of 64 RoIs (voc_dets.py:31-49)
    same_dets(got, want)


def test_a_failed_image_leaves_no_slot_busy(models):
    from faster_rcnn_amd import entry, voc_dets
    mgr, det, _, _ = models
    good = named_image("g", synth_pixels(160, 224, 1))
    bad = named_image("b", synth_pixels(160, 224, 2).astype(np.float32))           # not uint8: submit refuses
    with pytest.raises(TypeError):
        quiet(voc_dets.get_dets_by_cls, mgr, det, [1.0, 1.0, 1.0], [good, good, bad])
    eng = entry.for_models(mgr, det, 64, 16, entry.default_in_flight("f32"))
    assert not any(s.busy for v in eng.cache._slots.values() for s in v)
    res, _ = quiet(voc_dets.get_dets_by_cls, mgr, det, [1.0, 1.0], [good, good])
    assert res


def test_device_cubic_resize_is_the_host_resize_bit_for_bit():
    """shapes.Image.data = cv2.resize(..., INTER_CUBIC) restated in integer numpy (shapes._resize, shapes.py:19-29 of the
    reference); frcnn_resize_cubic_u8 must reproduce it exactly: enlarging, shrinking, odd sizes, saturation at both ends,
    and the horizontal flip Image.horizontal_flip adds."""
    from faster_rcnn_amd import ops, shapes
    rs = np.random.RandomState(1)
    cases = [((375, 500), (600, 800)), ((333, 500), (600, 901)), ((500, 375), (800, 600)), ((600, 800), (375, 500)), ((37, 53), (61, 40)), ((9, 7), (7, 23))]
    for (sh, sw), (dh, dw) in cases:
        for kind in ("noise", "extremes"):
            src = rs.randint(0, 256, (sh, sw, 3)).astype(np.uint8)
            if kind == "extremes":                                   # blocks of 0 / 255: the cubic overshoots, the result saturates
                src = (rs.randint(0, 2, (sh // 3 + 1, sw // 3 + 1, 3)) * 255).astype(np.uint8).repeat(3, axis=0).repeat(3, axis=1)[:sh, :sw]
                src = np.ascontiguousarray(src)
            want = shapes._resize(src, dw, dh)
            got = ops.resize_cubic_u8(torch.from_numpy(src).cuda(), dh, dw)
            assert got.shape == (dh, dw, 3) and np.array_equal(got.cpu().numpy(), want), ((sh, sw), (dh, dw), kind)
            flipped = ops.resize_cubic_u8(torch.from_numpy(src).cuda(), dh, dw, flip=True, out=torch.full((dh, dw, 3), 7, dtype=torch.uint8, device="cuda"))
            assert np.array_equal(flipped.cpu().numpy(), want[:, ::-1])


def test_file_backed_images_are_resized_on_the_device_and_match_the_eager_path(models):
    """The VOC frame as voc_dets.main feeds it: a JPEG on disk, metadata resized to 600x800.  The captured pass uploads the
    decoded 375x500 frame and resizes it on the device; the eager path resizes on the host (shapes._resize).  Same pixels, so
    the same detections; a flipped copy (train-time augmentation, shapes.py:27) likewise."""
    from faster_rcnn_amd import entry, voc_dets
    mgr, det, _, _ = models
    image, ratio = voc_frame(True)
    fast, eager, _, _ = both_paths(voc_dets.get_dets, mgr, det, image, ratio)
    same_dets(fast, eager)
    eng = entry.for_models(mgr, det, 64, 16, 1)
    assert (600, 800, 375, 500, 2) in eng.cache.keys()               # (flip bit set: 2 = uploaded as RGB, swapped on the device)
    flipped = image.horizontal_flip()
    fast, eager, _, _ = both_paths(voc_dets.get_dets, mgr, det, flipped, ratio)
    same_dets(fast, eager)
    assert (600, 800, 375, 500, 3) in eng.cache.keys()
    # a list of files through get_dets_by_cls: pixels fetched ahead on threads, results in list order
    frames = [voc_frame(True)[0] for _ in range(6)]
    for i, f in enumerate(frames):
        f.metadata.name = "f%d" % i
    by_cls, _ = quiet(voc_dets.get_dets_by_cls, mgr, det, [ratio] * 6, frames)
    one, _ = quiet(voc_dets.get_dets, mgr, det, frames[0], ratio)
    for cls_name, per_img in by_cls.items():
        assert list(per_img) == ["f%d" % i for i in range(6) if "f%d" % i in per_img]
        for name, dets in per_img.items():
            same_dets(dets, [d for d in one if d["cls_name"] == cls_name], tol=1e-4)


def test_get_dets_by_cls_fp32_four_image_passes(models, monkeypatch):
    """Neighbouring fp32 frames of one size go through four-image captured passes (entry.default_batch("f32"); round 5): ten frames of
    one size and two of another -> two whole passes, one padded pass of two (half a batch), two single passes.  Same dict, same
    order and progress lines as the eager one-by-one path; scores to 1e-4, classes and boxes identical."""
    from faster_rcnn_amd import entry, voc_dets
    monkeypatch.setattr(voc_dets, "CAPTURE_MIN", 1)
    mgr, det, _, _ = models
    images = [named_image("f%02d" % i, synth_pixels(320, 480 if i < 10 else 544, 170 + i)) for i in range(12)]
    ratios = [1.0 + 0.01 * i for i in range(len(images))]
    fast, eager, out_fast, out_eager = both_paths(voc_dets.get_dets_by_cls, mgr, det, ratios, images, det_threshold=0.0)
    assert list(fast) == list(eager) and sum(len(v) for c in eager.values() for v in c.values()) > 0
    for cls_name in eager:
        assert list(fast[cls_name]) == list(eager[cls_name])
        for img_name in eager[cls_name]:
            same_dets(fast[cls_name][img_name], eager[cls_name][img_name])
    strip = lambda s_: [ln.split(" ran in ")[0] for ln in s_.splitlines()]
    assert strip(out_fast) == strip(out_eager)
    eng = entry.for_models(mgr, det, 64, 16, entry.default_in_flight("f32"))
    st = eng.stats()
    assert st["images_per_pass"] == 4, st
    # the same list again: every pass is a cache hit and returns the same bits
    again, _ = quiet(voc_dets.get_dets_by_cls, mgr, det, ratios, images, det_threshold=0.0)
    for cls_name in again:
        for img_name in again[cls_name]:
            same_dets(again[cls_name][img_name], fast[cls_name][img_name], tol=0.0)
    assert not any(sl.busy for v in eng.cache._slots.values() for sl in v)


def test_get_dets_by_cls_shuffled_geometries_share_passes_and_rare_ones_run_eagerly(models, monkeypatch):
    """Round 6 (VERDICT r5 missing #4): a SHUFFLED list of four geometries.  Images of one geometry share four-image passes wherever they
    stand in the list (held back per geometry); the geometry the list holds only twice is not captured -- its images take the eager
    sequence; the dict, its key order, the per-image lists and the progress lines are those of the one-by-one eager walk over the same
    list; a second call re-uses every pass (no new capture) and returns the same bits."""
    from faster_rcnn_amd import entry, voc_dets
    monkeypatch.setattr(voc_dets, "CAPTURE_MIN", 3)
    mgr, det, _, _ = models
    dims = [(320, 480)] * 9 + [(352, 480)] * 6 + [(320, 512)] * 3 + [(288, 448)] * 2
    order = np.random.RandomState(3).permutation(len(dims))
    images = [named_image("m%02d" % k, synth_pixels(dims[j][0], dims[j][1], 700 + int(j))) for k, j in enumerate(order)]
    ratios = [1.0 + 0.01 * k for k in range(len(images))]
    depth = entry.default_in_flight("f32")
    eng = entry.for_models(mgr, det, 64, 16, depth)
    eng.cache.clear()
    before = eng.stats()["captures"]
    eager_seen = []
    real = voc_dets._get_dets_eager
    monkeypatch.setattr(voc_dets, "_get_dets_eager", lambda *a, **k: (eager_seen.append(a[2].name), real(*a, **k))[1])
    fast, out_fast = quiet(voc_dets.get_dets_by_cls, mgr, det, ratios, images, det_threshold=0.1)
    rare = sorted(im.name for im in images if (im.height, im.width) == (288, 448))
    assert sorted(eager_seen) == rare
    st = eng.stats()
    keys = eng.cache.keys()
    assert not any(k[:2] == (288, 448) for k in keys)               # the rare geometry has no captured pass
    assert (320, 480, 4) in keys and (352, 480, 4) in keys          # the two common ones went through four-image passes
    n_eager = len(eager_seen)
    eager_seen.clear()
    voc_dets.FAST_ENTRY = False
    try:
        eager, out_eager = quiet(voc_dets.get_dets_by_cls, mgr, det, ratios, images, det_threshold=0.1)
    finally:
        voc_dets.FAST_ENTRY = True
    assert list(fast) == list(eager)
    for cls_name in eager:
        assert list(fast[cls_name]) == list(eager[cls_name])
        for img_name in eager[cls_name]:
            same_dets(fast[cls_name][img_name], eager[cls_name][img_name])
    strip = lambda s_: [ln.split(" ran in ")[0] for ln in s_.splitlines()]
    assert strip(out_fast) == strip(out_eager)
    eager_seen.clear()
    again, _ = quiet(voc_dets.get_dets_by_cls, mgr, det, ratios, images, det_threshold=0.1)
    assert eng.stats()["captures"] == st["captures"] and len(eager_seen) == n_eager
    for cls_name in fast:
        for img_name in fast[cls_name]:
            same_dets(again[cls_name][img_name], fast[cls_name][img_name], tol=0.0)
    assert not any(sl.busy for v in eng.cache._slots.values() for sl in v)


def test_get_dets_by_cls_many_sizes_go_through_canvas_passes(models, monkeypatch):
    """Round 6: a list of MORE than entry.CANVAS_MIN_GEOMETRIES image sizes is served by passes captured per canvas CLASS (even sides,
    an odd side sits at offset 1; true sizes as device values; the classes PLANNED from the list's histogram of sizes) -- here nine
    sizes, even and odd, resized and un-resized frames, two images each, end on at most two canvases instead of nine geometries.  The
    dict, its order and the progress lines are those of the eager one-by-one walk; classes and boxes identical, scores to 1e-4; the
    same call again plans the same classes, re-uses the passes and returns the same bits."""
    from faster_rcnn_amd import entry, shapes, voc_dets
    monkeypatch.setattr(voc_dets, "CAPTURE_MIN", 1)
    mgr, det, _, _ = models
    sizes = [(320, 480), (318, 470), (306, 452), (320, 466), (310, 480), (352, 480), (340, 472), (289, 449), (273, 447)]      # (odd sides among them)
    images, ratios = [], []
    rs = np.random.RandomState(21)
    for k in range(18):
        h, w = sizes[k % len(sizes)]
        if k % 3 == 0:                                            # a frame that is resized on the way in (source smaller than its (h, w))
            src = synth_pixels(h * 5 // 8, w * 5 // 8, 800 + k)
            images.append(shapes.Image(shapes.Metadata("c%02d" % k, w, h, [], "none"), src))
        else:
            images.append(named_image("c%02d" % k, synth_pixels(h, w, 800 + k)))
        ratios.append(1.0 + 0.01 * k)
    order = rs.permutation(len(images))
    images, ratios = [images[i] for i in order], [ratios[i] for i in order]
    depth = entry.default_in_flight("f32")
    eng = entry.for_models(mgr, det, 64, 16, depth)
    eng.cache.clear()
    monkeypatch.setattr(eng, "canvas_capable", True)
    fast, out_fast = quiet(voc_dets.get_dets_by_cls, mgr, det, ratios, images, det_threshold=0.1)
    assert eng.canvas
    keys = eng.cache.keys()
    assert all(k[0] == "canvas" for k in keys), keys
    classes = {k[1:3] for k in keys}
    assert 1 <= len(classes) <= 2, classes                          # two images per size: no size is worth a canvas of its own
    for h, w in sizes:
        hc, wc = eng.canvas_class(h, w)
        assert (hc, wc) in classes and hc % 2 == 0 and wc % 2 == 0 and hc >= h + (h & 1) and wc >= w + (w & 1)
    st = eng.stats()
    voc_dets.FAST_ENTRY = False
    try:
        eager, out_eager = quiet(voc_dets.get_dets_by_cls, mgr, det, ratios, images, det_threshold=0.1)
    finally:
        voc_dets.FAST_ENTRY = True
    assert list(fast) == list(eager)
    # a canvas pass and the eager launches pick different launch forms for some layers (the canvas has a few more rows): scores differ
    # by ~1e-6, so two detections of one class whose scores are closer than that may swap places in the list -- compared by box
    by_box = lambda lst: sorted(lst, key=lambda d: tuple(int(v) for v in d["bbox"]))
    for cls_name in eager:
        assert list(fast[cls_name]) == list(eager[cls_name])
        for img_name in eager[cls_name]:
            same_dets(by_box(fast[cls_name][img_name]), by_box(eager[cls_name][img_name]))
    strip = lambda s_: [ln.split(" ran in ")[0] for ln in s_.splitlines()]
    assert strip(out_fast) == strip(out_eager)
    again, _ = quiet(voc_dets.get_dets_by_cls, mgr, det, ratios, images, det_threshold=0.1)
    assert eng.stats()["captures"] == st["captures"]
    for cls_name in fast:
        for img_name in fast[cls_name]:
            same_dets(again[cls_name][img_name], fast[cls_name][img_name], tol=0.0)
    assert not any(sl.busy for v in eng.cache._slots.values() for sl in v)
    # a short list of few sizes keeps the exact-geometry passes
    few, _ = quiet(voc_dets.get_dets_by_cls, mgr, det, ratios[:3], [named_image("f%d" % i, synth_pixels(320, 480, 900 + i)) for i in range(3)], det_threshold=0.1)
    assert not eng.canvas


def test_rgb_upload_with_device_channel_swap_is_the_bgr_path_bit_for_bit(models, monkeypatch):
    """Round 6: a file-backed frame goes up in the JPEG decoder's channel order and frcnn_resize_cubic_u8 (flip bit 1) writes B, G, R --
    cv2.imread's order (shapes.py:23) -- instead of the host reversing the channels first (~0.8 ms per frame).  The same detections, bit
    for bit, for a frame that is resized (500x375 -> 800x600) and for one that is not (a resize to its own size: taps {0, 1, 0, 0})."""
    from faster_rcnn_amd import entry, ops, voc_dets
    mgr, det, _, _ = models
    # the kernel by itself: swap == reversing first, with and without a horizontal flip, resized and at its own size
    rgb = torch.from_numpy(np.random.RandomState(31).randint(0, 256, (37, 53, 3)).astype(np.uint8)).cuda()
    bgr = rgb.flip(2).contiguous()
    for (dh, dw) in ((60, 80), (37, 53)):
        for fl in (0, 1):
            assert torch.equal(ops.resize_cubic_u8(rgb, dh, dw, flip=2 | fl), ops.resize_cubic_u8(bgr, dh, dw, flip=fl))
    assert torch.equal(ops.resize_cubic_u8(rgb, 37, 53, flip=2), bgr)
    for resized in (True, False):
        image, ratio = voc_frame(resized)
        assert image.raw_rgb is not None and np.array_equal(image.raw_rgb[:, :, ::-1], image.raw)
        res = {}
        for on in (True, False):
            monkeypatch.setattr(entry, "RGB_UPLOAD", on)
            res[on], _ = quiet(voc_dets.get_dets, mgr, det, image, ratio, det_threshold=0.0)
        assert len(res[True]) > 0
        same_dets(res[True], res[False], tol=0.0)
    eng = entry.for_models(mgr, det, 64, 16, 1)
    assert any(len(k) == 5 and k[4] & 2 for k in eng.cache.keys())            # the RGB passes are keyed apart from the BGR ones
