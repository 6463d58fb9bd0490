"""Host logic of the captured inference entry (faster_rcnn_amd/entry.py) that needs no GPU: the per-size graph cache's
recency order, byte budget and busy-slot rule, and which (manager, detector) pairs take the captured path."""
import types

import numpy as np
import pytest

from faster_rcnn_amd import entry


def fake(nbytes):
    s = entry._Slot()
    s.nbytes, s.busy = nbytes, False
    return s


def test_graph_cache_lru_budget_and_busy_slots():
    made = []

    def make(n):
        def f():
            made.append(n)
            return fake(n)
        return f
    c = entry.GraphCache(byte_budget=250)
    a = c.acquire((600, 1000), make(100))
    b = c.acquire((600, 800), make(100))
    assert c.acquire((600, 1000), make(100)) is a and c.hits == 1            # idle slot of that size: reused, now most recent
    assert c.keys() == [(600, 800), (600, 1000)]
    a.busy = True
    a2 = c.acquire((600, 1000), make(100))                                   # same size while the first is in flight: a second instance
    assert a2 is not a and len(c) == 2 and c.evictions == 1                  # 300 > 250: the idle least-recent size (600x800) went
    assert c.keys() == [(600, 1000)] and c.nbytes == 200
    a2.busy = True
    d = c.acquire((375, 500), make(100))                                     # both 600x1000 slots busy: nothing evictable but over budget
    assert len(c) == 3 and c.nbytes == 300 and c.evictions == 1
    a.busy = a2.busy = False
    e = c.acquire((500, 375), make(100))                                     # now the oldest idle ones go until the budget holds
    assert c.nbytes <= 250 and e in [s for v in c._slots.values() for s in v] and d in [s for v in c._slots.values() for s in v]
    assert made == [100] * 5 and c.captures == 5
    c.clear()
    assert len(c) == 0 and c.nbytes == 0
    assert b is not None


def test_only_this_packages_models_take_the_captured_path():
    mgr = types.SimpleNamespace(rpn_model=object(), conv_only=True)
    assert not entry.DetectionEntry.usable(mgr, object(), 64)                # foreign Keras-style models: eager path
    assert entry.for_models(mgr, object()) is None
    assert entry.default_in_flight("bf16") == 4


def test_cubic_tap_tables_from_the_c_abi_equal_the_numpy_restatement():
    """frcnn_resize_cubic_taps is a HOST function of the library (no GPU): OpenCV's f32 tap arithmetic written in C must give
    the integers shapes._cubic_taps derives in numpy, for enlarging, shrinking, odd and degenerate sizes."""
    import numpy as np
    from faster_rcnn_amd import ops, shapes
    rs = np.random.RandomState(0)
    pairs = [(800, 500), (600, 375), (375, 600), (1000, 353), (901, 500), (7, 3), (3, 7), (1, 5), (5, 1), (640, 640)]
    pairs += [(int(rs.randint(1, 1600)), int(rs.randint(1, 1600))) for _ in range(60)]
    for dst, src in pairs:
        tab = ops.resize_cubic_taps(dst, src)
        idx, coef = shapes._cubic_taps(dst, src)
        assert tab.shape == (dst, 8) and np.array_equal(tab[:, :4], idx) and np.array_equal(tab[:, 4:], coef), (dst, src)


class _FakeEngine:
    """Stands in for entry.DetectionEntry in voc_dets.get_dets_by_cls: records what is submitted, returns one detection per image."""

    def __init__(self, batch, in_flight, fetch_ms=0.0):
        self.batch, self.in_flight, self.fetch_ms = batch, in_flight, fetch_ms
        self.submitted, self.fetch_threads, self.open_tickets = [], set(), 0
        self.max_open = 0

    num_rois = 64

    def prefetchable(self, image):
        return True

    def probe_geometry(self, image):
        return tuple(image.size) if getattr(self, "probe", True) else None

    def has_geometry(self, key):
        return key in getattr(self, "cached", ())

    def host_pixels(self, image):
        import threading
        import time
        self.fetch_threads.add(threading.current_thread().name)
        if self.fetch_ms:
            time.sleep(self.fetch_ms / 1e3)
        h, w = image.size
        return (image.name, h, w, None, False)

    @staticmethod
    def geometry(pixels):
        return entry.DetectionEntry.geometry_of(pixels)

    def submit_batch(self, images, ratios, thr, pixels, batch=None):
        assert 1 <= len(images) <= (batch or self.batch) and len({self.geometry(p) for p in pixels}) == 1
        assert [p[0] for p in pixels] == [im.name for im in images]          # each image travels with ITS pixels
        self.submitted.append((batch, [im.name for im in images], list(ratios)))
        self.open_tickets += 1
        self.max_open = max(self.max_open, self.open_tickets)
        return types.SimpleNamespace(slot=types.SimpleNamespace(event=types.SimpleNamespace(synchronize=lambda: None), busy=True), image=list(images))

    def collect_batch(self, ticket):
        self.open_tickets -= 1
        return [(300, [{"bbox": [0, 0, 1, 1], "cls_name": "car" if i % 2 else "person", "prob": 0.5}]) for i, _ in enumerate(ticket.image)]


def _run_by_cls(monkeypatch, eng, sizes, **kw):
    import contextlib
    import io
    from faster_rcnn_amd import voc_dets
    monkeypatch.setattr(voc_dets.entry, "for_models", lambda *a, **k: eng)
    monkeypatch.setattr(voc_dets.entry, "default_in_flight", lambda dtype: eng.in_flight)
    images = [types.SimpleNamespace(name="im%02d" % i, size=s) for i, s in enumerate(sizes)]
    ratios = [1.0 + 0.1 * i for i in range(len(images))]
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        dets = voc_dets.get_dets_by_cls(object(), types.SimpleNamespace(head=None), ratios, images, **kw)
    return images, ratios, dets, buf.getvalue()


def test_get_dets_by_cls_gathers_one_geometry_into_batched_passes(monkeypatch):
    """Images of one geometry share batched passes even when they are NOT neighbours in the list (round 6: they are held back per
    geometry, at most voc_dets.REORDER_WINDOW of them); a padded pass when the rest fills at least half a batch, single passes
    otherwise; never more tickets open than the engine's depth; results folded in LIST order, the reference's lines."""
    A, B = (600, 1500), (600, 1400)
    sizes = [A] * 13 + [B] * 2 + [A] * 3 + [B] * 9
    eng = _FakeEngine(batch=8, in_flight=2)
    images, ratios, dets, out = _run_by_cls(monkeypatch, eng, sizes)
    shape = [(b, len(names)) for b, names, _ in eng.submitted]
    assert shape == [(8, 8), (8, 8), (8, 8), (1, 1), (1, 1), (1, 1)], shape
    by_name = {im.name: (im.size, r) for im, r in zip(images, ratios)}
    for _, names, rs_ in eng.submitted:
        assert len({by_name[n][0] for n in names}) == 1                    # one pass, one geometry
        assert [by_name[n][1] for n in names] == rs_                       # each image with ITS ratio
    assert sorted(n for _, names, _ in eng.submitted for n in names) == sorted(im.name for im in images)
    assert eng.submitted[1][1] == ["im%02d" % i for i in (8, 9, 10, 11, 12, 15, 16, 17)]       # the A's on either side of the two B's
    assert eng.max_open <= 2 and eng.open_tickets == 0
    lines = [ln.split(" ran in ")[0] for ln in out.splitlines()]
    assert lines == [x for im in images for x in ("num rois: 300", "image %s" % im.name)]      # list order, whatever order they ran in
    assert set(dets) == {"person", "car"} and sum(len(v) for c in dets.values() for v in c.values()) == len(images)
    assert eng.fetch_threads == {__import__("threading").current_thread().name}      # fast fetches stay inline


def test_get_dets_by_cls_rare_geometries_run_eagerly_and_the_window_is_bounded(monkeypatch):
    """A geometry the list holds fewer than CAPTURE_MIN times is not captured: its images take the eager sequence (unless a pass of
    that geometry is cached already, or the image's geometry could not be probed).  The window bounds how many images wait."""
    from faster_rcnn_amd import voc_dets
    eager = []
    monkeypatch.setattr(voc_dets, "_get_dets_eager", lambda mgr, det, image, ratio, num_rois, stride, thr: (eager.append((image.name, ratio)) or print("num rois: 300") or
                                                                                                      [{"bbox": [0, 0, 1, 1], "cls_name": "dog", "prob": 0.9}]))
    monkeypatch.setattr(voc_dets, "CAPTURE_MIN", 3)
    A, B, C, D = (600, 800), (600, 901), (600, 898), (500, 600)
    sizes = [A, B, A, C, A, D, A, C, A, B, A, A, A]                        # A x8, B x2, C x2, D x1
    eng = _FakeEngine(batch=4, in_flight=2)
    eng.cached = {C}                                                       # a pass of C exists already: C is served from it
    images, ratios, dets, out = _run_by_cls(monkeypatch, eng, sizes)
    assert sorted(n for n, _ in eager) == ["im01", "im05", "im09"]         # the two B's and the D
    assert all(ratios[int(n[2:])] == r for n, r in eager)
    shape = [(b, names) for b, names, _ in eng.submitted]
    assert (4, ["im00", "im02", "im04", "im06"]) in shape and (4, ["im08", "im10", "im11", "im12"]) in shape
    assert (4, ["im03", "im07"]) in shape                                  # C: two of them = half a batch of four: one padded pass
    lines = [ln.split(" ran in ")[0] for ln in out.splitlines()]
    assert lines == [x for im in images for x in ("num rois: 300", "image %s" % im.name)]
    assert sum(len(v) for c in dets.values() for v in c.values()) == len(images)
    # an engine that cannot probe captures as before (every geometry), and a small window flushes the oldest geometry early
    eager.clear()
    eng2 = _FakeEngine(batch=4, in_flight=2)
    eng2.probe = False
    monkeypatch.setattr(voc_dets, "REORDER_WINDOW", 3)
    images, ratios, dets, out = _run_by_cls(monkeypatch, eng2, [A, B, C, D, A, B, C, D])
    assert not eager and sorted(n for _, names, _ in eng2.submitted for n in names) == sorted(im.name for im in images)
    assert eng2.submitted[0][1] == ["im00"]                                # four waiting > window of three: the oldest geometry goes alone
    lines = [ln.split(" ran in ")[0] for ln in out.splitlines()]
    assert lines == [x for im in images for x in ("num rois: 300", "image %s" % im.name)]


def test_get_dets_by_cls_single_image_engine_and_slow_fetches(monkeypatch):
    from faster_rcnn_amd import voc_dets
    eng = _FakeEngine(batch=1, in_flight=3, fetch_ms=voc_dets.DECODE_INLINE_MS + 3.0)
    images, ratios, dets, out = _run_by_cls(monkeypatch, eng, [(375, 500)] * 7)
    assert [(b, len(n)) for b, n, _ in eng.submitted] == [(1, 1)] * 7 and eng.max_open <= 3
    assert len(eng.fetch_threads) >= 2                                     # two slow fetches in a row: the next ones come from pool threads
    assert sum(len(v) for c in dets.values() for v in c.values()) == 7


def test_get_dets_by_cls_failure_leaves_no_ticket_open(monkeypatch):
    import pytest
    eng = _FakeEngine(batch=8, in_flight=2)
    real = eng.host_pixels

    def failing(image):
        if image.name == "im10":
            raise TypeError("bad pixels")
        return real(image)
    eng.host_pixels = failing
    synced = []
    orig_submit = eng.submit_batch

    def submit(*a, **k):
        t = orig_submit(*a, **k)
        t.slot.event.synchronize = lambda: synced.append(1)
        return t
    eng.submit_batch = submit
    with pytest.raises(TypeError):
        _run_by_cls(monkeypatch, eng, [(600, 1500)] * 12)
    assert len(eng.submitted) == 1 and synced == [1]                       # the whole batch of eight was in flight: waited for, slot released


def test_image_data_falls_back_to_the_host_resize_when_the_device_path_fails(monkeypatch):
    """``Image.data`` is a pure-host property; a device resize that cannot run (no library, no memory, a forked
    worker) must fall back to the bit-identical numpy restatement instead of raising, and a process that has not started the
    HIP runtime must not start it just to read pixels."""
    from faster_rcnn_amd import shapes
    rs = np.random.RandomState(5)
    img = rs.randint(0, 256, (37, 53, 3)).astype(np.uint8)
    want = shapes._resize(img, 80, 60)
    assert shapes._device_resize_ok() is False                      # CPU box / runtime not initialised: host path
    assert np.array_equal(shapes._resize_any(img, 80, 60), want)
    monkeypatch.setattr(shapes, "_device_resize_ok", lambda: True)  # pretend the runtime is up: the device path raises here
    assert np.array_equal(shapes._resize_any(img, 80, 60), want)
    monkeypatch.setattr(shapes, "DEVICE_RESIZE", False)
    assert np.array_equal(shapes.InMemoryImage(img, 80, 60).data, want)


def test_canvas_sides_are_even_with_room_for_the_offset_and_levels_follow_the_reference_chain():
    """entry.canvas_side: even, a multiple of the granule, at least the side plus the offset an odd side sits at (the offset stands for
    the extra zero row SAME padding puts in front of an odd side under conv1's window, resnet.py:408); nets.Extents.levels_of: the
    per-level true sizes a canvas pass masks with end in resnet.get_conv_rows_cols' numbers (resnet.py:78-93) for every size."""
    from faster_rcnn_amd import nets, resnet
    for n in list(range(7, 80)) + [375, 500, 599, 600, 601, 800, 898, 901, 904, 999, 1000, 1023, 1024, 1025]:
        c = entry.canvas_side(n)
        assert c >= n + (n & 1) and c % 2 == 0 and c % entry.CANVAS_GRANULE == 0 and c - n <= entry.CANVAS_GRANULE, (n, c)
        assert entry.canvas_side(c) == c                       # a canvas is its own class
    assert entry.canvas_side(600) == 608 and entry.canvas_side(607) == 608 and entry.canvas_side(800) == 800 and entry.canvas_side(901) == 928
    assert entry.canvas_side(601, granule=2) == 602 and entry.canvas_side(600, granule=2) == 600
    with pytest.raises(AssertionError):
        entry.canvas_side(600, granule=3)
    for h, w in [(600, 1000), (601, 901), (375, 500), (333, 499), (224, 224), (607, 927)]:
        lv = nets.Extents.levels_of(h, w)
        assert len(lv) == nets.Extents.LEVELS and list(lv[-1]) == list(resnet.get_conv_rows_cols(h, w))
        assert all(a[0] >= b[0] and a[1] >= b[1] for a, b in zip(lv, lv[1:]))
    # a canvas never has FEWER cells than the image at any level (the true extent fits inside), for the tightest canvas too
    for h, w in [(600, 901), (599, 800), (306, 451), (601, 1001)]:
        for g in (None, 2):
            hc, wc = entry.canvas_side(h, g), entry.canvas_side(w, g)
            assert all(c[0] >= t[0] and c[1] >= t[1] for c, t in zip(nets.Extents.levels_of(hc, wc), nets.Extents.levels_of(h, w)))


def test_canvas_plan_merges_what_is_not_worth_a_capture_and_is_stable():
    """entry.plan_canvas_classes: a histogram of sizes -> few classes.  Rare sizes join a class that holds them, the common formats
    keep canvases of their own, landscape and portrait never share one; classes that already have a captured pass are free, so the
    plan of a second call over the same list is the first call's."""
    counts = {(600, 800): 106, (600, 901): 36, (800, 600): 28, (600, 898): 11, (901, 600): 10, (600, 802): 8, (600, 904): 8, (600, 750): 5,
              (562, 1000): 4, (600, 645): 1, (600, 971): 1, (538, 1000): 1, (600, 619): 1}
    plan = entry.plan_canvas_classes(counts)
    assert set(plan) == set(counts)
    classes = set(plan.values())
    assert 2 <= len(classes) <= 6, classes
    for (h, w), (hc, wc) in plan.items():
        assert hc >= h + (h & 1) and wc >= w + (w & 1) and hc % 2 == 0 and wc % 2 == 0
        assert (hc > wc) == (h > w)                               # a portrait frame on a landscape canvas would double its pixels
    assert plan[(600, 800)] == (608, 800)                         # 106 images: their own tight canvas
    assert plan[(600, 901)] == plan[(600, 898)] == plan[(600, 904)]
    assert plan[(600, 619)] in classes - {(entry.canvas_side(600), entry.canvas_side(619))} or len(classes) == 1      # one image is not worth a capture
    again = entry.plan_canvas_classes(counts, existing=classes)
    assert again == plan
    # a long list amortises more captures: twenty times the images, at least as many classes
    big = entry.plan_canvas_classes({k: 20 * v for k, v in counts.items()})
    assert len(set(big.values())) >= len(classes)
    # one class at most when captures are dear, every geometry its own when they are free
    assert len(set(entry.plan_canvas_classes(counts, capture_images=1e9).values())) <= 2
    assert len(set(entry.plan_canvas_classes(counts, capture_images=0.0, max_new=99).values())) == len({(entry.canvas_side(h), entry.canvas_side(w)) for h, w in counts})
    assert len(set(entry.plan_canvas_classes(counts, capture_images=0.0, max_new=3).values())) <= 3


def test_canvas_pass_allowances_follow_the_lists_shares_and_the_byte_budget():
    """DetectionEntry.plan_canvases: a class may hold its share of the passes in flight (at least one; two when it fills more than one pass)
    -- and never more passes than the cache's byte budget holds (they would only evict each other)."""
    import collections

    class FakeCache:
        def __init__(self, budget):
            self.byte_budget, self._slots = budget, collections.OrderedDict()

        def keys(self):
            return list(self._slots)
    counts = {(600, 800): 106, (600, 901): 36, (800, 600): 28, (562, 1000): 4}
    roomy = types.SimpleNamespace(cache=FakeCache(72 << 30), in_flight=4, batch=4, _canvas_of={}, _canvas_slots={})
    plan = entry.DetectionEntry.plan_canvases(roomy, counts)
    assert plan == entry.plan_canvas_classes(counts) and roomy._canvas_of == plan
    main = ("canvas",) + plan[(600, 800)]
    assert roomy._canvas_slots[main] == 3 and all(1 <= v <= 4 for v in roomy._canvas_slots.values())     # 61 % of the list, four passes in flight
    tight = types.SimpleNamespace(cache=FakeCache(6 << 30), in_flight=4, batch=4, _canvas_of={}, _canvas_slots={})
    entry.DetectionEntry.plan_canvases(tight, counts)
    est = sum(v * entry.CANVAS_BYTES_PER_PIXEL * 4 * k[1] * k[2] for k, v in tight._canvas_slots.items())
    assert est <= 0.9 * (6 << 30) or all(v == 1 for v in tight._canvas_slots.values())
    assert tight._canvas_slots[main] == max(tight._canvas_slots.values())          # the common class keeps the most


def test_a_file_backed_frame_is_decoded_once_on_its_way_to_the_device():
    """Round 6: `hasattr(image, "raw")` RAN Image.raw -- a JPEG decode plus a channel reversal, ~1.7 ms -- only to learn that the attribute
    exists (from JPEG files 406-429 img/s instead of 509-524).  shapes.declares looks at the class; host_pixels / prefetchable / the training
    feed's test decode the file ONCE, through raw_rgb, and probe_geometry asks PIL for the header once however often it is called."""
    import os
    from faster_rcnn_amd import shapes
    from faster_rcnn_amd.data.voc_data_helpers import extract_img_data

    class Loud:
        @property
        def raw(self):
            raise AssertionError("the property ran")
        height = 3
    assert shapes.declares(Loud(), "raw") and shapes.declares(Loud(), "height") and not shapes.declares(Loud(), "width")
    inst = types.SimpleNamespace(raw=1)
    assert shapes.declares(inst, "raw") and not shapes.declares(inst, "data")
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "VOC_test")
    img = extract_img_data(root, "000005")
    from PIL import Image as PilImage
    opened = []
    real_open = PilImage.open
    PilImage.open = lambda *a, **k: opened.append(1) or real_open(*a, **k)
    try:
        eng = types.SimpleNamespace(device_preprocess=True, canvas=False, canvas_capable=False)
        assert entry.DetectionEntry.prefetchable(eng, img) and not opened
        arr, H, W, src, flip = entry.DetectionEntry.host_pixels(eng, img)
        assert len(opened) == 1 and arr.shape == (375, 500, 3) and src == (375, 500) and flip == 2 and (H, W) == (375, 500)
        k1 = entry.DetectionEntry.probe_geometry(eng, img)
        k2 = entry.DetectionEntry.probe_geometry(eng, img)
        assert k1 == k2 == (375, 500, 375, 500, 2) and len(opened) == 1            # (the decode left the size behind: no header read at all)
        fresh = extract_img_data(root, "000005")
        entry.DetectionEntry.probe_geometry(eng, fresh)
        entry.DetectionEntry.probe_geometry(eng, fresh)
        assert len(opened) == 2
    finally:
        PilImage.open = real_open
    assert np.array_equal(arr[:, :, ::-1], img.raw)


def test_feed_decodes_a_named_frame_ahead_on_one_background_thread():
    """feed.decode_ahead (train_util names the image two iterations ahead): a file-backed frame's pixels come from the background thread's
    future, once; in-memory frames and frames nobody named decode where they are asked for; the training managers delegate (the detector's
    only on request: the thread does not pay there)."""
    import os
    from faster_rcnn_amd import feed, shapes
    from faster_rcnn_amd.data.voc_data_helpers import extract_img_data
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "VOC_test")
    img = extract_img_data(root, "000005")
    feed._DECODED.clear()
    feed.decode_ahead(img)
    feed.decode_ahead(img)                                          # (asked twice: one job)
    assert list(feed._DECODED) == [id(img)]
    got = feed._raw_rgb(img)
    assert not feed._DECODED and np.array_equal(got, img.raw_rgb) and got.shape == (375, 500, 3)
    assert np.array_equal(feed._raw_rgb(img), got)                  # nobody named it: decoded inline
    mem = shapes.Image(shapes.Metadata("m", 8, 6, [], "none"), np.zeros((6, 8, 3), np.uint8))
    feed.decode_ahead(mem)
    assert not feed._DECODED
    from faster_rcnn_amd import det_util, resnet, rpn_util, util
    anchors = util.get_anchors([128, 256, 512])
    rpn_util.RpnTrainingManager(resnet.get_conv_rows_cols, 16, resnet.preprocess, anchors).decode_ahead(img)
    assert list(feed._DECODED) == [id(img)]
    feed._DECODED.clear()
    det_util.DetTrainingManager(types.SimpleNamespace(output=[0, 1, 2]), {"bg": 0}, resnet.preprocess, anchor_dims=anchors).decode_ahead(img)
    assert not feed._DECODED


def test_full_collections_before_captures_are_throttled(monkeypatch):
    """pipeline.collect_before_capture: at most one full gc.collect() per interval (a collection costs more than a capture)."""
    from faster_rcnn_amd import pipeline
    calls = []
    monkeypatch.setattr(pipeline._gc, "collect", lambda *a: calls.append(1) or 0)
    monkeypatch.setattr(pipeline, "_LAST_COLLECT", [0.0])
    for _ in range(5):
        pipeline.collect_before_capture(min_interval_s=60.0)
    assert len(calls) == 1
    pipeline.collect_before_capture(min_interval_s=0.0)
    assert len(calls) == 2
