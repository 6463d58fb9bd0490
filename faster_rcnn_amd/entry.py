"""The reference's INFERENCE ENTRY POINTS on the captured device pipeline.

``voc_dets.get_dets`` / ``get_dets_by_cls`` (voc_dets.py:20-111) reach the device through
``DetTrainingManager.get_det_inputs`` (det_util.py:136-158) and ``detector.predict`` (voc_dets.py:49), crossing to
numpy twice per image (det_util.py:41, 49-55).  When the manager's ``rpn_model`` and the ``detector`` are this package's
models, the same call runs here as ONE captured pass per image:

    image.data (uint8 BGR)  ->  one H2D copy [pixels | resize_ratio, det_threshold]
                            ->  hipGraph: resnet.preprocess on the device, backbone, RPN heads, decode, top-8000,
                                NMS to 300, the reference's padded RoI list, RoI crop-resize, detector head, post-process
                            ->  one packed D2H copy [n_dets, n_rois | boxes | classes | scores]

* A captured pass is specific to an image SIZE (VOC has dozens after ``util.resize_imgs``): ``GraphCache`` keeps them per
  (height, width), least-recently-used first out under a byte budget; a size is captured at its first sighting.
  ``resize_ratio`` and ``det_threshold`` are per-image VALUES, read from device memory by the post-process
  (frcnn_detections_dyn), so they are not part of the key.
* ``get_dets_by_cls`` keeps several images in flight (one HIP stream each; a size seen several times in a row gets as
  many captured instances as are in flight) and hands results back in submission order, so the returned dict is the
  reference's, entry for entry.
* The frame is uploaded as DECODED (``image.raw``: the file's own size, 0.56 MB for a 500x375 VOC frame) and resized on the
  device inside the captured pass (frcnn_resize_cubic_u8: OpenCV's 8-bit INTER_CUBIC in integers, bit-identical to the host
  restatement in shapes._resize, which takes ~90 ms per frame in numpy); such a pass is keyed by (height, width, source
  height, source width, flip).  ``get_dets_by_cls`` decodes the JPEGs of the next images on a small thread pool meanwhile.
* A foreign ``preprocess_func`` (anything but resnet.preprocess / vgg.preprocess) is called on the host, as the reference
  does (det_util.py:36), and its float image is uploaded instead of the bytes; a foreign ``detector`` (any other object
  with Keras' ``predict``) keeps voc_dets' eager path.
* Weights that change (load_weights, a training step, set_weights) bump ``models.weights_epoch()``; every captured pass
  holds pointers to packed filters, so the cache is dropped and re-captured on the next call.
"""
import collections
import os

import numpy as np
import torch

from . import models, ops
from ._lib import FrcnnError
from .pipeline import InferencePipeline
from .shapes import declares as _declares

MEAN_BGR = (103.939, 116.779, 123.68)           # resnet.preprocess / vgg.preprocess (resnet.py:64-75, vgg.py:52-57)
PRE_NMS_TOP_N, MAX_PROPOSALS = 8000, 300        # det_util.py:151-156


def _default_budget():
    env = os.environ.get("FRCNN_GRAPH_CACHE_BYTES")
    if env:
        return int(float(env))
    total = torch.cuda.get_device_properties(torch.cuda.current_device()).total_memory
    return total // 4                           # 72 GB of an MI355X's 288: ~150 captured 600x1000 fp32 passes


def default_in_flight(dtype="f32"):
    """Captured passes in flight for get_dets_by_cls: four (FRCNN_ENTRY_IN_FLIGHT overrides).  What matters more is that the PROCESS has
    more hardware queues than ROCm's default four (GPU_MAX_HW_QUEUES, read when the runtime starts; voc_dets.main sets 8): on four, the
    passes' staging copies, replays and read-backs queue behind one another -- 497 img/s through get_dets_by_cls against 542-547 on 8 or
    16 queues, where 4 / 5 / 6 / 8 / 12 passes in flight measure 545 / 526 / 542 / 535 / 497 (fp32, four images per pass, 256 frames;
    scripts/dev/r6_entry_child_sweep.sh).  (Until round 6 this returned up to 12 on 12 queues: right for one-image passes.)"""
    env = int(os.environ.get("FRCNN_ENTRY_IN_FLIGHT", "0"))
    return env if env > 0 else 4


def default_batch(dtype="f32"):
    """Images per captured pass where neighbouring images share a geometry (FRCNN_ENTRY_BATCH / FRCNN_ENTRY_BATCH_F32)."""
    if dtype == "bf16":
        return max(1, int(os.environ.get("FRCNN_ENTRY_BATCH", "8")))
    return max(1, int(os.environ.get("FRCNN_ENTRY_BATCH_F32", "4")))


# The reference pads the last batch of 64 RoIs with copies of its first RoI and scores the copies too (voc_dets.py:42-51).  A copy
# has the box, class and score of its original, so the per-class NMS that follows (threshold 0.5) keeps exactly one of them where it
# would have kept the original: the returned list is the same with or without the copies.  The captured passes therefore score the
# kept proposals only (300 rows instead of 320: 6 % less detector-head work); FRCNN_ENTRY_PAD=1 scores the padded list as the eager
# path does.  tests/test_entry_gpu.py compares the two paths detection by detection.
PAD_TO_BATCH = os.environ.get("FRCNN_ENTRY_PAD", "0") != "0"
# eager passes in front of a capture: ONE sizes the split-K workspace, lowers what is lowered lazily and leaves the magnitude-record
# arena's high-water mark (round 6: two until then; a capture is ~2/3 warm-up, and a list of mixed sizes captures per geometry)
WARMUP_PASSES = max(1, int(os.environ.get("FRCNN_ENTRY_WARMUP", "1")))
# Padded canvases (round 6): a list of MANY image sizes is served by passes captured per canvas CLASS with the images' true sizes as
# device values (pipeline: ``extents``), instead of one capture per geometry.  A canvas has EVEN sides, multiples of CANVAS_GRANULE; an
# image sits at offset (H & 1, W & 1) (csrc/boxes.hip: the offset stands for the extra zero row / column SAME padding puts in front of
# an odd side).  voc_dets.get_dets_by_cls switches canvases on for a call whose list holds more than CANVAS_MIN_GEOMETRIES sizes and
# PLANS the classes from the list's size histogram (DetectionEntry.plan_canvases): few classes, each worth its captures.
# file-backed frames go up in the decoder's channel order and are swapped to BGR by the device resize (0: reverse on the host as before)
RGB_UPLOAD = os.environ.get("FRCNN_ENTRY_RGB_UPLOAD", "1") != "0"
CANVAS_GRANULE = int(os.environ.get("FRCNN_ENTRY_CANVAS_GRANULE", "32"))
CANVAS_MIN_GEOMETRIES = int(os.environ.get("FRCNN_ENTRY_CANVAS_MIN", "4"))
# captured passes kept per canvas class: with several classes interleaving in a list, two of one class in flight at once is the common
# worst case; a third would be a capture (~20-35 ms) to save a wait of a few ms -- get_dets_by_cls waits instead (should_wait)
CANVAS_SLOTS_PER_CLASS = int(os.environ.get("FRCNN_ENTRY_CANVAS_SLOTS", "2"))


# the plan's cost model, in image-times: a captured pass costs about this many images' worth of host time per class (15-25 ms per
# capture, a class has a few passes in flight); padded pixels cost the trunk's and the RPN's share of an image's time (the head's
# 300 RoIs do not grow with the canvas)
CANVAS_CAPTURE_IMAGES = float(os.environ.get("FRCNN_ENTRY_CANVAS_CAPTURE_IMAGES", "12"))
CANVAS_PIXEL_SHARE = float(os.environ.get("FRCNN_ENTRY_CANVAS_PIXEL_SHARE", "0.4"))
CANVAS_MAX_CLASSES = int(os.environ.get("FRCNN_ENTRY_CANVAS_MAX_CLASSES", "8"))
CANVAS_EXTRA_SLOT = int(os.environ.get("FRCNN_ENTRY_WINDOW_EXTRA", "0"))      # (voc_dets.WINDOW_EXTRA: passes submitted beyond the streams)
# passes a class may hold = factor x passes in flight x its share of the list, at least CANVAS_SLOT_MIN once it fills more than one pass.
# bench.py's mixed list (three classes, 52 / 31 / 17 %), first call / again, img/s: factor 1.5 (4 + 2 + 2 = 8 captures) 446 / 558; 1.0
# (3 + 2 + 2 = 7) 468-478 / 557-559; 1.0 with a minimum of one (3 + 2 + 1 = 6) 427-457 / 552-553; 0.75 (4 captures) 465-480 / 533
CANVAS_SLOT_FACTOR = float(os.environ.get("FRCNN_ENTRY_CANVAS_SLOT_FACTOR", "1.0"))
CANVAS_SLOT_MIN = int(os.environ.get("FRCNN_ENTRY_CANVAS_SLOT_MIN", "2"))
CANVAS_BYTES_PER_PIXEL = 600                             # a captured fp32 pass's memory per canvas pixel, before one of its class has been measured


def canvas_side(n, granule=None):
    """The smallest canvas side for a true side ``n``: even, a multiple of the granule, with room for the offset of an odd side."""
    g = CANVAS_GRANULE if granule is None else granule
    assert g > 0 and g % 2 == 0, "canvas granule: a positive even number"
    n = int(n)
    return -(-(n + (n & 1)) // g) * g


def plan_canvas_classes(counts, existing=(), capture_images=None, pixel_share=None, max_new=None, granule=None):
    """{(H, W): canvas (Hc, Wc)} for a histogram ``counts`` = {(H, W): number of images}: start with every geometry on its own smallest
    canvas (or on an ``existing`` class -- a pass already captured -- when the padding costs less than a capture), then merge the pair
    of classes whose union canvas lowers  sum over NEW classes of capture_images + sum over images of pixel_share * (canvas area /
    image area - 1)  the most, until no merge lowers it and at most ``max_new`` new classes are left.  Deterministic: the same
    histogram and the same existing classes give the same plan (a second call over the same list captures nothing)."""
    cap = CANVAS_CAPTURE_IMAGES if capture_images is None else capture_images
    share = CANVAS_PIXEL_SHARE if pixel_share is None else pixel_share
    max_new = CANVAS_MAX_CLASSES if max_new is None else max_new
    ex = set(tuple(e) for e in existing)
    area = lambda c: c[0] * c[1]

    def padding(geos, c):
        return sum(n * share * (area(c) / float(h * w) - 1.0) for (h, w), n in geos)

    def cost(c, geos):
        return (0.0 if c in ex else cap) + padding(geos, c)
    classes = {}
    for (h, w), n in sorted(counts.items()):
        own = (canvas_side(h, granule), canvas_side(w, granule))
        best, best_cost = own, cost(own, [((h, w), n)])
        for e in sorted(ex):
            if e[0] >= own[0] and e[1] >= own[1]:
                c = padding([((h, w), n)], e)
                if c < best_cost:
                    best, best_cost = e, c
        classes.setdefault(best, []).append(((h, w), n))
    while len(classes) > 1:
        keys, best = sorted(classes), None
        for i, a in enumerate(keys):
            for b in keys[i + 1:]:
                u = (max(a[0], b[0]), max(a[1], b[1]))
                if a in ex and b in ex and u not in (a, b):
                    continue                                  # (two captured classes: nothing to save by leaving both)
                third = classes[u] if (u in classes and u not in (a, b)) else []
                delta = cost(u, classes[a] + classes[b] + third) - cost(a, classes[a]) - cost(b, classes[b]) - (cost(u, third) if third else 0.0)
                if best is None or delta < best[0]:
                    best = (delta, a, b, u)
        if best is None or (best[0] >= 0.0 and sum(1 for k in keys if k not in ex) <= max_new):
            break
        _, a, b, u = best
        merged = classes.pop(a) + classes.pop(b) + classes.pop(u, [])
        classes[u] = merged
    return {g: c for c, geos in classes.items() for g, _ in geos}


class _PinnedArena:
    """Pinned host memory for the captured passes' staging, handed out in 4 KB-aligned pieces from a few large blocks: a capture
    took two hipHostMalloc calls (frame staging, detection read-back), several milliseconds each, and a list of mixed image sizes
    captures per geometry.  Pieces are not returned one by one: a block goes back when every slot cut from it has been closed."""
    BLOCK = 64 << 20

    def __init__(self):
        self.blocks = []                                 # [tensor, used bytes, live pieces]

    def take(self, nbytes, zero=False):
        nbytes = max(int(nbytes), 16)
        need = (nbytes + 4095) // 4096 * 4096
        for blk in self.blocks:
            if blk[0].numel() - blk[1] >= need:
                break
        else:
            blk = [torch.empty(max(self.BLOCK, need), dtype=torch.uint8).pin_memory(), 0, 0]
            self.blocks.append(blk)
        piece = blk[0][blk[1]:blk[1] + nbytes]
        blk[1] += need
        blk[2] += 1
        if zero:
            piece.zero_()
        piece._arena_block = blk
        return piece

    @staticmethod
    def give_back(piece):
        blk = getattr(piece, "_arena_block", None)
        if blk is None:
            return
        blk[2] -= 1
        if blk[2] == 0:
            blk[1] = 0                                   # every piece of the block is gone: the block starts over


_CAPTURE_STREAM = None


def _capture_stream():
    global _CAPTURE_STREAM
    if _CAPTURE_STREAM is None:
        _CAPTURE_STREAM = torch.cuda.Stream()
    return _CAPTURE_STREAM


class _Slot:
    """One captured pass for one image size, with its staging on both sides of PCIe."""
    __slots__ = ("key", "pipe", "graph", "out", "io_dev", "io_pin", "pix_host", "dyn_host", "out_pin", "event", "busy", "nbytes",
                 "x_f32", "ws", "seq", "tabs", "u8_resized", "batch", "pix_hosts", "out_packed", "amax", "_out_raw", "ready", "extents", "seg", "canvas", "_ext_raw")


def _close_slot(s):
    """Release a captured pass NOW, from the thread that owns it: the hipGraph is destroyed here (torch.cuda.CUDAGraph.reset) and the
    references to its tensors are dropped, so that nothing is left for a garbage-collector finalizer to do later from whatever thread
    happens to collect (a finalizer issuing HIP calls inside another capture crashed a launch).  Idle slots only."""
    assert not s.busy, "closing a captured pass with an image in flight"
    if getattr(s, "ready", None) is not None:
        s.ready.synchronize()                            # (never replayed: its warm-up pass may still be writing the slot's buffers)
    if getattr(s, "graph", None) is not None:
        s.event.synchronize()                            # its last replay (and the copies behind it) are done
        s.graph.reset()
    if getattr(s, "pipe", None) is not None and hasattr(s.pipe, "close"):
        s.pipe.close()
    for name in ("io_pin", "_out_raw", "_ext_raw"):
        piece = getattr(s, name, None)
        if piece is not None:
            _PinnedArena.give_back(piece)
    for name in ("graph", "out", "out_packed", "io_dev", "x_f32", "u8_resized", "tabs", "ws", "amax", "pipe", "io_pin", "_out_raw", "out_pin", "pix_hosts", "pix_host", "dyn_host", "extents", "_ext_raw"):
        setattr(s, name, None)


class GraphCache:
    """Captured passes by image size, least-recently-used first out under a byte budget.  ``acquire(key, make)`` hands out
    an idle slot of that size or captures a new one; busy slots (an image in flight) are never evicted."""

    def __init__(self, byte_budget):
        self.byte_budget = int(byte_budget)
        self._slots = collections.OrderedDict()          # key -> [slots]; order = recency
        self.nbytes = 0
        self.captures = self.evictions = self.hits = 0

    def __len__(self):
        return sum(len(v) for v in self._slots.values())

    def keys(self):
        return list(self._slots)

    def acquire(self, key, make):
        slots = self._slots.get(key)
        if slots is not None:
            self._slots.move_to_end(key)
            for s in slots:
                if not s.busy:
                    self.hits += 1
                    return s
        slot = make()
        self.captures += 1
        self._slots.setdefault(key, []).append(slot)
        self._slots.move_to_end(key)
        self.nbytes += slot.nbytes
        self._evict(keep=slot)
        return slot

    def _evict(self, keep):
        if self.nbytes <= self.byte_budget:
            return
        dropped = False
        for key in list(self._slots):                    # oldest size first
            slots = self._slots[key]
            for s in list(slots):
                if self.nbytes <= self.byte_budget:
                    break
                if s.busy or s is keep:
                    continue
                slots.remove(s)
                self.nbytes -= s.nbytes
                self.evictions += 1
                _close_slot(s)
                dropped = True
            if not slots:
                del self._slots[key]
            if self.nbytes <= self.byte_budget:
                break
        if dropped:
            s = slots = None                             # (the loop variables still hold the last slot examined: its pool must be free to go)
            torch.cuda.empty_cache()                     # a dropped graph's private pool goes back to the device

    def clear(self):
        assert not any(s.busy for v in self._slots.values() for s in v), "graph cache cleared with images in flight"
        for v in self._slots.values():
            for s in v:
                _close_slot(s)
        self._slots.clear()
        self.nbytes = 0
        torch.cuda.empty_cache()


class Ticket:
    __slots__ = ("slot", "image")

    def __init__(self, slot, image):
        self.slot, self.image = slot, image


class DetectionEntry:
    """voc_dets.get_dets for one (manager, detector) pair: ``submit(image, resize_ratio, det_threshold)`` enqueues an image,
    ``collect(ticket)`` returns ``(num_rois, dets)`` with ``dets`` the reference's list of
    ``{'bbox': int array [x1,y1,x2,y2], 'cls_name', 'prob'}`` in its order (voc_dets.py:73-86)."""

    def __init__(self, manager, detector, num_rois=64, stride=16, in_flight=1, byte_budget=None, f32_engine=None):
        self.manager, self.detector = manager, detector
        # the matrix path of the fp32 convolutions inside the captured passes (ops.f32_engine): by default the large launches run
        # on the fp16 matrix cores by a two-way operand split with a scaled low part -- fp32-grade results (csrc/conv_h3.hip:
        # error against fp64 under the native kernel's), three matrix instructions per block of products; FRCNN_F32_ENGINE=bf16x6
        # selects the exact three-way bf16 split (six), FRCNN_F32_ENGINE=native keeps v_mfma_f32_32x32x2_f32 everywhere
        self.f32_engine = f32_engine or os.environ.get("FRCNN_F32_ENGINE", "f16x3")
        self.num_rois, self.stride, self.in_flight = int(num_rois), stride, max(1, int(in_flight))
        # images per captured pass.  The bf16 detector makes ONE head pass over the RoIs of eight images and runs its trunk at batch
        # eight (pipeline.BatchedInferencePipeline: what configs[3]'s headline times); get_dets_by_cls groups neighbouring images of
        # one size into such passes.  The f32 ResNet models take four per pass since round 5 (the TRUNK is what gains: its launches are
        # latency-bound at one image -- 0.59 -> 0.445 ms per image at four; the head is unchanged, as DESIGN 11 found in round 2).
        head = getattr(detector, "head", None)
        self.batch = 1
        if hasattr(head, "forward_batched") and getattr(head, "hoist", False):
            self.batch = default_batch(getattr(head, "dtype", "f32"))
        from . import resnet, vgg
        self.device_preprocess = manager.preprocess_func in (resnet.preprocess, vgg.preprocess)
        self.rev_class_mapping = dict((v, k) for k, v in manager.class_mapping.items())
        self.cache = GraphCache(_default_budget() if byte_budget is None else byte_budget)
        self._streams = [torch.cuda.Stream() for _ in range(self.in_flight)]
        self._seq = 0
        self._epoch = models.weights_epoch()
        self.capture_seconds = 0.0
        # canvas passes need the ResNet trunk's extent masks (nets.ResNetBase) and the device-side preprocess
        net = getattr(getattr(manager.rpn_model, "base", None), "net", None)
        # (FRCNN_ENTRY_CANVAS=0 switches them off: every geometry then gets passes of its own, as in round 5.)  Measured on bench.py's
        # mixed_sizes legs (256 frames, 36 geometries): see DESIGN 5 / 7.
        self.canvas_capable = self.device_preprocess and hasattr(net, "block_level") and os.environ.get("FRCNN_ENTRY_CANVAS", "1") != "0"
        self.canvas = False                              # set per call by voc_dets.get_dets_by_cls
        self._canvas_of = {}                             # (H, W) -> canvas class: the plan (plan_canvases) + sizes seen outside it
        self._canvas_slots = {}                          # canvas class -> captured passes it may hold (its share of the list x in_flight)
        self._taps_dev = {}
        self._pinned = _PinnedArena()                    # (per engine: the blocks go when the engine goes)
        self.capture_breakdown = {}

    # ------------------------------------------------------------------ eligibility
    @staticmethod
    def usable(manager, detector, num_rois):
        rpn = getattr(manager, "rpn_model", None)
        if not (isinstance(rpn, models.RpnModel) and isinstance(detector, models.DetModel)):
            return False                                # a foreign Keras-style model: voc_dets keeps the eager path
        if not manager.conv_only or detector.base is not None:
            return False                                # the detector must take the RPN's conv map (voc_dets.py:177-183)
        n_rows = -(-MAX_PROPOSALS // int(num_rois)) * int(num_rois)
        return 0 < n_rows <= 512                        # frcnn_detections: one workgroup, <= 512 scored rows

    # ------------------------------------------------------------------ capture
    def _capture(self, H, W, src=None, flip=False, B=1):
        """``src`` = (source height, source width): the pass starts from the decoded frame at the file's size and resizes
        (and flips) it on the device; None: the uploaded pixels are already (H, W).  ``B`` > 1: one pass over B frames of that
        geometry (pipeline.BatchedInferencePipeline), each with its own [resize_ratio, det_threshold] pair."""
        import gc
        import time
        t0 = time.perf_counter()
        # No cyclic-garbage collection inside the capture: a collection that starts while the stream is capturing may run the
        # finalizers of unrelated dead objects -- another engine's captured passes, their private memory pools -- whose HIP calls
        # are not legal in a capturing thread (seen as a crash inside a launch when a test's models died just before).
        gc_was_on = gc.isenabled()
        from .pipeline import collect_before_capture
        collect_before_capture()                                    # (at most one full collection per second: it costs more than the capture)
        gc.disable()
        try:
            return self._capture_locked(H, W, src, flip, B, t0)
        finally:
            if gc_was_on:
                gc.enable()

    def _device_taps(self, dst, src):
        t = self._taps_dev.get((dst, src))
        if t is None:
            if len(self._taps_dev) > 256:
                self._taps_dev.clear()
            t = self._taps_dev[(dst, src)] = torch.from_numpy(ops.resize_cubic_taps(dst, src)).cuda()
        return t

    def _capture_canvas_locked(self, Hc, Wc, B, t0):
        """A pass over B canvases of (Hc, Wc): the frames' resize + preprocess stay OUTSIDE the graph (their sizes are per image: a few
        eager launches in front of every replay), the graph holds trunk .. post-process with the images' true extents read from
        ``s.extents`` (device words)."""
        import time
        from . import nets
        m = self.manager
        kw = dict(stride=self.stride, pre_nms_top_n=PRE_NMS_TOP_N, max_proposals=MAX_PROPOSALS, roi_batch=self.num_rois, pad_to_batch=PAD_TO_BATCH,
                  bg_idx=m.class_mapping["bg"])
        if B > 1:
            from .pipeline import BatchedInferencePipeline
            pipe = BatchedInferencePipeline(m.rpn_model, self.detector, m.anchor_dims, B, **kw)
        else:
            pipe = InferencePipeline(m.rpn_model, self.detector, m.anchor_dims, **kw)
        reserved0 = torch.cuda.memory_reserved()
        stamps = [("pipeline", time.perf_counter())]
        seg = (Hc * Wc * 3 + 15) // 16 * 16                         # a frame's staging segment: any source of at most the canvas's pixel count
        off = B * seg
        s = _Slot()
        s.key, s.pipe, s.busy, s.seq, s.batch, s.canvas, s.seg = ("canvas", Hc, Wc), pipe, False, 0, B, True, seg
        s.io_dev = torch.zeros(off + 16 * B, dtype=torch.uint8, device="cuda")
        stamps.append(("buffers: device staging", time.perf_counter()))
        s.io_pin = self._pinned.take(off + 16 * B, zero=True)
        stamps.append(("buffers: pinned staging", time.perf_counter()))
        host = s.io_pin.numpy()
        s.pix_hosts = [host[i * seg:(i + 1) * seg] for i in range(B)]
        s.pix_host = s.pix_hosts[0]
        s.dyn_host = host[off:off + 16 * B].view(np.float64).reshape(B, 2)
        s.dyn_host[:] = (1.0, 0.0)
        dyn_dev = s.io_dev[off:off + 16 * B].view(torch.float64).view(B, 2)
        s.tabs = None
        s.u8_resized = torch.empty((B, seg), dtype=torch.uint8, device="cuda")
        s.x_f32 = torch.zeros((B, Hc, Wc, 3), dtype=torch.float32, device="cuda")
        stamps.append(("buffers: canvases", time.perf_counter()))
        s._ext_raw = self._pinned.take(nets.Extents.LEVELS * B * 8)
        s.extents = nets.Extents(B, pinned=s._ext_raw)
        for i in range(B):
            s.extents.set(i, Hc, Wc)
        s.extents.upload()
        run = lambda: pipe.forward_dev(s.x_f32, dyn=dyn_dev if B > 1 else dyn_dev[0], extents=s.extents)
        shared = self.in_flight > 1
        dtype = getattr(getattr(self.detector, "head", None), "dtype", "f32")
        s.ws = ops.NO_SPLIT_K if ((shared and (dtype == "bf16" or os.environ.get("FRCNN_ENTRY_NO_SPLITK"))) or B > 1) else ops.ConvWorkspace()
        stamps.append(("buffers: extents", time.perf_counter()))
        s.io_dev.copy_(s.io_pin)
        stamps.append(("buffers: first upload", time.perf_counter()))
        s.amax = ops.AmaxArena() if self.f32_engine == "f16x3" else None
        side = _capture_stream()
        side.wait_stream(torch.cuda.current_stream())
        stamps.append(("buffers", time.perf_counter()))
        with torch.cuda.stream(side), ops.conv_workspace(s.ws), ops.tile_policy(shared), ops.f32_engine(self.f32_engine), ops.amax_arena(s.amax):
            for _ in range(WARMUP_PASSES):
                run()
            stamps.append(("warm-up enqueue", time.perf_counter()))
            s.ready = torch.cuda.Event()
            s.ready.record(side)
            s.graph = torch.cuda.CUDAGraph()
            s.graph.capture_begin(pool=torch.cuda.graph_pool_handle(), capture_error_mode="thread_local")
            try:
                s.out = run()
                stamps.append(("capture enqueue", time.perf_counter()))
            finally:
                s.graph.capture_end()
        stamps.append(("instantiate", time.perf_counter()))
        packed = s.out["det_packed"]
        s.out_packed = packed if isinstance(packed, (list, tuple)) else [packed]
        s._out_raw = self._pinned.take(4 * B * s.out_packed[0].numel())
        s.out_pin = s._out_raw.view(torch.int32).view((B,) + tuple(s.out_packed[0].shape))
        s.event = torch.cuda.Event()
        s.nbytes = max(int(torch.cuda.memory_reserved() - reserved0), int(s.io_dev.numel() + s.x_f32.numel() * 4))
        stamps.append(("read-back buffers", time.perf_counter()))
        self.capture_seconds += time.perf_counter() - t0
        self._book_capture(t0, stamps)
        return s

    def _capture_canvas(self, Hc, Wc, B):
        import gc
        import time
        t0 = time.perf_counter()
        gc_was_on = gc.isenabled()
        from .pipeline import collect_before_capture
        collect_before_capture()
        gc.disable()
        try:
            return self._capture_canvas_locked(Hc, Wc, B, t0)
        finally:
            if gc_was_on:
                gc.enable()

    def _submit_canvas(self, images, resize_ratios, det_threshold, pixels, B):
        """``submit_batch`` in canvas mode: per image the frame goes up at ITS size, is resized (and flipped) to ITS (H, W) and
        preprocessed into the corner of canvas i by eager launches on the pass's stream; the true sizes go up as extents."""
        _, H0, W0, _, _ = pixels[0]
        Hc, Wc = self.canvas_class(H0, W0)
        s = self.cache.acquire(("canvas", Hc, Wc) + ((B,) if B > 1 else ()), lambda: self._capture_canvas(Hc, Wc, B))
        metas = []
        for i in range(B):
            j = i if i < len(images) else 0
            arr, H, W, src, flip = pixels[j]
            arr = np.ascontiguousarray(arr)
            assert arr.dtype == np.uint8 and arr.nbytes <= s.seg and H + (H & 1) <= Hc and W + (W & 1) <= Wc, "one pass, one canvas class"
            s.pix_hosts[i][:arr.nbytes] = arr.reshape(-1)
            s.dyn_host[i, 0], s.dyn_host[i, 1] = float(resize_ratios[j]), float(det_threshold)
            s.extents.set(i, H, W)
            metas.append((arr.shape[0], arr.shape[1], H, W, src, flip))
        st = self._streams[self._seq % self.in_flight]
        self._seq += 1
        with torch.cuda.stream(st):
            if s.ready is not None:
                st.wait_event(s.ready)
                s.ready = None
            s.io_dev.copy_(s.io_pin, non_blocking=True)
            s.extents.upload()
            for i, (in_h, in_w, H, W, src, flip) in enumerate(metas):
                frame = s.io_dev[i * s.seg:i * s.seg + in_h * in_w * 3].view(in_h, in_w, 3)
                if src is not None:                                 # shapes.Image.data: INTER_CUBIC resize (+ flip) of the decoded frame
                    frame = ops.resize_cubic_u8(frame, H, W, flip=flip, tabs=(self._device_taps(W, in_w), self._device_taps(H, in_h)),
                                                out=s.u8_resized[i][:H * W * 3].view(H, W, 3))
                ops.preprocess_u8_canvas(frame, MEAN_BGR, s.x_f32[i])
            s.graph.replay()
            for i in range(len(images)):
                s.out_pin[i].copy_(s.out_packed[i], non_blocking=True)
            s.event.record(st)
        s.busy = True
        return Ticket(s, list(images))

    def _capture_locked(self, H, W, src, flip, B, t0):
        import time
        m = self.manager
        kw = dict(stride=self.stride, pre_nms_top_n=PRE_NMS_TOP_N, max_proposals=MAX_PROPOSALS, roi_batch=self.num_rois, pad_to_batch=PAD_TO_BATCH,
                  bg_idx=m.class_mapping["bg"])
        if B > 1:
            from .pipeline import BatchedInferencePipeline
            pipe = BatchedInferencePipeline(m.rpn_model, self.detector, m.anchor_dims, B, **kw)
        else:
            pipe = InferencePipeline(m.rpn_model, self.detector, m.anchor_dims, **kw)
        reserved0 = torch.cuda.memory_reserved()
        stamps = [("pipeline", time.perf_counter())]
        in_h, in_w = src if src is not None else (H, W)
        npix = in_h * in_w * 3
        pix_bytes = npix if self.device_preprocess else 4 * npix
        seg = (pix_bytes + 15) // 16 * 16                           # one frame's segment of the staging buffer
        off = B * seg                                               # ... then B x [resize_ratio, det_threshold]
        s = _Slot()
        s.key, s.pipe, s.busy, s.seq, s.batch = (H, W), pipe, False, 0, B
        s.io_dev = torch.zeros(off + 16 * B, dtype=torch.uint8, device="cuda")
        s.io_pin = self._pinned.take(off + 16 * B, zero=True)
        host = s.io_pin.numpy()
        if self.device_preprocess:
            s.pix_hosts = [host[i * seg:i * seg + npix].reshape(in_h, in_w, 3) for i in range(B)]
        else:
            s.pix_hosts = [host[i * seg:i * seg + pix_bytes].view(np.float32).reshape(H, W, 3) for i in range(B)]
        s.pix_host = s.pix_hosts[0]
        s.dyn_host = host[off:off + 16 * B].view(np.float64).reshape(B, 2)
        s.dyn_host[:] = (1.0, 0.0)
        dyn_dev = s.io_dev[off:off + 16 * B].view(torch.float64).view(B, 2)
        s.tabs = s.u8_resized = None
        if self.device_preprocess:
            u8 = [s.io_dev[i * seg:i * seg + npix].view(in_h, in_w, 3) for i in range(B)]
            s.x_f32 = torch.empty((B, H, W, 3), dtype=torch.float32, device="cuda")
            if src is not None:
                assert self.device_preprocess
                s.tabs = (torch.from_numpy(ops.resize_cubic_taps(W, in_w)).cuda(), torch.from_numpy(ops.resize_cubic_taps(H, in_h)).cuda())
                s.u8_resized = torch.empty((H, W, 3), dtype=torch.uint8, device="cuda")
        else:
            assert B == 1
            s.x_f32 = s.io_dev[:pix_bytes].view(torch.float32).view(1, H, W, 3)

        def run():
            if self.device_preprocess:
                for i in range(B):
                    frame = u8[i]
                    if src is not None:                             # shapes.Image.data: INTER_CUBIC resize (+ flip) of the decoded frame
                        frame = ops.resize_cubic_u8(u8[i], H, W, flip=flip, tabs=s.tabs, out=s.u8_resized)
                    ops.preprocess_u8(frame, MEAN_BGR, out=s.x_f32[i:i + 1])     # resnet.preprocess + the f32 feed cast, bit for bit
            return pipe.forward_dev(s.x_f32, dyn=dyn_dev if B > 1 else dyn_dev[0])

        shared = self.in_flight > 1
        # one image in flight: split-K on the small grids (a latency tool); several: plain launches, tiles for a shared chip
        # (fp32 models keep split-K also beside other passes, as bench.py's headline does: the small grids of rpn_conv1 / stage 4 then run
        # the split engine's split-K form; the bf16 engine loses with it when several passes share the chip)
        dtype = getattr(getattr(self.detector, "head", None), "dtype", "f32")
        s.ws = ops.NO_SPLIT_K if ((shared and (dtype == "bf16" or os.environ.get("FRCNN_ENTRY_NO_SPLITK"))) or B > 1) else ops.ConvWorkspace()
        s.io_dev.copy_(s.io_pin)
        s.amax = ops.AmaxArena() if self.f32_engine == "f16x3" else None      # the pass's magnitude records (fixed addresses across replays)
        # Warm-up and capture on ONE side stream, with no device-wide synchronisation: the passes of other geometries that are in flight on
        # the engine's streams keep running (round 6: torch.cuda.graph's enter -- synchronize, empty_cache -- and two more synchronize calls
        # drained them at every capture, and a list of mixed image sizes captures per geometry).  The warm-up pass writes the slot's own
        # buffers, as the replays will: ``s.ready`` orders the first replay behind it.
        side = _capture_stream()
        side.wait_stream(torch.cuda.current_stream())
        stamps.append(("buffers", time.perf_counter()))
        with torch.cuda.stream(side), ops.conv_workspace(s.ws), ops.tile_policy(shared), ops.f32_engine(self.f32_engine), ops.amax_arena(s.amax):
            for _ in range(WARMUP_PASSES):
                run()
            stamps.append(("warm-up enqueue", time.perf_counter()))
            s.ready = torch.cuda.Event()
            s.ready.record(side)
            s.graph = torch.cuda.CUDAGraph()
            s.graph.capture_begin(pool=torch.cuda.graph_pool_handle(), capture_error_mode="thread_local")
            try:
                s.out = run()
                stamps.append(("capture enqueue", time.perf_counter()))
            finally:
                s.graph.capture_end()
        stamps.append(("instantiate", time.perf_counter()))
        packed = s.out["det_packed"]
        s.out_packed = packed if isinstance(packed, (list, tuple)) else [packed]
        s._out_raw = self._pinned.take(4 * B * s.out_packed[0].numel())
        s.out_pin = s._out_raw.view(torch.int32).view((B,) + tuple(s.out_packed[0].shape))
        s.event = torch.cuda.Event()
        s.nbytes = max(int(torch.cuda.memory_reserved() - reserved0), int(s.io_dev.numel() + s.x_f32.numel() * 4))
        stamps.append(("read-back buffers", time.perf_counter()))
        self.capture_seconds += time.perf_counter() - t0
        self._book_capture(t0, stamps)
        return s

    def _book_capture(self, t0, stamps):
        """Where a capture's host time goes (stats()["capture_breakdown_ms"], summed over the engine's captures)."""
        prev = t0
        for name, t in stamps:
            self.capture_breakdown[name] = self.capture_breakdown.get(name, 0.0) + (t - prev) * 1e3
            prev = t

    # ------------------------------------------------------------------ the call
    def _check_epoch(self):
        self.manager.rpn_model._flush_trainer()
        self.detector._flush_trainer()
        if models.weights_epoch() != self._epoch:
            self.cache.clear()
            self._epoch = models.weights_epoch()

    def prefetchable(self, image):
        """True when ``host_pixels(image)`` is pure host work (a JPEG decode) that another thread may do ahead of time; images
        without ``raw`` and foreign preprocess functions are fetched inline (their ``data`` may use the device)."""
        return self.device_preprocess and _declares(image, "raw") and _declares(image, "height")

    def host_pixels(self, image):
        """What ``submit`` uploads for this image -- safe to call from another thread ahead of time (JPEG decode).
        Returns (array, H, W, src or None, flip)."""
        if self.device_preprocess and _declares(image, "raw") and _declares(image, "height"):
            H, W, flip = int(image.height), int(image.width), bool(getattr(image, "flipped", False))
            rgb = getattr(image, "raw_rgb", None) if RGB_UPLOAD else None
            if rgb is not None:
                # a file-backed frame goes up as the decoder delivers it (RGB); the device resize writes B, G, R (flip bit 1) -- a resize to
                # its own size when none is needed: taps {0, 1, 0, 0}, i.e. a copy with the channels swapped
                rgb = np.asarray(rgb)
                if rgb.dtype != np.uint8 or rgb.ndim != 3 or rgb.shape[2] != 3:
                    raise TypeError("image pixels must be uint8 (h, w, 3) (shapes.py:19-29), got %s %s" % (rgb.dtype, rgb.shape))
                return rgb, H, W, (int(rgb.shape[0]), int(rgb.shape[1])), 2 | int(flip)
            raw = np.asarray(image.raw)
            if raw.dtype != np.uint8 or raw.ndim != 3 or raw.shape[2] != 3:
                raise TypeError("image pixels must be uint8 BGR (h, w, 3) (shapes.py:19-29), got %s %s" % (raw.dtype, raw.shape))
            if raw.shape[:2] == (H, W) and not flip:
                return raw, H, W, None, False
            return raw, H, W, (int(raw.shape[0]), int(raw.shape[1])), flip
        data = image.data
        if self.device_preprocess:
            data = np.asarray(data)
            if data.dtype != np.uint8:
                raise TypeError("image.data must be uint8 BGR (shapes.py:19-29), got %s" % data.dtype)
        else:
            data = self.manager.preprocess_func(data)               # det_util.py:36 (float64 on the host, cast on feed)
        return data, int(data.shape[0]), int(data.shape[1]), None, False

    def plan_canvases(self, counts):
        """Before a call over a list whose sizes are known (voc_dets.get_dets_by_cls reads the headers): the canvas classes for that
        histogram {(H, W): images} (plan_canvas_classes; classes with a captured pass count as free) and how many passes each may
        hold in flight -- its share of the list times ``in_flight``, at least one, two when it fills more than one pass."""
        existing = {k[1:3] for k in self.cache.keys() if k[:1] == ("canvas",)}
        plan = plan_canvas_classes(counts, existing=existing)
        self._canvas_of.update(plan)
        per_class, total = collections.Counter(), float(sum(counts.values())) or 1.0
        for g, c in plan.items():
            per_class[c] += counts[g]
        for c, n in per_class.items():
            want = int(np.ceil(CANVAS_SLOT_FACTOR * self.in_flight * n / total - 1e-9))
            self._canvas_slots[("canvas",) + c] = max(1 if (n <= self.batch or CANVAS_SLOT_MIN < 2) else 2, min(self.in_flight, want))
            if want >= self.in_flight:                            # the class that fills every stream: one pass more, queued behind the oldest
                self._canvas_slots[("canvas",) + c] += CANVAS_EXTRA_SLOT
        # ... within the cache's byte budget: passes beyond it would only evict each other (a pass costs what one of its class did, else
        # CANVAS_BYTES_PER_PIXEL of its canvases: ~1.4 GB per four 600 x 1000 images)
        def cost(c):
            known = [sl.nbytes for k, v in self.cache._slots.items() if k[:3] == ("canvas",) + c for sl in v]
            return max(known) if known else CANVAS_BYTES_PER_PIXEL * self.batch * c[0] * c[1]
        live = [c for c in per_class]
        while sum(self._canvas_slots[("canvas",) + c] * cost(c) for c in live) > 0.9 * self.cache.byte_budget:
            c = max(live, key=lambda k: (self._canvas_slots[("canvas",) + k], -per_class[k]))     # (the fullest allowance; among equals the rarest class)
            if self._canvas_slots[("canvas",) + c] <= 1:
                break
            self._canvas_slots[("canvas",) + c] -= 1
        return plan

    def canvas_class(self, H, W):
        """The canvas (Hc, Wc) of a true size: the plan's, else the smallest known class that holds it with at most a third more
        pixels, else its own smallest canvas (which becomes a class)."""
        c = self._canvas_of.get((H, W))
        if c is None:
            own = (canvas_side(H), canvas_side(W))
            fits = [k for k in set(self._canvas_of.values()) if k[0] >= own[0] and k[1] >= own[1] and 3 * k[0] * k[1] <= 4 * own[0] * own[1]]
            c = self._canvas_of[(H, W)] = min(fits, key=lambda k: (k[0] * k[1], k)) if fits else own
            if len(self._canvas_of) > 4096:
                self._canvas_of.clear()
        return c

    def canvas_ok(self, pixels):
        """Can this frame go through a canvas pass?  (device preprocess, a source no larger than its canvas)"""
        arr, H, W, src, flip = pixels
        if not (self.canvas_capable and getattr(arr, "dtype", None) == np.uint8):
            return False
        Hc, Wc = self.canvas_class(H, W)
        return int(np.prod(arr.shape)) <= Hc * Wc * 3

    @staticmethod
    def geometry_of(pixels):
        """The exact geometry of a ``host_pixels`` result (resized size, source size, flip)."""
        _, H, W, src, flip = pixels
        return (H, W) if src is None else (H, W) + src + (flip,)

    def geometry(self, pixels):
        """The captured-pass key of a ``host_pixels`` result: images with equal keys can share a batched pass.  In canvas mode the key
        is the canvas CLASS of the image's size."""
        if self.canvas and self.canvas_ok(pixels):
            return ("canvas",) + self.canvas_class(pixels[1], pixels[2])
        return self.geometry_of(pixels)

    def probe_geometry(self, image):
        """``geometry(host_pixels(image))`` WITHOUT decoding the pixels (a file's header gives its size), or None when that cannot be
        known cheaply.  get_dets_by_cls counts how often each geometry occurs in its list before it decides what to capture."""
        if not (self.device_preprocess and hasattr(image, "raw_size") and hasattr(image, "height")):
            return None
        try:
            size = image.raw_size()
        except Exception:                                       # noqa: BLE001 -- an unreadable header: the decode will say what is wrong
            return None
        if size is None:
            return None
        H, W, flip = int(image.height), int(image.width), bool(getattr(image, "flipped", False))
        if self.canvas and self.canvas_capable:
            Hc, Wc = self.canvas_class(H, W)
            if size[0] * size[1] <= Hc * Wc:
                return ("canvas", Hc, Wc)
        if RGB_UPLOAD and getattr(image, "_pixels", 0) is None and hasattr(type(image), "raw_rgb"):      # file-backed: uploaded as RGB (host_pixels; the
            # attribute is looked up on the CLASS: hasattr on the instance would run the property, i.e. decode the JPEG)
            return (H, W, int(size[0]), int(size[1]), 2 | int(flip))
        if tuple(size) == (H, W) and not flip:
            return (H, W)
        return (H, W, int(size[0]), int(size[1]), flip)

    def should_wait(self, pixels, batch):
        """Before a ``submit_batch`` of these frames: True when every captured pass of their canvas class is in flight and the class
        already has the passes its plan allows (CANVAS_SLOTS_PER_CLASS outside a plan) -- collecting an older ticket frees one sooner
        than a capture would make another."""
        key = self.geometry(pixels[0])
        if key[0] != "canvas":
            return False
        slots = self.cache._slots.get(key + ((batch,) if batch > 1 else ()))
        return slots is not None and len(slots) >= self._canvas_slots.get(key, CANVAS_SLOTS_PER_CLASS) and all(sl.busy for sl in slots)

    def exact_geometry(self, image):
        """``probe_geometry`` with canvas mode off: the image's OWN geometry (get_dets_by_cls counts the distinct ones)."""
        was, self.canvas = self.canvas, False
        try:
            return self.probe_geometry(image)
        finally:
            self.canvas = was

    def has_geometry(self, key):
        """A captured pass of this geometry (one image per pass, or the batched form) is already in the cache."""
        return key in self.cache._slots or (key + (self.batch,)) in self.cache._slots

    def submit(self, image, resize_ratio, det_threshold=0.0, pixels=None):
        """``pixels``: the result of ``host_pixels(image)`` when the caller fetched it ahead of time."""
        return self.submit_batch([image], [resize_ratio], det_threshold, [self.host_pixels(image) if pixels is None else pixels], batch=1)

    def submit_batch(self, images, resize_ratios, det_threshold, pixels, batch=None):
        """Up to ``batch`` images of ONE geometry (``geometry(pixels[i])`` equal) in one captured pass; a short group is padded with
        copies of its first frame, whose results nobody reads.  ``collect_batch`` returns the images' results in order."""
        self._check_epoch()
        B = self.batch if batch is None else batch
        assert 1 <= len(images) <= B and len(images) == len(pixels) == len(resize_ratios)
        _, H, W, src, flip = pixels[0]
        key = self.geometry(pixels[0])
        assert all(self.geometry(p) == key for p in pixels), "one pass, one geometry"
        if key[0] == "canvas":
            return self._submit_canvas(images, resize_ratios, det_threshold, pixels, B)
        s = self.cache.acquire(key + ((B,) if B > 1 else ()), lambda: self._capture(H, W, src, flip, B))
        for i in range(B):
            j = i if i < len(images) else 0
            np.copyto(s.pix_hosts[i], pixels[j][0], casting="same_kind")      # into pinned memory (f64 -> f32 cast for a foreign preprocess)
            s.dyn_host[i, 0], s.dyn_host[i, 1] = float(resize_ratios[j]), float(det_threshold)
        st = self._streams[self._seq % self.in_flight]
        self._seq += 1
        with torch.cuda.stream(st):
            if s.ready is not None:
                st.wait_event(s.ready)                             # (the first replay of a fresh pass: behind its warm-up)
                s.ready = None
            s.io_dev.copy_(s.io_pin, non_blocking=True)
            s.graph.replay()
            for i in range(len(images)):
                s.out_pin[i].copy_(s.out_packed[i], non_blocking=True)
            s.event.record(st)
        s.busy = True
        return Ticket(s, list(images))

    def collect_batch(self, ticket):
        """-> [(num_rois, dets)] for the images of a ``submit_batch`` ticket."""
        s = ticket.slot
        try:
            s.event.synchronize()
            rev = self.rev_class_mapping
            res = []
            if s.amax is not None:
                bits = int(s.out_pin[0].numpy()[2])                 # the pass's f16x3 status word (pipeline._pass_status)
                if bits:
                    raise FrcnnError("f16x3 pass not trustworthy (%s): status %d; run the image on the native or bf16x6 engine" % (ops.h3_status_text(bits), bits))
            for i in range(len(ticket.image)):
                packed = s.out_pin[i].numpy()
                nd, n_rois = int(packed[0]), int(packed[1])
                rows = (packed.size - 4) // 7
                bbox = packed[4:4 + 4 * nd].reshape(nd, 4).astype(np.int64)
                cls = packed[4 + 4 * rows:4 + 4 * rows + nd].copy()
                prob = packed[4 + 5 * rows:4 + 5 * rows + nd].view(np.float32).copy()
                res.append((n_rois, [{"bbox": bbox[k], "cls_name": rev[int(cls[k])], "prob": prob[k]} for k in range(nd)]))
            return res
        finally:
            s.busy = False                               # whatever happened while decoding the records: the slot is free again

    def collect(self, ticket):
        return self.collect_batch(ticket)[0]

    def stats(self):
        c = self.cache
        return {"graphs": len(c), "sizes": len(c.keys()), "bytes": c.nbytes, "byte_budget": c.byte_budget, "captures": c.captures,
                "hits": c.hits, "evictions": c.evictions, "capture_seconds": round(self.capture_seconds, 3), "in_flight": self.in_flight,
                "device_preprocess": self.device_preprocess, "f32_engine": self.f32_engine, "images_per_pass": self.batch,
                "capture_breakdown_ms": {k: round(v, 1) for k, v in self.capture_breakdown.items()}}


def for_models(manager, detector, num_rois=64, stride=16, in_flight=1):
    """The DetectionEntry of this (manager, detector, num_rois, stride, in_flight), built on first use and kept on the
    manager; None when the pair cannot take the captured path (voc_dets then runs the eager one)."""
    if not torch.cuda.is_available() or not DetectionEntry.usable(manager, detector, num_rois):
        return None
    table = manager.__dict__.setdefault("_entries", {})
    key = (id(detector), int(num_rois), stride, int(in_flight))
    eng = table.get(key)
    if eng is None or eng.detector is not detector:
        global _QUEUES_SAID
        if not _QUEUES_SAID and in_flight > 1 and int(os.environ.get("GPU_MAX_HW_QUEUES", "4")) < 8:
            _QUEUES_SAID = True
            import warnings
            warnings.warn("faster_rcnn_amd.entry: this process runs on %s hardware queues (GPU_MAX_HW_QUEUES; ROCm's default is 4): get_dets_by_cls measures "
                          "~9 %% slower there than on 8 (497 against 542-547 img/s).  The setting is read when the HIP runtime starts: export "
                          "GPU_MAX_HW_QUEUES=8 before the first device call, as voc_dets.main does." % os.environ.get("GPU_MAX_HW_QUEUES", "4"))
        eng = table[key] = DetectionEntry(manager, detector, num_rois, stride, in_flight)
    return eng


_QUEUES_SAID = False
