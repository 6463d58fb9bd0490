"""Mirror of the reference's voc_dets.py inference entry points (voc_dets.py:17-129).

``get_dets(training_manager, detector, image, resize_ratio, num_rois, stride, det_threshold)``
returns the same list of ``{'bbox': int array [x1,y1,x2,y2], 'cls_name', 'prob'}`` dicts in the
same order.

Two paths behind the same call:
* **captured** (entry.DetectionEntry; taken whenever the manager's RPN model and the detector are this package's
  models): one H2D copy of the uint8 frame, ONE hipGraph per image from preprocess to post-process, one packed D2H
  copy; graphs cached per image size; ``get_dets_by_cls`` keeps several images in flight.  No host round trip between
  the RPN and the detector (the reference makes two per image: det_util.py:41, 49-55).
* **eager** (``FAST_ENTRY = False``, or a foreign Keras-style ``detector``): ``get_det_inputs`` exactly as the reference
  sequences it (conv map and RoIs come back as numpy), the detector once over all padded RoIs, and the arg-max / decode /
  per-class NMS / rescale tail in one device kernel (frcnn_detections) instead of the reference's Python loops.
"""
import collections
import os
import timeit

import numpy as np
import torch

from . import entry, nets, ops
from .det_util import DetTrainingManager, nms  # noqa: F401  (re-exported like the reference module)

DEFAULT_DET_THRESHOLD = 0.0
# get_dets_by_cls: threads fetching the next images' pixels (0: always inline).  They are only started when a frame's decode is slow
# enough to matter (DECODE_INLINE_MS; the device needs ~1.85 ms per image): 500x375 VOC JPEGs decode in 1.0 ms inline and threads add
# nothing there (508-518 img/s with 0 / 1 / 2 / 4 threads, round 6) while a multi-megapixel photograph needs them.  (Rounds 4-5
# measured threads SLOWER, 257 against 310-318: that was the hidden second decode of every frame -- entry.py's hasattr on a property --
# running under the interpreter lock in every worker.)
DECODE_THREADS = int(os.environ.get("FRCNN_DECODE_THREADS", "4"))
DECODE_INLINE_MS = float(os.environ.get("FRCNN_DECODE_INLINE_MS", "4.0"))
FAST_ENTRY = os.environ.get("FRCNN_ENTRY_EAGER", "0") == "0"      # False: always the eager path (tests and bench.py compare the two)


def _pad_rois(rois, num_rois):
    """voc_dets.py:31-47: the last batch is padded with copies of ITS first RoI."""
    n = rois.shape[0]
    if n % num_rois == 0:
        return rois
    last = (n // num_rois) * num_rois
    extra = np.tile(rois[last], (num_rois - (n - last), 1))
    return np.concatenate([rois, extra])


def get_dets(training_manager, detector, image, resize_ratio, num_rois=64, stride=16,
             det_threshold=DEFAULT_DET_THRESHOLD):
    eng = entry.for_models(training_manager, detector, num_rois, stride, in_flight=1) if FAST_ENTRY else None
    if eng is not None:
        num_boxes, dets = eng.collect(eng.submit(image, resize_ratio, det_threshold))
        print("num rois: {}".format(num_boxes))
        return dets
    return _get_dets_eager(training_manager, detector, image, resize_ratio, num_rois, stride, det_threshold)


def _get_dets_eager(training_manager, detector, image, resize_ratio, num_rois, stride, det_threshold):
    conv_out, rois = training_manager.get_det_inputs(image)
    class_mapping = training_manager.class_mapping
    rev_class_mapping = dict((v, k) for k, v in class_mapping.items())
    num_boxes = rois.shape[0]
    print("num rois: {}".format(num_boxes))
    if num_boxes == 0:
        return []
    padded = _pad_rois(np.asarray(rois, dtype=np.float32), num_rois)
    rois_d = torch.from_numpy(np.ascontiguousarray(padded)).cuda()
    if hasattr(detector, "forward_dev"):
        out_cls, out_reg = detector.forward_dev(nets.to_device_image(conv_out), rois_d)
    else:                                   # any other object with the Keras predict() contract
        cls_parts, reg_parts = [], []
        for b in range(0, len(padded), num_rois):
            c, r = detector.predict([conv_out, padded[None, b:b + num_rois]])
            cls_parts.append(np.asarray(c)[0])
            reg_parts.append(np.asarray(r)[0])
        out_cls = torch.from_numpy(np.concatenate(cls_parts).astype(np.float32)).cuda()
        out_reg = torch.from_numpy(np.concatenate(reg_parts).astype(np.float32)).cuda()
    n_rows = torch.tensor([len(padded)], dtype=torch.int32, device="cuda")
    res = ops.detections(rois_d, n_rows, out_cls, out_reg, num_rois, class_mapping["bg"], det_threshold, stride, resize_ratio)
    n_dets, bbox, cls, prob, _ = ops.split_detections(res["det_packed"].cpu())      # one copy for all five outputs
    nd = int(n_dets.item())
    det_cls, det_prob, det_bbox = cls.numpy()[:nd], prob.numpy()[:nd], bbox.numpy()[:nd]
    return [{"bbox": det_bbox[i].astype(np.int64), "cls_name": rev_class_mapping[int(det_cls[i])], "prob": det_prob[i]}
            for i in range(nd)]


def get_dets_by_cls(training_manager, detector, resized_ratios, images, stride=16, det_threshold=DEFAULT_DET_THRESHOLD):
    """voc_dets.py:91-111.  On the captured path the images are pipelined: up to ``entry.default_in_flight()`` are enqueued
    (own HIP stream each) before the oldest one's detections are read, and results are folded into the dict in list order,
    so the dict -- keys, per-image lists, their order -- is the one the reference's one-by-one loop builds."""
    dets_by_cls = {}

    def fold(image, dets, start_time):
        for det in dets:
            dets_by_cls.setdefault(det["cls_name"], {}).setdefault(image.name, []).append(det)
        print("image {} ran in {} seconds".format(image.name, timeit.default_timer() - start_time))

    eng = None
    if FAST_ENTRY:
        dtype = getattr(getattr(detector, "head", None), "dtype", "f32")
        eng = entry.for_models(training_manager, detector, 64, stride, in_flight=entry.default_in_flight(dtype))
    if eng is None:
        for image, resized_ratio in zip(images, resized_ratios):
            start_time = timeit.default_timer()
            dets = get_dets(training_manager, detector, image, resized_ratio, stride=stride, det_threshold=det_threshold)
            fold(image, dets, start_time)
        return dets_by_cls
    return _get_dets_by_cls_captured(eng, training_manager, detector, resized_ratios, images, stride, det_threshold, fold, dets_by_cls)


# A list of mixed image sizes (voc_dets.py:91-111 walks whatever list it is given; shapes.py:106-123 gives every source size its own
# resized geometry): a captured pass serves ONE geometry, and four images of one geometry share a pass.  Neighbours rarely share one,
# so the images are held back per geometry -- at most REORDER_WINDOW of them, counted over all geometries -- until a geometry has a
# full pass; results are folded into the dict in LIST order whatever order they were computed in.  A geometry that the list holds
# fewer than CAPTURE_MIN times is not worth a capture (~20 ms beside passes in flight and 0.4-1.4 GB, against ~7 ms for the eager
# sequence): those images take the eager path (same detections up to the summation order of a few layers, tests/test_entry_gpu.py).
# Measured on bench.py's mixed_sizes leg (256 frames, 36 geometries, first call / same call again): CAPTURE_MIN 1: 170 / 514 img/s
# (45 captures), 2: 300 / 442 (22 captures, 23 eager), 3: 274 / 428.  A long run over a whole dataset amortises every capture
# (FRCNN_ENTRY_CAPTURE_MIN=1); the default favours the call that sees its list once.
# passes submitted beyond the engine's streams before the oldest is collected (dev knob; 0, 1, 2 and 4 all measure 490-496 img/s on four
# hardware queues: a stream idling while the host stages the next frames is NOT what holds get_dets_by_cls back there -- the queues are)
WINDOW_EXTRA = int(os.environ.get("FRCNN_ENTRY_WINDOW_EXTRA", "0"))
ENTRY_HW_QUEUES = "8"                            # what main() asks the runtime for (GPU_MAX_HW_QUEUES) unless the environment says otherwise
REORDER_WINDOW = int(os.environ.get("FRCNN_ENTRY_REORDER", "64"))
CAPTURE_MIN = int(os.environ.get("FRCNN_ENTRY_CAPTURE_MIN", "2"))


def _get_dets_by_cls_captured(eng, training_manager, detector, resized_ratios, images, stride, det_threshold, fold, dets_by_cls):
    from concurrent.futures import ThreadPoolExecutor
    images, resized_ratios = list(images), list(resized_ratios)
    n = min(len(images), len(resized_ratios))
    # how often each geometry occurs in the list, from the images' headers (no pixel decode); unknown (a foreign image object): always capture
    # a list of many sizes: passes per canvas CLASS with the true sizes as device values (entry.py, round 6) instead of one per geometry
    if getattr(eng, "canvas_capable", False):
        exact = [eng.exact_geometry(images[i]) for i in range(n)]
        eng.canvas = len({k for k in exact if k is not None}) > entry.CANVAS_MIN_GEOMETRIES
        if eng.canvas:                                   # the classes for THIS list's histogram of sizes: few, each worth its captures
            eng.plan_canvases(collections.Counter(k[:2] for k in exact if k is not None))
    keys = [eng.probe_geometry(images[i]) for i in range(n)]
    counts = collections.Counter(k for k in keys if k is not None)
    unprobed = set(i for i in range(n) if keys[i] is None)

    def worth_a_capture(pos):
        k = keys[pos]
        return pos in unprobed or counts[k] >= CAPTURE_MIN or eng.has_geometry(k)
    window = collections.deque()                     # (positions, tickets) of passes in flight, oldest first
    done = {}                                        # position -> (num_boxes or None, dets, start_time): computed, not yet folded
    next_fold = 0

    def fold_ready():
        nonlocal next_fold
        while next_fold in done:
            num_boxes, dets, start_time = done.pop(next_fold)
            if num_boxes is not None:
                print("num rois: {}".format(num_boxes))
            fold(images[next_fold], dets, start_time)
            next_fold += 1

    def finish():
        part, ticket = window[0]                             # (taken off the window only once collected: a failing collect leaves no
        try:                                                 #  ticket behind that nobody can see)
            results = eng.collect_batch(ticket)
        finally:
            window.popleft()
        for (pos, start_time), (num_boxes, dets) in zip(part, results):
            done[pos] = (num_boxes, dets, start_time)
        fold_ready()

    def flush(group, whole_only=False):
        """Items (position, pixels, start_time) of ONE geometry: whole batched passes where the engine has them, single-image passes
        for what is left unless it fills at least half a batch; with ``whole_only`` what does not fill a pass is handed back."""
        B = eng.batch
        if group and not worth_a_capture(group[0][0]):
            if whole_only:
                return group
            import contextlib
            import io
            for pos, pixels, start_time in group:            # a rare geometry: the eager sequence (the host waits for it; passes in flight keep running)
                said = io.StringIO()
                with contextlib.redirect_stdout(said):       # its "num rois" line is printed when the image's turn in the LIST comes
                    dets = _get_dets_eager(training_manager, detector, images[pos], resized_ratios[pos], eng.num_rois, stride, det_threshold)
                line = said.getvalue().strip().splitlines()
                done[pos] = (line[0][len("num rois: "):] if line and line[0].startswith("num rois: ") else None, dets, start_time)
            fold_ready()
            return []
        canvas = B > 1 and isinstance(keys[group[0][0]], tuple) and keys[group[0][0]][:1] == ("canvas",) if group else False
        while group:
            if B > 1 and len(group) >= B:
                take = B
            elif whole_only:
                break
            else:
                # (a canvas class keeps ONE pass shape: what is left of it goes through a padded pass rather than a capture of its own)
                take = B if (B > 1 and (canvas or len(group) >= max(2, B // 2))) else 1
            part, group = group[:take], group[take:]
            while window and hasattr(eng, "should_wait") and eng.should_wait([g[1] for g in part], B if take > 1 else 1):
                finish()                                     # every pass of this canvas class is in flight: collect the oldest ticket first
            ticket = eng.submit_batch([images[g[0]] for g in part], [resized_ratios[g[0]] for g in part], det_threshold, [g[1] for g in part],
                                      batch=B if take > 1 else 1)
            window.append(([(g[0], g[2]) for g in part], ticket))
            if len(window) >= eng.in_flight + WINDOW_EXTRA:
                finish()
        return group

    # the pixels are fetched inline, as the reference does (shapes.py:19-29), until that proves slow (DECODE_INLINE_MS); from then on
    # the NEXT images are fetched on a few threads while the GPU works (PIL's JPEG decode releases the GIL)
    ahead = 2 * eng.in_flight * eng.batch
    pool = None
    pending = {}
    slow_fetches = 0
    buckets = collections.OrderedDict()              # geometry -> items held back, in list order; dict order = age of the oldest item
    held = 0
    try:
        for i in range(n):
            if pool is not None:
                for j in range(i, min(n, i + ahead)):
                    if j not in pending and eng.prefetchable(images[j]):
                        pending[j] = pool.submit(eng.host_pixels, images[j])
            start_time = timeit.default_timer()
            if i in pending:
                pixels = pending.pop(i).result()
            else:
                pixels = eng.host_pixels(images[i])
                slow_fetches = slow_fetches + 1 if (timeit.default_timer() - start_time) * 1e3 > DECODE_INLINE_MS else 0
                if slow_fetches >= 2 and pool is None and DECODE_THREADS > 0 and i + 1 < n and eng.prefetchable(images[i]):
                    pool = ThreadPoolExecutor(max_workers=DECODE_THREADS)       # (two slow fetches in a row: not a cold file cache)
            key = eng.geometry(pixels)
            if keys[i] != key:                               # (the header said otherwise, or nothing: the decoded frame decides)
                if keys[i] is not None:
                    counts[keys[i]] -= 1
                keys[i] = key
                counts[key] += 1
            bucket = buckets.setdefault(key, [])
            bucket.append((i, pixels, start_time))
            held += 1
            if eng.batch == 1 or len(bucket) >= eng.batch:
                held -= len(bucket)
                rest = flush(buckets.pop(key), whole_only=eng.batch > 1)
                if rest:
                    buckets[key] = rest
                    held += len(rest)
            while held > REORDER_WINDOW:                     # the oldest geometry has waited long enough: its items go as they are
                k0 = next(iter(buckets))
                held -= len(buckets[k0])
                flush(buckets.pop(k0))
        for k0 in list(buckets):
            flush(buckets.pop(k0))
        while window:
            finish()
        fold_ready()
        assert next_fold == n, "get_dets_by_cls: %d of %d images folded" % (next_fold, n)
    finally:
        if pool is not None:
            pool.shutdown(wait=True, cancel_futures=True)
        for _, ticket in window:                       # an exception mid-list: no slot stays marked busy
            ticket.slot.event.synchronize()
            ticket.slot.busy = False
    return dets_by_cls


def write_dets(dets, out_dir):
    """voc_dets.py:114-129: comp3_det_test_<cls>.txt, 'name prob x1 y1 x2 y2' with +1 coords."""
    os.makedirs(out_dir, exist_ok=True)
    for cls_name in dets:
        file_path = os.path.join(out_dir, "comp3_det_test_{}.txt".format(cls_name))
        with open(file_path, "w") as outfile:
            for image_name in dets[cls_name]:
                for det in dets[cls_name][image_name]:
                    x1, y1, x2, y2 = det["bbox"] + 1
                    outfile.write("{} {} {} {} {} {}\n".format(image_name, det["prob"], x1, y1, x2, y2))


def build_parser():
    """The reference's command line (voc_dets.py:132-160): two positional checkpoints + the same flags."""
    import argparse
    p = argparse.ArgumentParser(description="Run the RPN + detector over a VOC-style image set and write comp3_det_test_<cls>.txt files")
    p.add_argument("step3_model_path", help="weights of the RPN trained in step 3 (Keras .h5 or this package's .npz)")
    p.add_argument("step4_model_path", help="weights of the detector trained in step 4 (must share the RPN's base)")
    p.add_argument("--voc_path", dest="voc_path", required=True, help="base path of the VOC-style test set")
    p.add_argument("--kitti", dest="kitti", action="store_true", help="KITTI classes instead of Pascal VOC")
    p.add_argument("--img_set", dest="img_set", choices=("val", "test", "trainval"), default="val")
    p.add_argument("--resize_dims", dest="resize_dims", default="600,1000")
    p.add_argument("--anchor_scales", dest="anchor_scales", default="128,256,512")
    p.add_argument("--network", dest="network", choices=("vgg16", "resnet50", "resnet101"), default="vgg16")
    p.add_argument("--out_dir", dest="out_dir", default=".")
    p.add_argument("--det_threshold", dest="det_threshold", default=DEFAULT_DET_THRESHOLD)
    return p


def main(argv=None):
    """voc_dets.py:161-192: load the two models, resize the image set, detect, write the per-class files."""
    from . import resnet, vgg
    from .args_util import anchor_scales_from_str, base_paths_to_imgs, resize_dims_from_str
    from .data.voc_data_helpers import KITTI_CLASS_MAPPING, VOC_CLASS_MAPPING
    from .util import get_anchors, resize_imgs
    args = build_parser().parse_args(argv)
    # four passes in flight -- staging copy, replay, read-back each -- want more hardware queues than the runtime's default four (497 ->
    # 542-547 img/s on eight, entry.default_in_flight); read when the HIP runtime starts, an explicit setting wins
    os.environ.setdefault("GPU_MAX_HW_QUEUES", ENTRY_HW_QUEUES)
    test_imgs = base_paths_to_imgs(args.voc_path, img_set=args.img_set, do_flip=False)
    anchors = get_anchors(anchor_scales_from_str(args.anchor_scales))
    print("num test_imgs: ", len(test_imgs))
    class_mapping = KITTI_CLASS_MAPPING if args.kitti else VOC_CLASS_MAPPING
    if args.network == "vgg16":
        model_rpn = vgg.rpn_from_h5(args.step3_model_path, anchors_per_loc=len(anchors))
        model_det = vgg.det_from_h5(args.step4_model_path, num_classes=len(class_mapping))
        stride, preprocess = vgg.STRIDE, vgg.preprocess
    else:
        depth = 50 if args.network == "resnet50" else 101
        model_rpn = resnet.rpn_from_h5(args.step3_model_path, anchors_per_loc=len(anchors), depth=depth)
        model_det = resnet.det_from_h5(args.step4_model_path, num_classes=len(class_mapping), depth=depth)
        stride, preprocess = resnet.STRIDE, resnet.preprocess
    manager = DetTrainingManager(rpn_model=model_rpn, class_mapping=class_mapping, preprocess_func=preprocess, anchor_dims=anchors)
    resize_min, resize_max = resize_dims_from_str(args.resize_dims)
    processed_imgs, resized_ratios = resize_imgs(test_imgs, min_size=resize_min, max_size=resize_max)
    dets = get_dets_by_cls(manager, model_det, resized_ratios, processed_imgs, stride=stride, det_threshold=float(args.det_threshold))
    write_dets(dets, args.out_dir)
    return dets


if __name__ == "__main__":
    main()
