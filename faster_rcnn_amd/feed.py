"""Device-side feed of the training managers' fast paths (rpn_util.RpnTrainingManager, det_util.DetTrainingManager).

The reference hands ``train_on_batch`` host arrays: ``np.expand_dims(preprocess_func(image.data), 0)`` -- a float64 (1,H,W,3)
array, 14 MB for a 600x1000 frame (rpn_util.py:50-51, det_util.py:35-36) -- and the step's feed casts it to float32 and uploads
it.  When the manager's ``preprocess_func`` is one of this package's mean subtractions (resnet.preprocess / vgg.preprocess) the
same float32 tensor is produced ON the device from the decoded uint8 frame: one 0.6-1.8 MB upload, the INTER_CUBIC resize and
flip of ``shapes.Image.data`` (frcnn_resize_cubic_u8, the same integers as the host restatement) and frcnn_preprocess_u8 (bit for
bit float32(float64(pixel) - mean)).  tests/test_train_loop_gpu.py holds the two feeds to the same bits.
"""
import os
import numpy as np
import torch

from . import ops
from .shapes import declares as _declares

MEAN_BGR = (103.939, 116.779, 123.68)           # resnet.preprocess / vgg.preprocess (resnet.py:64-75, vgg.py:52-57)
RGB_UPLOAD = os.environ.get("FRCNN_FEED_RGB_UPLOAD", "1") != "0"


class _PinRing:
    """A few grow-only pinned staging areas used in turn: ``upload(array)`` copies the array into the next one and starts an
    ASYNCHRONOUS copy to the device on the current stream (a pageable source makes the runtime stage the bytes itself and block the
    host for the whole transfer: 0.25 ms for a 1.8 MB frame, on a loop whose iteration is 2 ms).  An area is reused only after the
    copy out of it has finished (its event)."""

    def __init__(self, n=6):
        self.slots = [[None, None] for _ in range(n)]
        self.i = 0

    def upload(self, arr):
        arr = np.ascontiguousarray(arr)
        k, self.i = self.i, (self.i + 1) % len(self.slots)
        slot = self.slots[k]
        if slot[1] is not None:
            slot[1].synchronize()
        nbytes = max(arr.nbytes, 16)
        if slot[0] is None or slot[0].numel() < nbytes:
            slot[0] = torch.empty(int(nbytes * 1.25) + 256, dtype=torch.uint8).pin_memory()
            slot[1] = torch.cuda.Event()
        pin = slot[0][:arr.nbytes]
        np.copyto(pin.numpy().view(arr.dtype).reshape(arr.shape), arr)
        dev = pin.view(_TORCH_DTYPE[arr.dtype.type]).view(arr.shape).to("cuda", non_blocking=True)
        slot[1].record()
        return dev

    def host_slot(self, shape, dtype):
        """A pinned host tensor for a device -> host copy (e.g. two counters), valid until the ring comes round again."""
        k, self.i = self.i, (self.i + 1) % len(self.slots)
        slot = self.slots[k]
        if slot[1] is not None:
            slot[1].synchronize()
        n = int(np.prod(shape)) * torch.empty((), dtype=dtype).element_size()
        if slot[0] is None or slot[0].numel() < n:
            slot[0] = torch.empty(max(256, n), dtype=torch.uint8).pin_memory()
            slot[1] = torch.cuda.Event()
        return slot[0][:n].view(dtype).view(shape), slot[1]


_TORCH_DTYPE = {np.uint8: torch.uint8, np.int32: torch.int32, np.float32: torch.float32, np.int16: torch.int16, np.float64: torch.float64}
_RING = None


def ring():
    global _RING
    if _RING is None:
        _RING = _PinRing()
    return _RING


def upload(arr):
    """numpy array -> device tensor through pinned staging, asynchronous on the current stream."""
    return ring().upload(arr)


def device_preprocess(preprocess_func):
    from . import resnet, vgg
    return preprocess_func in (resnet.preprocess, vgg.preprocess)


# The JPEG decode of a file-backed frame (1.0 ms for a 500 x 375 VOC image) on the training loop's thread made a mixed-precision RPN iteration
# host-bound: 1.45 ms from files against 1.03 from memory.  train_util's loops name the image TWO iterations ahead (when fetching it needs no
# shuffle) and ONE background thread decodes it -- PIL releases the interpreter lock inside its decoder -- while the loop enqueues the
# step between.  FRCNN_FEED_DECODE_AHEAD=0: decode on the loop's thread.
DECODE_AHEAD = os.environ.get("FRCNN_FEED_DECODE_AHEAD", "1") != "0"
_DECODER = None
_DECODED = {}                                    # id(image) -> (image, Future of its raw_rgb)


def decode_ahead(image):
    """Start decoding a file-backed image's pixels on the background thread (device_image picks the result up)."""
    global _DECODER
    if not (DECODE_AHEAD and RGB_UPLOAD and hasattr(type(image), "raw_rgb")) or getattr(image, "_pixels", 0) is not None or id(image) in _DECODED:
        return
    if _DECODER is None:
        from concurrent.futures import ThreadPoolExecutor
        _DECODER = ThreadPoolExecutor(max_workers=1, thread_name_prefix="frcnn-decode")
    if len(_DECODED) > 8:
        _DECODED.clear()                                      # (frames asked for and never taken: forget them)
    _DECODED[id(image)] = (image, _DECODER.submit(lambda: image.raw_rgb))


def _raw_rgb(image):
    ent = _DECODED.pop(id(image), None)
    if ent is not None and ent[0] is image:
        return ent[1].result()
    return image.raw_rgb


def device_image(image, preprocess_func):
    """(1,H,W,3) float32 device tensor == float32(np.expand_dims(preprocess_func(image.data), 0)), on the current stream."""
    if device_preprocess(preprocess_func) and _declares(image, "raw") and _declares(image, "height"):
        H, W, flip = int(image.height), int(image.width), bool(getattr(image, "flipped", False))
        # a file-backed frame goes up in the JPEG decoder's channel order; the device resize (to its own size when none is needed: a
        # copy) writes B, G, R -- the host's channel reversal cost as much as half the decode (round 6, as entry.DetectionEntry)
        rgb = _raw_rgb(image) if (RGB_UPLOAD and hasattr(type(image), "raw_rgb")) else None
        if rgb is not None:
            rgb = np.asarray(rgb)
            if rgb.dtype == np.uint8 and rgb.ndim == 3 and rgb.shape[2] == 3:
                return ops.preprocess_u8(ops.resize_cubic_u8(upload(rgb), H, W, flip=2 | int(flip)), MEAN_BGR)
        raw = np.asarray(image.raw)
        if raw.dtype == np.uint8 and raw.ndim == 3 and raw.shape[2] == 3:
            if raw.shape[0] != H or raw.shape[1] != W:
                u8 = ops.resize_cubic_u8(upload(raw), H, W, flip=flip)
            else:
                u8 = upload(raw[:, ::-1] if flip else raw)
            return ops.preprocess_u8(u8, MEAN_BGR)
    x = np.expand_dims(preprocess_func(image.data), axis=0)
    t = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).cuda()
    return t if t.dim() == 4 else t.unsqueeze(0)


class Ready:
    """Marks device tensors produced on a side stream: ``tensor._ready`` is an event the consuming streams wait for
    (train._StepDriver._run_step), and the tensors are registered with those streams so the caching allocator does not recycle them
    under a kernel that still reads them."""

    @staticmethod
    def mark(*tensors):
        ev = torch.cuda.Event()
        ev.record()
        for t in tensors:
            t._ready = ev
        return tensors


# ---- the stream the training managers' device work runs on (rpn_util.RpnTrainingManager, det_util.DetTrainingManager).  ONE per process,
# and one that shares a hardware queue with NEITHER the caller's stream NOR the training step's weight-gradient and prefix streams.
# HIP deals its hardware queues (GPU_MAX_HW_QUEUES, 4 by default) to streams as they are created; once they are taken, streams share.
# A manager stream that shares with the weight-gradient stream has its launches -- and the event the host waits on for the sampling
# counts -- queued behind the previous step's batched weight gradients: 2.7 ms per fp32 RPN iteration instead of 2.05, decided by how
# many streams the process happened to have created before (scripts/dev/stream_probe.py prints the sharing matrix: the same loop leg
# fast or slow by its place in the process).  So candidates are PROBED: a ~3 ms spin on the other stream, an event on the candidate --
# a candidate whose event completes while the spin is still running is independent of it.  More queues are NOT the answer: with
# GPU_MAX_HW_QUEUES = 8 / 12 the same loops ran at 2.5 (mixed) / 3.9 ms (fp32) per iteration.
# FRCNN_MANAGER_STREAM: "own" (default: the probed stream) | "prefix" (train._prefix_stream()) | "main" (the caller's stream).
_MANAGER_STREAM = None
MANAGER_STREAM = os.environ.get("FRCNN_MANAGER_STREAM", "own")


def _runs_beside(busy_stream, stream):
    """Does an event on ``stream`` complete while ``busy_stream`` spins?  (~3 ms per question.)"""
    first = torch.cuda.Event()
    first.record(stream)
    first.synchronize()                                      # (a stream's queue is set up at its first use: not inside the measurement)
    with torch.cuda.stream(busy_stream):
        torch.cuda._sleep(4_000_000)
        busy = torch.cuda.Event()
        busy.record(busy_stream)
    ev = torch.cuda.Event()
    ev.record(stream)
    ev.synchronize()
    beside = not busy.query()
    busy.synchronize()
    return beside


def manager_stream():
    global _MANAGER_STREAM
    if MANAGER_STREAM == "main":
        return torch.cuda.current_stream()
    if MANAGER_STREAM == "prefix":
        from . import train
        return train._prefix_stream()
    if _MANAGER_STREAM is None:
        from . import train
        cur = torch.cuda.current_stream()
        cur.synchronize()
        others = [cur, train._wgrad_stream(), train._prefix_stream()]       # (created now if the process has not trained yet: the order is fixed here)
        tried, best = [], None
        for _ in range(8):
            cand = torch.cuda.Stream()
            tried.append(cand)                               # (kept alive: a released stream's queue slot would be dealt out again)
            free = [_runs_beside(o, cand) for o in others]
            if all(free):
                best = cand
                break
            if best is None and free[0] and free[1]:
                best = cand                                  # sharing with the prefix stream only: 2.10 against 2.05 ms
        if best is None:
            import warnings
            best = tried[-1]
            warnings.warn("faster_rcnn_amd.feed: no HIP stream independent of the training step's streams found among %d candidates; the "
                          "managers' prefetch will queue behind the step's weight gradients (slower loop, same results)" % len(tried))
        _MANAGER_STREAM = best
        # the losing candidates go back (each held a hardware-queue slot for the life of the process); the queues were dealt
        # when the streams were created, so releasing them does not move the chosen one
        manager_stream.tried = len(tried)
        del tried
    return _MANAGER_STREAM
