"""Host data model the reference's managers consume (shapes.py:5-408), OpenCV-free.

Only what the hot path touches: ``Image`` (metadata view: width/height/gt_boxes/cache_key,
``resize_within_bounds``, ``horizontal_flip``), ``InMemoryImage``, ``Metadata``,
``GroundTruthBox`` and ``Box`` with the reference's arithmetic (float coordinates after a
resize, ``x -> width - x`` flip without -1, shapes.py:298).  Pixel decode (cv2.imread +
INTER_CUBIC, shapes.py:19-29) is an f2 "next" row: the resize restates OpenCV's fixed-point INTER_CUBIC (no cv2 here
to pin it bit for bit; tests hold it within one grey level of an independent a = -0.75 float bicubic); ``Image.data``
decodes with PIL when a file is given, whose JPEG decoder may differ from OpenCV's by a grey level; synthetic configs
use InMemoryImage.
"""
import numpy as np


class Box:
    def __init__(self, x1, y1, x2, y2):
        self.x1, self.y1, self.x2, self.y2 = x1, y1, x2, y2

    @staticmethod
    def from_center_dims_int(x_center, y_center, width, height):
        x1 = x_center - width // 2
        y1 = y_center - height // 2
        return Box(x1, y1, x1 + width, y1 + height)

    @staticmethod
    def from_corners(coords):
        return Box(*coords)

    width = property(lambda s: s.x2 - s.x1)
    height = property(lambda s: s.y2 - s.y1)
    x_center = property(lambda s: (s.x2 + s.x1) / 2)
    y_center = property(lambda s: (s.y1 + s.y2) / 2)
    corners = property(lambda s: np.array([s.x1, s.y1, s.x2, s.y2]))
    corner_dims = property(lambda s: np.array([s.x1, s.y1, s.width, s.height]))
    center_dims = property(lambda s: np.array([s.x_center, s.y_center, s.width, s.height]))

    def resize(self, scale_ratio):
        return Box(self.x1 * scale_ratio, self.y1 * scale_ratio, self.x2 * scale_ratio, self.y2 * scale_ratio)

    def __repr__(self):
        return "<image.Box x1: {}, y1: {}, x2: {}, y2: {}>".format(self.x1, self.y1, self.x2, self.y2)


class GroundTruthBox:
    def __init__(self, obj_cls, difficult, box):
        self.obj_cls, self.difficult, self.box = obj_cls, difficult, box

    def __getattr__(self, name):               # x1, y1, x2, y2, width, height, corners, ... forward to the box
        if name in ("x1", "y1", "x2", "y2", "width", "height", "x_center", "y_center", "corners", "corner_dims", "center_dims"):
            return getattr(self.box, name)
        raise AttributeError(name)

    def resize(self, scale_ratio):
        return GroundTruthBox(self.obj_cls, self.difficult, self.box.resize(scale_ratio))

    def horizontal_flip(self, width):
        return GroundTruthBox(self.obj_cls, self.difficult, Box(width - self.x2, self.y1, width - self.x1, self.y2))

    def __repr__(self):
        return "<shapes.GroundTruthBox obj_cls: {}, difficult: {}, box: {}>".format(self.obj_cls, self.difficult, self.box)


class Metadata:
    def __init__(self, name, width, height, gt_boxes, image_path, flipped=False):
        self.name, self.width, self.height = name, width, height
        self.gt_boxes, self.image_path, self.flipped = gt_boxes, image_path, flipped

    def horizontal_flip(self):
        return Metadata(self.name, self.width, self.height, [b.horizontal_flip(self.width) for b in self.gt_boxes],
                        self.image_path, flipped=not self.flipped)


def _bounds_ratio(width, height, min_size, max_size):
    short_dim, long_dim = min(width, height), max(width, height)
    min_ratio = min_size / short_dim
    return max_size / long_dim if min_ratio * long_dim > max_size else min_ratio


def declares(obj, name):
    """``hasattr`` without RUNNING a property: is ``name`` an attribute of the object's class or of the instance itself?  (hasattr(image,
    "raw") executes Image.raw -- a JPEG decode plus a channel reversal, ~1.7 ms -- just to learn that it exists.)"""
    return hasattr(type(obj), name) or name in getattr(obj, "__dict__", ())


class Image:
    def __init__(self, metadata, pixels=None):
        self.metadata = metadata
        self._pixels = pixels               # optional in-memory BGR uint8 array (synthetic data)

    width = property(lambda s: s.metadata.width)
    height = property(lambda s: s.metadata.height)
    flipped = property(lambda s: s.metadata.flipped)
    gt_boxes = property(lambda s: s.metadata.gt_boxes)
    num_gt_boxes = property(lambda s: len(s.metadata.gt_boxes))
    name = property(lambda s: s.metadata.name)
    cache_key = property(lambda s: s.metadata.name + str(s.metadata.flipped))
    _image_path = property(lambda s: s.metadata.image_path)

    @property
    def raw(self):
        """The decoded frame as cv2.imread delivers it: BGR uint8 at the FILE's size, before the resize and the flip that
        ``data`` applies (shapes.py:24-27).  entry.DetectionEntry uploads this and resizes on the device."""
        if self._pixels is not None:
            return self._pixels
        from PIL import Image as PilImage
        return np.ascontiguousarray(np.asarray(PilImage.open(self._image_path).convert("RGB"))[:, :, ::-1])

    @property
    def raw_rgb(self):
        """The decoded frame in the DECODER's channel order (RGB), or None for in-memory pixels (which are BGR already): ``raw`` without
        the channel reversal, which costs as much host time as half the JPEG decode (entry.DetectionEntry swaps on the device)."""
        if self._pixels is not None:
            return None
        from PIL import Image as PilImage
        with PilImage.open(self._image_path) as im:
            if im.mode != "RGB":
                im = im.convert("RGB")                      # (an RGB JPEG is decoded as it is: convert() would copy the frame once more)
            arr = np.asarray(im)
        self.__dict__["_raw_size"] = (int(arr.shape[0]), int(arr.shape[1]))
        return arr

    def raw_size(self):
        """(height, width) of ``raw`` without decoding the pixels (PIL reads the header only)."""
        if self._pixels is not None:
            return int(self._pixels.shape[0]), int(self._pixels.shape[1])
        size = self.__dict__.get("_raw_size")               # (remembered: get_dets_by_cls asks twice per image, ~0.15 ms of PIL header parsing each)
        if size is None:
            from PIL import Image as PilImage
            with PilImage.open(self._image_path) as im:
                w, h = im.size
            size = self.__dict__["_raw_size"] = (int(h), int(w))
        return size

    @property
    def data(self):
        """BGR uint8 (height, width, 3) at the metadata's size, flipped if requested."""
        img = self.raw
        if img.shape[0] != self.height or img.shape[1] != self.width:
            img = _resize_any(img, self.width, self.height)
        return img[:, ::-1].copy() if self.flipped else img

    def resize(self, scale_ratio):
        w, h = int(round(scale_ratio * self.width)), int(round(scale_ratio * self.height))
        md = Metadata(self.name, w, h, [b.resize(scale_ratio) for b in self.gt_boxes], self._image_path, flipped=self.flipped)
        return Image(md, self._pixels)

    def resize_within_bounds(self, min_size, max_size):
        ratio = _bounds_ratio(self.width, self.height, min_size, max_size)
        return self.resize(ratio), ratio

    def horizontal_flip(self):
        return Image(self.metadata.horizontal_flip(), self._pixels)


class InMemoryImage:
    def __init__(self, data, width, height):
        self._data, self.width, self.height = data, width, height

    @property
    def raw(self):
        return self._data

    @property
    def data(self):
        return _resize_any(self._data, self.width, self.height)

    def resize(self, scale_ratio):
        return InMemoryImage(self._data, int(round(scale_ratio * self.width)), int(round(scale_ratio * self.height)))

    def resize_within_bounds(self, min_size, max_size):
        ratio = _bounds_ratio(self.width, self.height, min_size, max_size)
        return self.resize(ratio), ratio


def _cubic_taps(dst, src):
    """OpenCV resize(INTER_CUBIC) tap table for one axis (imgproc/resize.cpp, 8-bit path): source coordinate
    f = (d + 0.5) * src / dst - 0.5, s = floor(f), four taps s-1..s+2 clamped to the border (BORDER_REPLICATE), cubic
    kernel with A = -0.75 in float32, coefficients in fixed point with 11 fractional bits (cvRound, saturate to int16)."""
    # OpenCV: scale is a double, fx = (float)((dx + 0.5) * scale - 0.5), sx = cvFloor(fx), fx -= sx  (all in float from there)
    f = ((np.arange(dst, dtype=np.float64) + 0.5) * (float(src) / float(dst)) - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    x = f - s.astype(np.float32)
    A = np.float32(-0.75)
    one = np.float32(1.0)
    c0 = ((A * (x + one) - np.float32(5) * A) * (x + one) + np.float32(8) * A) * (x + one) - np.float32(4) * A
    c1 = ((A + np.float32(2)) * x - (A + np.float32(3))) * x * x + one
    c2 = ((A + np.float32(2)) * (one - x) - (A + np.float32(3))) * (one - x) * (one - x) + one
    c3 = one - c0 - c1 - c2
    coef = np.stack([c0, c1, c2, c3], axis=1).astype(np.float32) * np.float32(2048.0)
    icoef = np.clip(np.rint(coef), -32768, 32767).astype(np.int64)            # cvRound = round half to even
    idx = np.clip(s[:, None] + np.arange(-1, 3)[None, :], 0, src - 1)
    return idx, icoef


# Whether ``Image.data`` may resize on the device.  None (default): only when this process has ALREADY initialised the HIP
# runtime (a trainer or a detection entry is running) -- a host-only tool that reads pixels (eval, statistics, a launcher
# that still has to fork its ranks) never starts the runtime as a side effect.  True / False force it either way.
DEVICE_RESIZE = None
_RESIZE_STREAM = None
_RESIZE_WARNED = False


def _device_resize_ok():
    if DEVICE_RESIZE is False:
        return False
    import torch
    if DEVICE_RESIZE is True:
        return torch.cuda.is_available()
    return torch.cuda.is_initialized()


def _resize_any(img, width, height):
    """``_resize`` through the device kernel when the HIP runtime is up (frcnn_resize_cubic_u8: the same integers, ~1 ms with
    both copies instead of ~90 ms of numpy for a VOC frame -- at one image per 2.5 ms training step the host resize would
    otherwise bound the loop), the numpy restatement below otherwise.  The device pass runs on a stream of its own, so the
    copy back waits for THIS resize only, not for the training step queued on the current stream; anything that goes wrong
    on the way (library not built, no memory, a forked worker without a context) falls back to the bit-identical host form."""
    if img.shape[0] == height and img.shape[1] == width:
        return img
    if img.ndim == 3 and img.shape[2] == 3 and img.dtype == np.uint8:
        try:
            if _device_resize_ok():
                import torch
                from . import ops
                global _RESIZE_STREAM
                if _RESIZE_STREAM is None:
                    _RESIZE_STREAM = torch.cuda.Stream()
                with torch.cuda.stream(_RESIZE_STREAM):
                    dev = torch.from_numpy(np.ascontiguousarray(img)).cuda(non_blocking=False)
                    out = ops.resize_cubic_u8(dev, height, width)
                    host = out.cpu()                              # synchronises _RESIZE_STREAM only
                return host.numpy()
        except (ImportError, RuntimeError) as e:                  # library not built / no context in a forked worker / HIP error (FrcnnError is a RuntimeError)
            if DEVICE_RESIZE is True:
                raise                                             # the caller asked for the device path: its failure is the caller's to see
            global _RESIZE_WARNED
            if not _RESIZE_WARNED:
                _RESIZE_WARNED = True
                import warnings
                warnings.warn("faster_rcnn_amd.shapes: device resize failed (%s: %s); falling back to the host restatement (~90 ms per frame, same pixels)"
                              % (type(e).__name__, e))
    return _resize(img, width, height)


def _resize(img, width, height):
    """cv2.resize(img, (width, height), interpolation=cv2.INTER_CUBIC) for uint8 images, restated in integer arithmetic
    (shapes.py:24 of the reference resizes every image this way).  OpenCV is not installable here, so this follows the
    published algorithm (separable, horizontal then vertical, 22-bit fixed point, FixedPtCast rounding) and is NOT
    pinned against cv2 output; OpenCV's SIMD vertical pass rounds through float and may differ by one grey level."""
    if img.shape[0] == height and img.shape[1] == width:
        return img
    src = np.ascontiguousarray(img)
    squeeze = src.ndim == 2
    if squeeze:
        src = src[:, :, None]
    xi, xa = _cubic_taps(width, src.shape[1])
    yi, yb = _cubic_taps(height, src.shape[0])
    rows = src.astype(np.int64)
    horiz = np.zeros((src.shape[0], width, src.shape[2]), dtype=np.int64)
    for k in range(4):
        horiz += rows[:, xi[:, k], :] * xa[None, :, k, None]
    out = np.zeros((height, width, src.shape[2]), dtype=np.int64)
    for k in range(4):
        out += horiz[yi[:, k], :, :] * yb[:, k, None, None]
    out = np.clip((out + (1 << 21)) >> 22, 0, 255).astype(np.uint8)
    return out[:, :, 0] if squeeze else out
