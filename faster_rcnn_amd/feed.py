"""Device-side feed of the training managers' fast paths (rpn_util.RpnTrainingManager, det_util.DetTrainingManager).

The reference hands ``train_on_batch`` host arrays: ``np.expand_dims(preprocess_func(image.data), 0)`` -- a float64 (1,H,W,3)
array, 14 MB for a 600x1000 frame (rpn_util.py:50-51, det_util.py:35-36) -- and the step's feed casts it to float32 and uploads
it.  When the manager's ``preprocess_func`` is one of this package's mean subtractions (resnet.preprocess / vgg.preprocess) the
same float32 tensor is produced ON the device from the decoded uint8 frame: one 0.6-1.8 MB upload, the INTER_CUBIC resize and
flip of ``shapes.Image.data`` (frcnn_resize_cubic_u8, the same integers as the host restatement) and frcnn_preprocess_u8 (bit for
bit float32(float64(pixel) - mean)).  tests/test_train_loop_gpu.py holds the two feeds to the same bits.
"""
import numpy as np
import torch

from . import ops

MEAN_BGR = (103.939, 116.779, 123.68)           # resnet.preprocess / vgg.preprocess (resnet.py:64-75, vgg.py:52-57)


def device_preprocess(preprocess_func):
    from . import resnet, vgg
    return preprocess_func in (resnet.preprocess, vgg.preprocess)


def device_image(image, preprocess_func):
    """(1,H,W,3) float32 device tensor == float32(np.expand_dims(preprocess_func(image.data), 0)), on the current stream."""
    if device_preprocess(preprocess_func) and hasattr(image, "raw") and hasattr(image, "height"):
        raw = np.asarray(image.raw)
        if raw.dtype == np.uint8 and raw.ndim == 3 and raw.shape[2] == 3:
            H, W, flip = int(image.height), int(image.width), bool(getattr(image, "flipped", False))
            u8 = torch.from_numpy(np.ascontiguousarray(raw)).cuda()
            if u8.shape[0] != H or u8.shape[1] != W:
                u8 = ops.resize_cubic_u8(u8, H, W, flip=flip)
            elif flip:
                u8 = torch.from_numpy(np.ascontiguousarray(raw[:, ::-1])).cuda()
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
