"""Mirror of the reference's resnet.py builder API on the HIP engine (resnet.py:22-93, 395-686).

Same function names, argument meaning and returned duck type; the graphs are lowered by
nets.py instead of being built from Keras layers.  The reference downloads ImageNet weights
inside the builders (resnet.py:481-485, 542-546); with no network the builders take a
``weights=`` dict (weights.py) and default to the seeded synthetic set.
"""
import numpy as np

from . import nets
from .models import BaseModel, DetModel, RpnModel
from .shared_constants import DEFAULT_ANCHORS_PER_LOC
from .weights import load_npz, synthetic_resnet

POOLING_REGIONS = 7
FINAL_CONV_FILTERS = 1024
STRIDE = 16


class _L2:
    def __init__(self, l2):
        self.l2 = l2


WEIGHT_REGULARIZER = _L2(1e-4)      # resnet.py:26-29
BIAS_REGULARIZER = _L2(1e-4)
ACTIVITY_REGULARIZER = _L2(1e-4)

_MEAN_BGR = np.array([103.939, 116.779, 123.68])


def preprocess(data):
    """resnet.preprocess (resnet.py:64-75): BGR->RGB, then Keras resnet50.preprocess_input which
    flips back to BGR and subtracts the ImageNet means => BGR - mean, float64."""
    return np.asarray(data).astype("float64") - _MEAN_BGR


def get_conv_rows_cols(height, width):
    """resnet.py:78-93."""
    dims = [height, width]
    for i in range(2):
        dims[i] += 6
        for filter_size in (7, 3, 1, 1):
            dims[i] = (dims[i] - filter_size) // 2 + 1
    return dims


def _base(depth, freeze_blocks, weight_regularizer, bias_regularizer, weights, dtype="f32"):
    weights = weights if weights is not None else synthetic_resnet(depth)
    return BaseModel(weights, nets.ResNetBase(weights, depth, dtype), "resnet%d" % depth, freeze_blocks, weight_regularizer, bias_regularizer)


def resnet50_base(freeze_blocks=[1, 2, 3], weight_regularizer=None, bias_regularizer=None, weights=None, dtype="f32"):
    """dtype="bf16" selects the bf16 conv path (BASELINE configs[3]); the reference has no such knob."""
    return _base(50, freeze_blocks, weight_regularizer, bias_regularizer, weights, dtype)


def resnet101_base(freeze_blocks=[1, 2, 3], weight_regularizer=None, bias_regularizer=None, weights=None, dtype="f32"):
    return _base(101, freeze_blocks, weight_regularizer, bias_regularizer, weights, dtype)


def resnet50_rpn(base_model, weight_regularizer=None, bias_regularizer=None, include_conv=False,
                 anchors_per_loc=DEFAULT_ANCHORS_PER_LOC):
    assert base_model.weights["rpn_out_cls"][0].shape[-1] == anchors_per_loc, "weights were drawn for a different anchor count"
    m = RpnModel(base_model, include_conv, anchors_per_loc)
    m.weight_regularizer = weight_regularizer            # the heads' own regulariser (step 3 regularises only them)
    return m


resnet101_rpn = resnet50_rpn


def _classifier(depth, num_rois, num_classes, base_model, weights, dtype="f32"):
    if base_model is not None:
        weights = base_model.weights
        dtype = getattr(base_model.net, "dtype", dtype)
    elif weights is None:
        weights = synthetic_resnet(depth, num_classes=num_classes)
    assert "dense_class_%d" % num_classes in weights, "weights were drawn for a different class count"
    return DetModel(weights, nets.ResNetHead(weights, depth, num_classes, dtype), num_rois, num_classes, base_model)


def resnet50_classifier(num_rois, num_classes, base_model=None, weight_regularizer=None, bias_regularizer=None, weights=None, dtype="f32"):
    m = _classifier(50, num_rois, num_classes, base_model, weights, dtype)
    m.weight_regularizer = weight_regularizer
    return m


def resnet101_classifier(num_rois, num_classes, base_model=None, weight_regularizer=None, bias_regularizer=None, weights=None, dtype="f32"):
    m = _classifier(101, num_rois, num_classes, base_model, weights, dtype)
    m.weight_regularizer = weight_regularizer
    return m


def rpn_from_h5(h5_path, anchors_per_loc=DEFAULT_ANCHORS_PER_LOC, depth=50):
    """resnet.rpn_from_h5 (resnet.py:32-44): a Keras 2.0.x ``.h5`` (read in-process, h5lite.py) or the
    ``.npz`` this package's ``save_weights`` writes.  Step-4 RPN models carry the conv4 map as third
    output (train_det_step4.py:80)."""
    w = load_npz(h5_path)
    base = _base(depth, [1, 2, 3], None, None, w)
    return RpnModel(base, True, anchors_per_loc)


def det_from_h5(h5_path, num_classes, depth=50):
    w = load_npz(h5_path)
    return DetModel(w, nets.ResNetHead(w, depth, num_classes), 64, num_classes, None)
