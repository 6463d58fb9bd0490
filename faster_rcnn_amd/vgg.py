"""Mirror of the reference's vgg.py builder API (vgg.py:19-61, 91-255) on the HIP engine."""
import numpy as np

from . import nets
from .models import BaseModel, DetModel, RpnModel
from .shared_constants import DEFAULT_ANCHORS_PER_LOC
from .weights import load_npz, synthetic_vgg16

POOLING_REGIONS = 7
FINAL_CONV_FILTERS = 512
STRIDE = 16
WEIGHT_REGULARIZER = None       # vgg.py:22-23
BIAS_REGULARIZER = None

_MEAN_BGR = np.array([103.939, 116.779, 123.68])


def preprocess(data):
    """vgg.preprocess (vgg.py:52-57): arithmetically identical to resnet.preprocess."""
    return np.asarray(data).astype("float64") - _MEAN_BGR


def get_conv_rows_cols(height, width):
    return height // STRIDE, width // STRIDE


def vgg16_base(freeze_blocks=[1, 2], weight_regularizer=None, bias_regularizer=None, weights=None):
    weights = weights if weights is not None else synthetic_vgg16()
    return BaseModel(weights, nets.VggBase(weights), "vgg16", freeze_blocks, weight_regularizer, bias_regularizer)


def vgg16_rpn(base_model, include_conv=False, weight_regularizer=None, bias_regularizer=None,
              anchors_per_loc=DEFAULT_ANCHORS_PER_LOC):
    assert base_model.weights["rpn_out_cls"][0].shape[-1] == anchors_per_loc
    m = RpnModel(base_model, include_conv, anchors_per_loc)
    m.weight_regularizer = weight_regularizer
    return m


def vgg16_classifier(num_rois, num_classes, base_model=None, weight_regularizer=None, bias_regularizer=None, weights=None):
    if base_model is not None:
        weights = base_model.weights
    elif weights is None:
        weights = synthetic_vgg16(num_classes=num_classes)
    m = DetModel(weights, nets.VggHead(weights, num_classes), num_rois, num_classes, base_model)
    m.weight_regularizer = weight_regularizer
    return m


def rpn_from_h5(h5_path, anchors_per_loc=DEFAULT_ANCHORS_PER_LOC):
    w = load_npz(h5_path)
    return RpnModel(vgg16_base(weights=w), True, anchors_per_loc)


def det_from_h5(h5_path, num_classes):
    w = load_npz(h5_path)
    return DetModel(w, nets.VggHead(w, num_classes), 64, num_classes, None)
