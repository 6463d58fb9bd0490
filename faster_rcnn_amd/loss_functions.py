"""Mirror of the reference's loss_functions.py (loss_functions.py:8-76).

The factories keep their names and arguments; what they return is a small descriptor that
``model.compile(loss=[...])`` accepts.  The arithmetic (value and gradient, including the
reference's quirks -- constant N_CLS/N_REG, mask outside the sum in bbreg_loss_rpn, the 1e-4 in
the bbreg_loss_det denominator) lives in train.hip: frcnn_loss_rpn_cls/_rpn_reg/_det_cls/_det_reg.
"""
from .shared_constants import DEFAULT_ANCHORS_PER_LOC

N_CLS = 256
N_REG = 2400
LAMBDA_REG = 10.0
LAMBDA_REG_DET = 1


class _Loss:
    def __init__(self, kind, **kw):
        self.kind = kind
        self.__dict__.update(kw)
        self.__name__ = kind

    def __repr__(self):
        return "<loss %s>" % self.kind


def cls_loss_rpn(anchors_per_loc=DEFAULT_ANCHORS_PER_LOC):
    return _Loss("cls_loss_rpn", anchors_per_loc=anchors_per_loc)


def bbreg_loss_rpn(anchors_per_loc=DEFAULT_ANCHORS_PER_LOC):
    return _Loss("bbreg_loss_rpn", anchors_per_loc=anchors_per_loc)


def bbreg_loss_det(num_classes):
    return _Loss("bbreg_loss_det", num_classes=num_classes)


cls_loss_det = _Loss("cls_loss_det")
