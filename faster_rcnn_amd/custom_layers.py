"""Mirror of the reference's custom Keras layers (custom_layers.py:7-134) on the HIP library."""
import numpy as np
import torch

from . import ops


class RoiResizeConv:
    """custom_layers.RoiResizeConv (custom_layers.py:7-56): crops each region out of the feature
    map and bilinearly resizes it to pool_size x pool_size (TF 1.3 resize_images semantics).
    Call with ``[feature_map (1,R,C,Cf), rois (1,num_rois,4)]`` -> (1,num_rois,pool,pool,Cf).
    One HIP launch for all regions (the reference emits num_rois slice+resize graph nodes)."""

    def __init__(self, pool_size, num_rois, **kwargs):
        self.pool_size, self.num_rois = pool_size, num_rois
        self.nb_channels = None

    def build(self, input_shape):
        self.nb_channels = input_shape[0][3]

    def compute_output_shape(self, input_shape):
        return None, self.num_rois, self.pool_size, self.pool_size, self.nb_channels

    def get_config(self):
        return {"pool_size": self.pool_size, "num_rois": self.num_rois}

    def call(self, x):
        img, rois = x
        as_np = not isinstance(img, torch.Tensor)
        img_d = torch.from_numpy(np.ascontiguousarray(img, dtype=np.float32)).cuda() if as_np else img
        rois_d = torch.from_numpy(np.ascontiguousarray(rois, dtype=np.float32)).cuda() if not isinstance(rois, torch.Tensor) else rois
        self.nb_channels = img_d.shape[-1]
        rois_d = rois_d.reshape(-1, 4)[:self.num_rois]
        out = ops.roi_crop_resize(img_d, rois_d, self.pool_size)[None]
        return out.cpu().numpy() if as_np else out

    __call__ = call


class Scale:
    """custom_layers.Scale (custom_layers.py:59-134): out = gamma * x + beta along ``axis``.
    In the lowered graphs it never runs on its own: nets.ConvUnit folds it into the producing
    convolution's epilogue.  This standalone form exists for API parity."""

    def __init__(self, weights=None, axis=-1, momentum=0.9, beta_init="zero", gamma_init="one", **kwargs):
        self.axis, self.momentum = axis, momentum
        self.gamma = self.beta = None
        self.initial_weights = weights

    def build(self, input_shape):
        c = int(input_shape[self.axis])
        self.gamma, self.beta = np.ones(c, np.float32), np.zeros(c, np.float32)
        if self.initial_weights is not None:
            self.set_weights(self.initial_weights)

    def set_weights(self, w):
        self.gamma, self.beta = (np.asarray(a, dtype=np.float32) for a in w)

    def get_weights(self):
        return [self.gamma, self.beta]

    def call(self, x, mask=None):
        if self.gamma is None:
            self.build(x.shape)
        shape = [1] * x.ndim
        shape[self.axis] = -1
        return self.gamma.reshape(shape) * np.asarray(x) + self.beta.reshape(shape)

    __call__ = call

    def get_config(self):
        return {"momentum": self.momentum, "axis": self.axis}
