"""The Keras-model duck type the reference's callers rely on:
``predict_on_batch`` (det_util.py:41), ``predict`` (voc_dets.py:49), ``output`` (length test,
det_util.py:27), ``get_layer(name).get_weights()`` (train_rpn_test.py:41), ``load_weights`` /
``save_weights``.  numpy in / numpy out at this surface; ``forward_dev`` is the
device-resident form the fused pipeline uses (no host round trip)."""
import numpy as np
import torch

from . import nets
from .weights import load_npz, save_weights_file


_WEIGHTS_EPOCH = 0


def weights_epoch():
    """Bumped whenever ANY model re-lowers a layer (load_weights, a flushed training step, set_weights).  Captured passes
    hold pointers to packed filters, so whoever keeps hipGraphs across calls (entry.DetectionEntry) compares epochs and
    re-captures; one process-wide counter because models share base networks and weight dicts."""
    return _WEIGHTS_EPOCH


class _Layer:
    def __init__(self, weights, name, model=None):
        self._w, self.name, self._model = weights, name, model

    def get_weights(self):
        return [np.array(a) for a in self._w[self.name]]

    def set_weights(self, arrs):
        self._w[self.name] = [np.asarray(a, dtype=np.float32) for a in arrs]
        if self._model is not None:
            self._model.invalidate(only={self.name})


class _Model:
    """Trained weights live in the trainer's flat fp32 master buffer (train.ParamSet) between steps; every way OUT of
    the model -- ``get_layer(...).get_weights()`` right after ``train_on_batch`` (train_rpn_test.py:41), ``predict*``,
    ``save*`` -- first flushes them back into the Keras-keyed weight dict (``_flush_trainer``), and ``load_weights``
    after ``compile`` rebuilds the trainer on the loaded values (Keras keeps training from what was loaded)."""
    _trainer = None
    _dirty = False

    def __init__(self, weights):
        self.weights = weights

    def _modules(self):
        return []

    def _flush_trainer(self):
        if self._trainer is not None and self._dirty:
            self._trainer.sync_weights()
            self._dirty = False

    def _new_trainer(self):
        raise NotImplementedError

    def compile(self, optimizer, loss=None):
        """Keras ``compile``: a fresh train function, i.e. fresh optimiser slots (train_util.py:31-33, 95)."""
        if self._trainer is None:
            self._trainer = self._new_trainer()
        self._trainer.compile(optimizer, loss)

    supports_deferred_losses = True

    def train_on_batch(self, x, y, skip=False, defer=False):
        """Keras' train_on_batch: [total, loss 1, loss 2].  ``defer=True`` (not in Keras) only ENQUEUES the step and
        returns a train.PendingLosses; its ``result()`` is that list.  The training loops use it to prepare the next
        image while the GPU runs this step."""
        self._dirty = True
        return self._trainer.train_on_batch(x, y, skip=skip, defer=defer)

    def get_layer(self, name):
        self._flush_trainer()
        if name not in self.weights:
            raise ValueError("No such layer: " + name)
        return _Layer(self.weights, name, self)

    def invalidate(self, only=None):
        """Re-lower (re-pack / re-fold) after weights changed; ``only``: the layer names that changed (a conv layer or the
        BatchNormalization / Scale folded into its launch)."""
        global _WEIGHTS_EPOCH
        _WEIGHTS_EPOCH += 1
        for m in self._modules():
            for u in m.units():
                if only is None or u.conv in only or u.bn in only or u.scale_name in only or any(n in only for n in getattr(u, "names", ())):
                    u.pc = None
            if hasattr(m, "invalidate_fused"):
                m.invalidate_fused(only)

    def load_weights(self, path, by_name=False):
        """Keras ``load_weights``: ``path`` is a Keras 2.0.x .h5 (weights or full model) or this package's .npz."""
        self._flush_trainer()                               # layers the file does not name keep their TRAINED values
        new = load_npz(path)
        for k, v in new.items():
            if k in self.weights or not by_name:
                self.weights[k] = v
        self.invalidate()
        if self._trainer is not None:
            # training continues from the loaded values: new master buffers and packed filters; the optimiser object
            # (with its step counter) and its slots carry over, as in Keras, where load_weights touches neither
            old = self._trainer
            old.drop_step_graphs()                          # (captured steps hold the old packed filters: destroy them here, not in a finalizer)
            self._trainer = self._new_trainer()
            if old.optimizer is not None:
                self._trainer.compile(old.optimizer)
                self._trainer.params.adopt_slots(old.params)

    def save_weights(self, path):
        """Keras ``save_weights``: ``.h5`` writes a Keras 2.0.x HDF5 weight file, any other suffix the .npz form."""
        self._flush_trainer()
        save_weights_file(path, self.weights)

    def save(self, path):
        """Keras ``model.save``: for ``.h5`` the weights go under ``model_weights`` like Keras' full-model files
        (architecture / optimiser state are not stored: the builders re-create the graph from the layer names)."""
        self._flush_trainer()
        save_weights_file(path, self.weights, full_model=True)


class BaseModel(_Model):
    def __init__(self, weights, net, kind, freeze_blocks, weight_regularizer=None, bias_regularizer=None):
        super().__init__(weights)
        self.net, self.kind, self.freeze_blocks = net, kind, list(freeze_blocks)
        self.weight_regularizer, self.bias_regularizer = weight_regularizer, bias_regularizer

    def _modules(self):
        return [self.net]

    def forward_dev(self, x):
        return self.net(x)


class RpnModel(_Model):
    """base + RPN heads.  ``output`` has 3 entries when the conv4 map is also returned
    (include_conv=True), which DetTrainingManager tests with len() (det_util.py:27)."""

    def __init__(self, base, include_conv, anchors_per_loc):
        super().__init__(base.weights)
        self.base, self.include_conv, self.anchors_per_loc = base, include_conv, anchors_per_loc
        self.head = nets.RpnHead(base.weights, getattr(base.net, "dtype", "f32"))
        self.output = ["rpn_out_cls", "rpn_out_bbreg"] + (["conv_out"] if include_conv else [])

    def _modules(self):
        return [self.base.net, self.head]

    def forward_dev(self, x, extents=None):
        """``extents`` (nets.Extents): x holds canvases (ResNet bases; see nets.ResNetBase.__call__)."""
        feat = self.base.net(x) if extents is None else self.base.net(x, extents)
        cls, reg = self.head(feat)
        return cls, reg, feat

    # ---- training (train_util.py:31-54): compile / train_on_batch / flushing live in _Model
    def _new_trainer(self):
        from .train import RpnTrainer
        reg = getattr(self, "weight_regularizer", None) or self.base.weight_regularizer
        return RpnTrainer(self, l2=reg.l2 if reg is not None else 0.0)

    def predict_on_batch(self, x):
        self._flush_trainer()
        cls, reg, feat = self.forward_dev(nets.to_device_image(x))
        outs = [cls.cpu().numpy(), reg.cpu().numpy()]
        if self.include_conv:
            outs.append(feat.float().cpu().numpy())
        return outs


class DetModel(_Model):
    """[image or conv4 map, rois] -> [class probabilities (1,n,C), box regressions (1,n,4(C-1))]."""

    def __init__(self, weights, head, num_rois, num_classes, base=None):
        super().__init__(weights)
        self.head, self.num_rois, self.num_classes, self.base = head, num_rois, num_classes, base
        self.output = ["dense_class_%d" % num_classes, "dense_reg_%d" % num_classes]

    def _modules(self):
        return ([self.base.net] if self.base is not None else []) + [self.head]

    def forward_dev(self, first, rois):
        feat = self.base.net(first) if self.base is not None else first
        if getattr(self.head, "dtype", "f32") == "bf16" and feat.dtype == torch.float32:
            from . import ops
            feat = ops.cast_bf16(feat)                      # a conv map that travelled as float32 numpy (det_util.get_det_inputs): exact narrowing
        return self.head(feat, rois)

    # ---- training (train_util.py:95-118): compile / train_on_batch / flushing live in _Model
    def _new_trainer(self):
        from .train import DetTrainer
        reg = self.base.weight_regularizer if self.base is not None else getattr(self, "weight_regularizer", None)
        return DetTrainer(self, l2=reg.l2 if reg is not None else 0.0)

    def predict(self, inputs):
        self._flush_trainer()
        first, rois = inputs
        first = nets.to_device_image(first)
        rois = torch.from_numpy(np.ascontiguousarray(np.asarray(rois, dtype=np.float32))).cuda().reshape(-1, 4)
        cls, reg = self.forward_dev(first, rois)
        return [cls.cpu().numpy()[None], reg.cpu().numpy()[None]]

    predict_on_batch = predict
