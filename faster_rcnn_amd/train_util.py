"""Mirror of the reference's training loops (train_util.py:10-193) with data parallelism.

Same signatures and per-iteration behaviour: per-phase ``compile`` with the phase's learning rate
(optimiser slots reset, train_util.py:31-33), the image schedule ``(i + num_iterations*phase) %
num_train`` with a shuffle whenever the index wraps to 0 (:39-41), batch = 1 image per step per
GPU, periodic saves (RPN loop also at i == 0, :58; detector loops only for i > 0, :122).
Under ``torch.distributed`` each global step consumes world_size consecutive images of that
schedule (dp.image_index), every rank runs its own image and the flat gradient buffer is
all-reduced once.  A rank whose image yields no RoIs still joins the exchange with zero gradients
(the reference just ``continue``s, :112-114).  The image walk and its shuffles live in
``dp.ImageSchedule``: a dedicated, identically seeded shuffle stream on every rank when world > 1 (the
global ``random`` stream keeps serving the per-image sampling), the reference's own ``random.shuffle``
when world == 1.
"""
import timeit

from . import dp
from .loss_functions import bbreg_loss_det, bbreg_loss_rpn, cls_loss_det, cls_loss_rpn
from .shared_constants import DEFAULT_LEARN_RATE, DEFAULT_NUM_ITERATIONS


# The loops below feed the steps through the managers' device-resident fast paths (rpn_util.RpnTrainingManager.rpn_inputs_dev,
# det_util.DetTrainingManager.get_training_input_dev) whenever model and manager are this package's: the same float32 inputs, built on
# the device, with the next image's RNG-free part prepared beside the current step.  FAST_FEED = False (or a foreign model / manager)
# takes the reference's calls literally: batched_image / rpn_y_true / get_training_input returning host numpy.
import os as _os
FAST_FEED = _os.environ.get("FRCNN_TRAIN_FAST_FEED", "1") != "0"


def _is_root():
    return dp.rank() == 0


def _fast(model, manager, method):
    return FAST_FEED and getattr(model, "supports_deferred_losses", False) and hasattr(manager, method) and hasattr(manager, "prefetch")


class _LossLog:
    """The reference prints each iteration's losses right after train_on_batch (train_util.py:54-57, 118-121), which
    makes the host wait for the step.  Here the step is only enqueued (``defer=True``) and its line is printed once the
    NEXT iteration's step has been enqueued -- the host decodes / samples the next image while the GPU trains on this
    one -- or earlier when something else is about to print.  Same lines, same order."""

    def __init__(self):
        self._pending = None

    def flush(self):
        if self._pending is not None:
            fmt, args, losses, t0 = self._pending
            self._pending = None
            value = losses.result() if hasattr(losses, "result") else losses
            if _is_root():
                print(fmt.format(*args, value, timeit.default_timer() - t0))

    def push(self, fmt, args, losses, t0):
        self.flush()
        self._pending = (fmt, args, losses, t0)


def _decode_ahead(manager, schedule, i):
    """The image of iteration ``i`` (two ahead of the step just enqueued), when naming it needs no shuffle: its JPEG decode starts on a
    background thread now, so that the prefetch one iteration from here finds the pixels (feed.decode_ahead)."""
    if hasattr(manager, "decode_ahead"):
        later = schedule.peek(i)
        if later is not None:
            manager.decode_ahead(later)


def _enqueue_step(model, x, y, **kw):
    """train_on_batch, deferred when the model offers it (models._Model does; a plain Keras-style model returns its list)."""
    if getattr(model, "supports_deferred_losses", False):
        return model.train_on_batch(x, y, defer=True, **kw)
    return model.train_on_batch(x, y, **kw)


def _save(model, i, save_frequency, save_weights_dest, save_model_dest, what, allow_zero):
    if save_frequency and (allow_zero or i > 0) and i % save_frequency == 0 and _is_root():
        if save_weights_dest is not None:
            model.save_weights(save_weights_dest)
            print("Saved {} weights to {}".format(what, save_weights_dest))
        if save_model_dest is not None:
            model.save(save_model_dest)
            print("Saved {} model to {}".format(what, save_model_dest))


def train_rpn(rpn_model, images, training_manager, optimizer, phases=[[DEFAULT_NUM_ITERATIONS, DEFAULT_LEARN_RATE]],
              save_frequency=None, save_weights_dest=None, save_model_dest=None):
    """train_util.train_rpn (train_util.py:10-66)."""
    anchors_per_loc = len(training_manager.anchor_dims)
    schedule = dp.ImageSchedule(images)
    for phase_num, (num_iterations, learn_rate) in enumerate(phases):
        optimizer.lr = learn_rate
        rpn_model.compile(optimizer=optimizer, loss=[cls_loss_rpn(anchors_per_loc=anchors_per_loc),
                                                     bbreg_loss_rpn(anchors_per_loc=anchors_per_loc)])
        print("Starting phase {} of training: {} iterations with learning rate {}".format(phase_num, num_iterations, learn_rate))
        schedule.begin_phase(phase_num, num_iterations)
        log = _LossLog()
        fast = _fast(rpn_model, training_manager, "rpn_inputs_dev")
        for i in range(num_iterations):
            img = schedule.image(i)
            if fast:
                batched_img, y_class, y_bbreg = training_manager.rpn_inputs_dev(img)
            else:
                batched_img = training_manager.batched_image(img)
                y_class, y_bbreg = training_manager.rpn_y_true(img)
            start_time = timeit.default_timer()
            loss_rpn = _enqueue_step(rpn_model, batched_img, [y_class, y_bbreg])
            ahead = schedule.peek(i + 1) if fast else None   # (None when fetching it would shuffle: the shuffle stays where the reference has it)
            if ahead is not None:
                training_manager.prefetch(ahead)             # upload, resize, preprocess, anchor assignment of the NEXT image beside this step
                _decode_ahead(training_manager, schedule, i + 2)
            log.push("phase {} iteration {} image {} flipped {}: loss_rpn {} ({:.4f} s)", (phase_num, i, img.name, img.flipped), loss_rpn, start_time)
            if save_frequency and i % save_frequency == 0:
                log.flush()
            _save(rpn_model, i, save_frequency, save_weights_dest, save_model_dest, "rpn", allow_zero=True)
        log.flush()
    return rpn_model


def _train_detector(detector, images, training_manager, optimizer, phases, save_frequency, save_weights_dest, save_model_dest):
    num_classes = len(training_manager.class_mapping) - 1
    schedule = dp.ImageSchedule(images)
    for phase_num, (num_iterations, learn_rate) in enumerate(phases):
        optimizer.lr = learn_rate
        detector.compile(optimizer=optimizer, loss=[cls_loss_det, bbreg_loss_det(num_classes)])
        print("Starting phase {} of training: {} iterations with learning rate {}".format(phase_num, num_iterations, learn_rate))
        schedule.begin_phase(phase_num, num_iterations)
        log = _LossLog()
        fast = _fast(detector, training_manager, "get_training_input_dev")
        for i in range(num_iterations):
            img = schedule.image(i)
            first_input, rois, y_class_num, y_transform = (training_manager.get_training_input_dev if fast else training_manager.get_training_input)(img)
            skip = rois is None
            if skip and dp.world() == 1:
                log.flush()
                print("Found no rois for this image")
                continue
            start_time = timeit.default_timer()
            loss_frcnn = _enqueue_step(detector, [first_input, rois], [y_class_num, y_transform], skip=skip)
            ahead = schedule.peek(i + 1) if fast else None
            if ahead is not None:
                training_manager.prefetch(ahead)             # the next image's RPN pass, proposals and RoI -> truth beside this step
                _decode_ahead(training_manager, schedule, i + 2)
            log.push("phase {} iteration {} image {} flipped {}: loss_frcnn {} ({:.4f} s)", (phase_num, i, img.name, img.flipped), loss_frcnn, start_time)
            if save_frequency and i % save_frequency == 0:
                log.flush()
            _save(detector, i, save_frequency, save_weights_dest, save_model_dest, "detector", allow_zero=False)
        log.flush()
    return detector


def train_detector_step2(detector, images, training_manager, optimizer, phases=[[DEFAULT_NUM_ITERATIONS, DEFAULT_LEARN_RATE]],
                         save_frequency=None, save_weights_dest=None, save_model_dest=None):
    """train_util.train_detector_step2 (train_util.py:69-130): detector fed with images."""
    return _train_detector(detector, images, training_manager, optimizer, phases, save_frequency, save_weights_dest, save_model_dest)


def train_detector_step4(detector, images, training_manager, optimizer, phases=[[DEFAULT_NUM_ITERATIONS, DEFAULT_LEARN_RATE]],
                         save_frequency=None, save_weights_dest=None, save_model_dest=None):
    """train_util.train_detector_step4 (train_util.py:133-193): same loop; the manager (conv_only: its RPN
    model has three outputs, det_util.py:27) hands the cached conv features instead of the image and the
    base-less detector trains its head only."""
    return _train_detector(detector, images, training_manager, optimizer, phases, save_frequency, save_weights_dest, save_model_dest)
