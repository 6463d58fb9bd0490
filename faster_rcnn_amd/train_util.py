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


def _is_root():
    return dp.rank() == 0


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
        for i in range(num_iterations):
            img = schedule.image(i)
            batched_img = training_manager.batched_image(img)
            y_class, y_bbreg = training_manager.rpn_y_true(img)
            start_time = timeit.default_timer()
            loss_rpn = rpn_model.train_on_batch(batched_img, [y_class, y_bbreg])
            if _is_root():
                print("phase {} iteration {} image {} flipped {}: loss_rpn {} ({:.4f} s)".format(
                    phase_num, i, img.name, img.flipped, loss_rpn, timeit.default_timer() - start_time))
            _save(rpn_model, i, save_frequency, save_weights_dest, save_model_dest, "rpn", allow_zero=True)
    return rpn_model


def _train_detector(detector, images, training_manager, optimizer, phases, save_frequency, save_weights_dest, save_model_dest):
    num_classes = len(training_manager.class_mapping) - 1
    schedule = dp.ImageSchedule(images)
    for phase_num, (num_iterations, learn_rate) in enumerate(phases):
        optimizer.lr = learn_rate
        detector.compile(optimizer=optimizer, loss=[cls_loss_det, bbreg_loss_det(num_classes)])
        print("Starting phase {} of training: {} iterations with learning rate {}".format(phase_num, num_iterations, learn_rate))
        schedule.begin_phase(phase_num, num_iterations)
        for i in range(num_iterations):
            img = schedule.image(i)
            first_input, rois, y_class_num, y_transform = training_manager.get_training_input(img)
            skip = rois is None
            if skip and dp.world() == 1:
                print("Found no rois for this image")
                continue
            start_time = timeit.default_timer()
            loss_frcnn = detector.train_on_batch([first_input, rois], [y_class_num, y_transform], skip=skip)
            if _is_root():
                print("phase {} iteration {} image {} flipped {}: loss_frcnn {} ({:.4f} s)".format(
                    phase_num, i, img.name, img.flipped, loss_frcnn, timeit.default_timer() - start_time))
            _save(detector, i, save_frequency, save_weights_dest, save_model_dest, "detector", allow_zero=False)
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
