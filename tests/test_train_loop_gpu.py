"""The training loops' device-resident feed (train_util.FAST_FEED; rpn_util.RpnTrainingManager.rpn_inputs_dev,
det_util.DetTrainingManager.get_training_input_dev) against the reference's own calls taken literally (batched_image / rpn_y_true /
get_training_input returning host numpy, train_util.py:37-54, 100-118): the SAME float32 step inputs bit for bit, the same host
random streams afterwards, and -- through train_util's loops over several images, phases and a wrap of the image list -- the same
weights bit for bit."""
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def frames(n, h, w, seed=0, boxes=5, src=None):
    """In-memory images (shapes.Image over decoded uint8 BGR pixels).  ``src``: (h, w) of the stored pixels when they differ from the
    metadata's size (the loaders' resize_within_bounds: Image.data then resizes with INTER_CUBIC)."""
    from faster_rcnn_amd import shapes
    rs = np.random.RandomState(seed)
    names = ["dog", "cat", "person", "chair", "car"]
    out = []
    for k in range(n):
        ph, pw = src or (h, w)
        px = rs.randint(0, 256, (ph, pw, 3)).astype(np.uint8)
        gts = []
        for _ in range(boxes):
            bw, bh = rs.choice([64, 96, 128, 180, 256]), rs.choice([64, 96, 128, 180, 256])
            x1, y1 = rs.randint(0, max(1, w - bw - 1)), rs.randint(0, max(1, h - bh - 1))
            gts.append(shapes.GroundTruthBox(names[rs.randint(5)], False, shapes.Box(int(x1), int(y1), int(min(w - 1, x1 + bw)), int(min(h - 1, y1 + bh)))))
        out.append(shapes.Image(shapes.Metadata("f%02d" % k, w, h, gts, "none"), px))
    return out


def file_frames(resized, flipped):
    """The committed VOC frame, file-backed: the device feed uploads it in the JPEG decoder's channel order (feed.RGB_UPLOAD) and the
    resize kernel writes B, G, R."""
    import os
    from faster_rcnn_amd import util
    from faster_rcnn_amd.data.voc_data_helpers import extract_img_data
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "VOC_test")
    img = extract_img_data(root, "000005")
    if resized:
        (img,), _ = util.resize_imgs([img], min_size=600, max_size=1000)
    return [img.horizontal_flip() if flipped else img]


@pytest.mark.parametrize("case", ["plain", "resized", "flipped", "many_positives", "file", "file_flipped", "file_resized", "file_resized_flipped"])
def test_rpn_inputs_dev_equal_the_host_feed(case):
    """x, y_class, y_bbreg from rpn_inputs_dev == float32(batched_image), float32(rpn_y_true) -- and the global `random` stream ends
    in the same state (the same two draws in the same order)."""
    from faster_rcnn_amd import resnet, rpn_util, util
    anchors = util.get_anchors([128, 256, 512])
    if case.startswith("file"):
        imgs = file_frames("resized" in case, "flipped" in case)
    elif case == "resized":
        imgs = frames(2, 320, 448, seed=3, src=(200, 280))
    elif case == "many_positives":
        imgs = frames(2, 480, 640, seed=4, boxes=260)            # > 128 usable positives (every box makes its best anchor one): the first draw happens too
    else:
        imgs = frames(2, 320, 448, seed=5)
    if case == "flipped":
        imgs = [im.horizontal_flip() for im in imgs]
    for im in imgs:
        host_mgr = rpn_util.RpnTrainingManager(resnet.get_conv_rows_cols, 16, resnet.preprocess, anchors)
        dev_mgr = rpn_util.RpnTrainingManager(resnet.get_conv_rows_cols, 16, resnet.preprocess, anchors)
        random.seed(21)
        x_h = host_mgr.batched_image(im).astype(np.float32)
        yc_h, yb_h = host_mgr.rpn_y_true(im)
        st_h = random.getstate()
        random.seed(21)
        dev_mgr.prefetch(im)
        x_d, yc_d, yb_d = dev_mgr.rpn_inputs_dev(im)
        st_d = random.getstate()
        torch.cuda.synchronize()
        assert st_h == st_d
        assert np.array_equal(x_d.cpu().numpy(), x_h)
        A = len(anchors)
        assert np.array_equal(yc_d.cpu().numpy().reshape(yc_h.shape), yc_h.astype(np.float32))
        assert np.array_equal(yb_d.cpu().numpy().reshape(yb_h.shape), yb_h.astype(np.float32))
        n_pos = int((yc_h[..., :A] & yc_h[..., A:]).sum())
        used = int(yc_h[..., :A].sum())
        assert used <= 256 and n_pos <= 128
        if case.startswith("file"):
            assert used > 0
        elif case == "many_positives":
            assert n_pos == 128, n_pos                            # (260 boxes leave hardly an anchor under 0.3 IoU with all of them: few negatives)
        else:
            assert used == 256
        assert not dev_mgr._dev                                   # the entry is consumed, like the reference's cache (rpn_util.py:121-123)


def _weights_equal(a, b):
    assert a.keys() == b.keys()
    for k in a:
        for u, v in zip(a[k], b[k]):
            assert np.array_equal(np.asarray(u), np.asarray(v)), k


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_train_rpn_loop_fast_feed_is_bit_identical(dtype, tmp_path):
    """train_util.train_rpn: 2 phases over 3 images (the list wraps and reshuffles: schedule.peek must hold the prefetch back there),
    once with the device-resident feed, once with the host calls: identical saved weights, identical RNG states."""
    from faster_rcnn_amd import resnet, rpn_util, train, train_util, util
    from faster_rcnn_amd.weights import load_npz, synthetic_resnet
    anchors = util.get_anchors([128, 256, 512])
    imgs0 = frames(3, 224, 320, seed=7) + frames(1, 224, 320, seed=8, src=(160, 230))
    res = {}
    for fast in (True, False):
        random.seed(1); np.random.seed(1337)
        imgs = list(imgs0)
        w = synthetic_resnet(50, anchors_per_loc=9, num_classes=21, seed=11)
        rpn = resnet.resnet50_rpn(resnet.resnet50_base(weight_regularizer=resnet.WEIGHT_REGULARIZER, bias_regularizer=resnet.BIAS_REGULARIZER, weights=w, dtype=dtype), anchors_per_loc=9)
        mgr = rpn_util.RpnTrainingManager(resnet.get_conv_rows_cols, 16, resnet.preprocess, anchors)
        dest = str(tmp_path / ("rpn_%s.npz" % fast))
        train_util.FAST_FEED = fast
        try:
            train_util.train_rpn(rpn, imgs, mgr, train.optimizer_from_str("sgd"), phases=[[6, 1e-3], [5, 1e-4]], save_frequency=5, save_weights_dest=dest)
        finally:
            train_util.FAST_FEED = True
        rpn.save_weights(dest)
        res[fast] = (load_npz(dest), random.getstate(), [im.name for im in imgs])
    _weights_equal(res[True][0], res[False][0])
    assert res[True][1] == res[False][1] and res[True][2] == res[False][2]
    assert not np.array_equal(res[True][0]["rpn_conv1"][0], synthetic_resnet(50, anchors_per_loc=9, num_classes=21, seed=11)["rpn_conv1"][0])


@pytest.mark.parametrize("step", ["step2", "step4"])
def test_train_detector_loop_fast_feed_is_bit_identical(step, tmp_path):
    """train_util.train_detector_step2 (detector with its own base, fed with images) and step4 (conv_only manager: the detector takes
    the RPN's conv4 map, which now never leaves the device): fast feed vs host calls, identical weights and np.random state."""
    from faster_rcnn_amd import det_util, resnet, train, train_util, util
    from faster_rcnn_amd.data.voc_data_helpers import VOC_CLASS_MAPPING
    from faster_rcnn_amd.weights import load_npz, synthetic_resnet
    anchors = util.get_anchors([128, 256, 512])
    imgs0 = frames(3, 224, 320, seed=9) + frames(1, 224, 320, seed=10, src=(150, 214))
    res = {}
    for fast in (True, False):
        random.seed(1); np.random.seed(1337)
        imgs = list(imgs0)
        rw = synthetic_resnet(50, anchors_per_loc=9, num_classes=21, seed=11)
        frozen = resnet.resnet50_rpn(resnet.resnet50_base(weights=rw), include_conv=(step == "step4"), anchors_per_loc=9)
        dw = synthetic_resnet(50, anchors_per_loc=9, num_classes=21, seed=12)
        reg = dict(weight_regularizer=resnet.WEIGHT_REGULARIZER, bias_regularizer=resnet.BIAS_REGULARIZER)
        if step == "step2":
            det = resnet.resnet50_classifier(64, 21, resnet.resnet50_base(weights=dw, **reg))
            loop = train_util.train_detector_step2
        else:
            det = resnet.resnet50_classifier(64, 21, weights=dw, **reg)
            loop = train_util.train_detector_step4
        mgr = det_util.DetTrainingManager(frozen, VOC_CLASS_MAPPING, resnet.preprocess, anchor_dims=anchors)
        assert mgr.conv_only == (step == "step4")
        dest = str(tmp_path / ("det_%s.npz" % fast))
        train_util.FAST_FEED = fast
        try:
            loop(det, imgs, mgr, train.optimizer_from_str("sgd"), phases=[[6, 1e-3], [3, 1e-4]], save_frequency=4, save_weights_dest=dest)
        finally:
            train_util.FAST_FEED = True
        det.save_weights(dest)
        res[fast] = (load_npz(dest), np.random.get_state()[1].copy(), random.getstate())
    _weights_equal(res[True][0], res[False][0])
    assert np.array_equal(res[True][1], res[False][1]) and res[True][2] == res[False][2]
    assert not np.array_equal(res[True][0]["res5a_branch2a"][0], synthetic_resnet(50, anchors_per_loc=9, num_classes=21, seed=12)["res5a_branch2a"][0])


def test_manager_stream_shares_no_queue_with_the_steps_streams():
    """feed.manager_stream(): ONE stream per process, chosen by probing -- work on it completes while the caller's stream, the step's
    weight-gradient stream and (if a candidate allows) its prefix stream are busy; a stream is never beside itself."""
    from faster_rcnn_amd import feed, train
    cur = torch.cuda.current_stream()
    assert not feed._runs_beside(cur, cur)
    s = feed.manager_stream()
    assert s is feed.manager_stream() and s is not cur
    assert feed._runs_beside(cur, s) and feed._runs_beside(train._wgrad_stream(), s)
    assert 1 <= feed.manager_stream.tried <= 8
    from faster_rcnn_amd import resnet, rpn_util, util
    mgr = rpn_util.RpnTrainingManager(resnet.get_conv_rows_cols, 16, resnet.preprocess, util.get_anchors([128, 256, 512]))
    assert mgr._own_stream() is s                                  # every manager object of the process uses it
