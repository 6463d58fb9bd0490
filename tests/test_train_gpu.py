"""GPU parity of one/two training steps (RPN step 1, detector step 2) vs the torch-autograd
restatement of Keras' compile + train_on_batch.  Weight UPDATES (new - old) are compared, scaled
by the largest update of the tensor: <= 2e-3 (f32 gradient sums over thousands of pixels vs f64)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def image(h, w, seed=0):
    rs = np.random.RandomState(seed)
    return (rs.randint(0, 256, (h, w, 3)).astype(np.float64) - np.array([103.939, 116.779, 123.68]))[None]


def check_updates(old, got, want, names, tol_fro=3e-2, tol_max=0.2):
    """Compare weight UPDATES.  Two bars per tensor: relative Frobenius error (the whole gradient) and
    max error over elements scaled by the largest update.  The max bar is looser because an f32
    pre-activation that lands within rounding of 0 can take the other ReLU branch than the f64 oracle
    (observed: one pixel of one channel of res4d_branch2c, 5e-3 of the max update); f32 master weights
    also cannot resolve an update below an ulp of the weight.
    Sizing of the bars: ONE flipped ReLU element in a 12x16x512 activation changes that layer's input
    gradient by 1/sqrt(#nonzero) ~ 0.5-1 % in Frobenius norm, and every layer below inherits it (measured
    on VGG block4_conv2 with tests/tools/debug_vgg_base.py: kernel vs conv_transpose 4.6e-7, one mask
    mismatch, 0.9 % gradient difference).  The kernels themselves are held to 1e-4 by test_conv_bwd_gpu."""
    for n in names:
        for o, g, w in zip(old[n], got[n], want[n]):
            dg, dw = np.asarray(g, np.float64) - o, np.asarray(w, np.float64) - o
            assert np.abs(dw).max() > 0, n
            ulp = 2.0 ** -23 * np.abs(o)
            err = np.maximum(np.abs(dg - dw) - 2 * ulp, 0)
            assert err.max() / np.abs(dw).max() < tol_max, (n, err.max() / np.abs(dw).max())
            assert np.sqrt((err ** 2).sum() / (dw ** 2).sum()) < tol_fro, (n, np.sqrt((err ** 2).sum() / (dw ** 2).sum()))


def rpn_targets(rows, cols, A, seed=1):
    rs = np.random.RandomState(seed)
    can_use = rs.rand(1, rows, cols, A) < 0.25
    is_pos = rs.rand(1, rows, cols, A) < 0.15
    y_class = np.concatenate([can_use, is_pos], axis=3)
    sel = np.repeat(can_use & is_pos, 4, axis=3).astype(np.float32)
    tg = (rs.randn(1, rows, cols, 4 * A) * is_pos.repeat(4, axis=3)).astype(np.float32)
    return y_class, np.concatenate([sel, tg], axis=3)


@pytest.mark.parametrize("opt_kind", ["sgd", "adam"])
def test_rpn_train_steps(opt_kind):
    from faster_rcnn_amd import resnet, train
    from faster_rcnn_amd.weights import synthetic_resnet
    from oracle import keras_train_ref as kt
    A = 9
    w0 = synthetic_resnet(50, anchors_per_loc=A, seed=7)
    old = {k: [np.array(a, dtype=np.float64) for a in v] for k, v in w0.items()}
    x = image(112, 144)
    rows, cols = resnet.get_conv_rows_cols(112, 144)
    y_class, y_bbreg = rpn_targets(rows, cols, A)
    base = resnet.resnet50_base(weights={k: [a.copy() for a in v] for k, v in w0.items()},
                                weight_regularizer=resnet.WEIGHT_REGULARIZER, bias_regularizer=resnet.BIAS_REGULARIZER)
    rpn = resnet.resnet50_rpn(base, anchors_per_loc=A)
    tr = train.RpnTrainer(rpn, l2=1e-4)
    lr = 1e-3
    opt = train.SGD(lr=lr, momentum=0.9) if opt_kind == "sgd" else train.Adam(lr=lr)
    tr.compile(opt)
    ref_opt = kt.Optim(opt_kind, lr)
    ref_w = w0
    names = kt.conv_layer_names(50, [4]) + ["rpn_conv1", "rpn_out_cls", "rpn_out_bbreg"]
    # Adam divides by sqrt(v)+1e-8: a weight whose gradient is ~1e-8 moves by an amount that depends on
    # the 1e-9-level rounding of an f32 sum, and every weight moves by ~lr regardless of |g|, so the
    # f32 product and the f64 oracle drift apart faster than under SGD (step-2 loss agrees to ~2e-3).
    loss_tol = [1e-4, 1e-4] if opt_kind == "sgd" else [1e-4, 5e-3]
    for step in range(2):                      # the second step exercises momentum / Adam slots and the re-packing
        losses = tr.train_on_batch(x, [y_class, y_bbreg])
        ref_w, ref_losses, _ = kt.rpn_train_step(ref_w, x, y_class, y_bbreg, A, ref_opt, l2=1e-4)
        for a, b in zip(losses, ref_losses):
            assert abs(a - b) <= loss_tol[step] * max(1.0, abs(b)), (step, losses, ref_losses)
    tr.sync_weights()
    if opt_kind == "sgd":
        check_updates(old, rpn.weights, ref_w, names, tol_fro=1e-3, tol_max=2e-2)     # no flip on this input: tight bars
    else:
        check_updates(old, rpn.weights, ref_w, names, tol_fro=5e-2, tol_max=2.5)
    # frozen layers untouched
    assert np.array_equal(rpn.weights["res3a_branch2a"][0], w0["res3a_branch2a"][0])
    # the model's inference path now runs on the trained weights
    cls, reg = rpn.predict_on_batch(x)[:2]
    from oracle.keras_ref import KerasGraphs
    g = KerasGraphs(ref_w, torch.float64)
    c64, r64 = g.rpn(g.resnet_base(x, 50))
    assert float((torch.as_tensor(reg).double() - r64).abs().max()) < (1e-3 if opt_kind == "sgd" else 5e-2)


def test_det_train_step():
    from faster_rcnn_amd import resnet, train
    from faster_rcnn_amd.weights import synthetic_resnet
    from oracle import keras_train_ref as kt
    C = 21
    w0 = synthetic_resnet(50, num_classes=C, seed=9)
    old = {k: [np.array(a, dtype=np.float64) for a in v] for k, v in w0.items()}
    x = image(112, 144, seed=2)
    rows, cols = resnet.get_conv_rows_cols(112, 144)
    rs = np.random.RandomState(4)
    n = 16
    x1 = rs.randint(0, cols - 2, n); y1 = rs.randint(0, rows - 2, n)
    rois = np.stack([x1, y1, np.minimum(cols - 1, x1 + 1 + rs.randint(0, 6, n)), np.minimum(rows - 1, y1 + 1 + rs.randint(0, 5, n))], axis=1).astype(np.float32)[None]
    cls_idx = rs.randint(0, C, n)
    y_class = np.zeros((1, n, C), np.int32); y_class[0, np.arange(n), cls_idx] = 1
    labels = np.zeros((n, 4 * (C - 1)), np.float32); targs = np.zeros((n, 4 * (C - 1)), np.float32)
    for i, c in enumerate(cls_idx):
        if c < C - 1:
            labels[i, 4 * c:4 * c + 4] = 1
            targs[i, 4 * c:4 * c + 4] = rs.randn(4) * 2
    y_bbreg = np.concatenate([labels, targs], axis=1)[None]
    base = resnet.resnet50_base(weights={k: [a.copy() for a in v] for k, v in w0.items()},
                                weight_regularizer=resnet.WEIGHT_REGULARIZER, bias_regularizer=resnet.BIAS_REGULARIZER)
    det = resnet.resnet50_classifier(n, C, base_model=base, weight_regularizer=resnet.WEIGHT_REGULARIZER, bias_regularizer=resnet.BIAS_REGULARIZER)
    tr = train.DetTrainer(det, l2=1e-4)
    tr.compile(train.SGD(lr=1e-3, momentum=0.9))
    ref_opt = kt.Optim("sgd", 1e-3)
    ref_w = w0
    for step in range(2):
        losses = tr.train_on_batch([x, rois], [y_class, y_bbreg])
        ref_w, ref_losses, _ = kt.det_train_step(ref_w, x, rois, y_class, y_bbreg, C, ref_opt, l2=1e-4)
        for a, b in zip(losses, ref_losses):
            assert abs(a - b) <= 1e-4 * max(1.0, abs(b)), (step, losses, ref_losses)
    tr.sync_weights()
    names = kt.conv_layer_names(50, [4, 5]) + ["dense_class_%d" % C, "dense_reg_%d" % C]
    check_updates(old, det.weights, ref_w, names)


def test_training_loops_through_reference_surface(tmp_path):
    """train_util.train_rpn / train_detector_step2 driven exactly like the reference scripts do, on
    synthetic in-memory images: a few iterations, loss decreases on the repeated image, weights saved."""
    import random
    from faster_rcnn_amd import det_util, resnet, rpn_util, shapes, train, train_util, util
    from faster_rcnn_amd.data.voc_data_helpers import VOC_CLASS_MAPPING
    from faster_rcnn_amd.weights import load_npz, synthetic_resnet
    random.seed(1); np.random.seed(1337)
    anchors = util.get_anchors([128, 256, 512])
    rs = np.random.RandomState(0)
    imgs = []
    for k in range(2):
        px = rs.randint(0, 256, (160, 224, 3)).astype(np.uint8)
        gts = [shapes.GroundTruthBox("dog", False, shapes.Box(20 + 10 * k, 30, 150, 140)), shapes.GroundTruthBox("cat", False, shapes.Box(120, 10, 210, 100))]
        imgs.append(shapes.Image(shapes.Metadata("im%d" % k, 224, 160, gts, "none"), px))
    w = synthetic_resnet(50, anchors_per_loc=9, num_classes=21, seed=11)
    base = resnet.resnet50_base(weight_regularizer=resnet.WEIGHT_REGULARIZER, bias_regularizer=resnet.BIAS_REGULARIZER, weights=w)
    rpn = resnet.resnet50_rpn(base, anchors_per_loc=9)
    mgr = rpn_util.RpnTrainingManager(resnet.get_conv_rows_cols, 16, resnet.preprocess, anchors)
    dest = str(tmp_path / "rpn_step1.npz")
    before = rpn.weights["rpn_conv1"][0].copy()
    train_util.train_rpn(rpn, imgs, mgr, train.optimizer_from_str("sgd"), phases=[[4, 1e-3], [2, 1e-4]], save_frequency=2, save_weights_dest=dest)
    saved = load_npz(dest)
    assert not np.array_equal(saved["rpn_conv1"][0], before)
    assert np.array_equal(saved["res2a_branch2a"][0], w["res2a_branch2a"][0])
    # step 2: detector on proposals from the (now frozen) step-1 RPN
    rpn_frozen = resnet.resnet50_rpn(resnet.resnet50_base(weights=saved), anchors_per_loc=9)
    dw = synthetic_resnet(50, anchors_per_loc=9, num_classes=21, seed=12)
    dw_before = dw["res5a_branch2a"][0].copy()
    det_base = resnet.resnet50_base(weight_regularizer=resnet.WEIGHT_REGULARIZER, bias_regularizer=resnet.BIAS_REGULARIZER, weights=dw)
    det = resnet.resnet50_classifier(64, 21, det_base)
    dmgr = det_util.DetTrainingManager(rpn_frozen, VOC_CLASS_MAPPING, resnet.preprocess, anchor_dims=anchors)
    ddest = str(tmp_path / "det_step2.npz")
    train_util.train_detector_step2(det, imgs, dmgr, train.optimizer_from_str("sgd"), phases=[[3, 1e-3]], save_frequency=2, save_weights_dest=ddest)
    dsaved = load_npz(ddest)
    assert "dense_class_21" in dsaved and dsaved["dense_class_21"][0].shape == (2048, 21)
    assert not np.array_equal(dsaved["res5a_branch2a"][0], dw_before)


# ------------------------------------------------------------------ the reference's own test flow (VGG16, Adam, 1 step)
def _voc_image():
    import os
    from faster_rcnn_amd.data.voc_data_helpers import extract_img_data
    return extract_img_data(os.path.join(os.path.dirname(__file__), "golden", "VOC_test"), "000005")


def test_vgg16_rpn_one_adam_step_like_train_rpn_test():
    """Mirror of the reference's train_rpn_test.py:21-46: VGG16 RPN, seeds 1337/1, ONE Adam step (lr 1e-5) on
    image 000005 (un-resized, 500x375), compare the block5_conv3 kernel.  The reference compares to a golden
    .h5 that is absent from its checkout; here the comparand is the autograd restatement (parity unpinned)."""
    import random
    from faster_rcnn_amd import rpn_util, train, train_util, util, vgg
    from faster_rcnn_amd.weights import synthetic_vgg16
    from oracle import keras_train_ref as kt
    np.random.seed(1337); random.seed(1)
    image = _voc_image()
    assert (image.width, image.height, image.num_gt_boxes) == (500, 375, 5)
    anchors = util.get_anchors([128, 256, 512])
    w0 = synthetic_vgg16(seed=21, with_classifier=False)
    old = {k: [np.array(a, dtype=np.float64) for a in v] for k, v in w0.items()}
    base = vgg.vgg16_base(weights={k: [a.copy() for a in v] for k, v in w0.items()})
    rpn = vgg.vgg16_rpn(base, anchors_per_loc=9)
    mgr = rpn_util.RpnTrainingManager(vgg.get_conv_rows_cols, vgg.STRIDE, vgg.preprocess, anchors)
    x = mgr.batched_image(image)
    y_class, y_bbreg = mgr.rpn_y_true(image)
    assert y_class.shape == (1, 23, 31, 18) and y_class[0, :, :, :9].sum() <= 256
    rpn.compile(train.Adam(lr=1e-5))
    losses = rpn.train_on_batch(x, [y_class, y_bbreg])
    ref_w, ref_losses, _ = kt.rpn_train_step(w0, x, y_class, y_bbreg, 9, kt.Optim("adam", 1e-5), freeze_blocks=(1, 2), arch="vgg")
    for a, b in zip(losses, ref_losses):
        assert abs(a - b) <= 1e-4 * max(1.0, abs(b)), (losses, ref_losses)
    rpn.save_weights("/tmp/_vgg_rpn_step.npz")
    got = rpn.get_layer("block5_conv3").get_weights()[0]
    # Adam's first step moves every weight by lr*g/(|g|+1e-8/sqrt(.001)) ~ +-lr: where the oracle's gradient is above the
    # epsilon's reach the update itself is compared (0.5 % of lr, plus the f32 master weight's own resolution); the rest by sign
    dg, dw = got.astype(np.float64) - old["block5_conv3"][0], ref_w["block5_conv3"][0] - old["block5_conv3"][0]
    assert np.abs(dw).max() > 5e-6
    big = np.abs(dw) > 0.9e-5                      # |g| well above 1e-8 / sqrt(0.001): the update has saturated at ~lr
    ulp = 2.0 ** -23 * np.abs(old["block5_conv3"][0])
    assert big.mean() > 0.5 and (np.maximum(np.abs(dg - dw) - 2 * ulp, 0)[big] < 5e-3 * 1e-5).all()
    agree = (np.sign(dg) == np.sign(dw)) | (np.abs(dw) < 2e-6)
    assert agree.mean() > 0.999, agree.mean()
    assert np.array_equal(rpn.get_layer("block2_conv2").get_weights()[0], w0["block2_conv2"][0])       # frozen blocks 1-2


def test_vgg16_det_step2_and_resnet_steps_3_4():
    from faster_rcnn_amd import resnet, train, vgg
    from faster_rcnn_amd.weights import synthetic_resnet, synthetic_vgg16
    from oracle import keras_train_ref as kt
    rs = np.random.RandomState(8)
    x = image(96, 128, seed=5)
    n, C = 8, 21

    def det_targets(rows, cols):
        x1 = rs.randint(0, cols - 2, n); y1 = rs.randint(0, rows - 2, n)
        rois = np.stack([x1, y1, np.minimum(cols - 1, x1 + 1 + rs.randint(0, 5, n)), np.minimum(rows - 1, y1 + 1 + rs.randint(0, 4, n))], axis=1).astype(np.float32)[None]
        ci = rs.randint(0, C, n)
        yc = np.zeros((1, n, C), np.int32); yc[0, np.arange(n), ci] = 1
        lab = np.zeros((n, 4 * (C - 1)), np.float32); tg = np.zeros((n, 4 * (C - 1)), np.float32)
        for i, c in enumerate(ci):
            if c < C - 1:
                lab[i, 4 * c:4 * c + 4] = 1; tg[i, 4 * c:4 * c + 4] = rs.randn(4)
        return rois, yc, np.concatenate([lab, tg], axis=1)[None]

    # ---- VGG16 detector step 2 (own base, blocks 3-5 + fc + dense train; train_det_test.py:28-60 flow)
    w0 = synthetic_vgg16(seed=22)
    old = {k: [np.array(a, dtype=np.float64) for a in v] for k, v in w0.items()}
    base = vgg.vgg16_base(weights={k: [a.copy() for a in v] for k, v in w0.items()})
    det = vgg.vgg16_classifier(n, C, base_model=base)
    rois, yc, yb = det_targets(6, 8)
    det.compile(train.SGD(1e-3, 0.9))
    losses = det.train_on_batch([x, rois], [yc, yb])
    ref_w, ref_losses, _ = kt.det_train_step(w0, x, rois, yc, yb, C, kt.Optim("sgd", 1e-3), freeze_blocks=(1, 2), arch="vgg")
    for a, b in zip(losses, ref_losses):
        assert abs(a - b) <= 1e-4 * max(1.0, abs(b)), (losses, ref_losses)
    det._trainer.sync_weights()
    check_updates(old, det.weights, ref_w, kt.vgg_conv_names((3, 4, 5)) + ["fc1", "fc2", "dense_class_21", "dense_reg_21"])

    # ---- ResNet-50 step 3: whole base frozen, only the (regularised) RPN heads train
    w0 = synthetic_resnet(50, seed=23)
    old = {k: [np.array(a, dtype=np.float64) for a in v] for k, v in w0.items()}
    base = resnet.resnet50_base(freeze_blocks=[1, 2, 3, 4], weights={k: [a.copy() for a in v] for k, v in w0.items()})
    rpn = resnet.resnet50_rpn(base, weight_regularizer=resnet.WEIGHT_REGULARIZER, bias_regularizer=resnet.BIAS_REGULARIZER, anchors_per_loc=9)
    rows, cols = resnet.get_conv_rows_cols(96, 128)
    y_class, y_bbreg = rpn_targets(rows, cols, 9)
    rpn.compile(train.SGD(1e-3, 0.9))
    losses = rpn.train_on_batch(x, [y_class, y_bbreg])
    ref_w, ref_losses, _ = kt.rpn_train_step(w0, x, y_class, y_bbreg, 9, kt.Optim("sgd", 1e-3), freeze_blocks=(1, 2, 3, 4), l2=1e-4, l2_base=False)
    for a, b in zip(losses, ref_losses):
        assert abs(a - b) <= 1e-4 * max(1.0, abs(b)), (losses, ref_losses)
    rpn._trainer.sync_weights()
    check_updates(old, rpn.weights, ref_w, ["rpn_conv1", "rpn_out_cls", "rpn_out_bbreg"])
    assert np.array_equal(rpn.weights["res4f_branch2c"][0], w0["res4f_branch2c"][0])

    # ---- ResNet-50 step 4: detector WITHOUT a base, fed with conv features
    w0 = synthetic_resnet(50, seed=24)
    old = {k: [np.array(a, dtype=np.float64) for a in v] for k, v in w0.items()}
    feat = np.maximum(rs.randn(1, rows, cols, 1024), 0).astype(np.float32)
    det = resnet.resnet50_classifier(n, C, weight_regularizer=resnet.WEIGHT_REGULARIZER, bias_regularizer=resnet.BIAS_REGULARIZER,
                                     weights={k: [a.copy() for a in v] for k, v in w0.items()})
    rois, yc, yb = det_targets(rows, cols)
    det.compile(train.SGD(1e-3, 0.9))
    losses = det.train_on_batch([feat, rois], [yc, yb])
    ref_w, ref_losses, _ = kt.det_train_step(w0, feat, rois, yc, yb, C, kt.Optim("sgd", 1e-3), l2=1e-4, with_base=False)
    for a, b in zip(losses, ref_losses):
        assert abs(a - b) <= 1e-4 * max(1.0, abs(b)), (losses, ref_losses)
    det._trainer.sync_weights()
    check_updates(old, det.weights, ref_w, kt.conv_layer_names(50, [5]) + ["dense_class_21", "dense_reg_21"])


@pytest.mark.parametrize("network", ["resnet50", "vgg16"])
def test_four_step_scripts_chain(tmp_path, network, capsys):
    """The reference's 4-step alternating training driven through the entry scripts with their own flags (two
    iterations each on the VOC_test fixture, small resize): every step reads the Keras-format .h5 the previous one
    wrote (h5lite), and the final pair feeds voc_dets."""
    import os
    from faster_rcnn_amd import h5lite, train_det_step2, train_det_step4, train_rpn_step1, train_rpn_step3, voc_dets
    voc = os.path.join(os.path.dirname(__file__), "golden", "VOC_test")
    common = ["--voc_paths", voc, "--network", network, "--phases", "2:1e-4", "--resize_dims", "208,288", "--img_set", "trainval"]
    f = lambda n: str(tmp_path / n)
    train_rpn_step1.main(common + ["--save_weights_dest", f("rpn1.h5"), "--save_model_dest", f("rpn1_model.h5")])
    assert h5lite.is_hdf5(f("rpn1.h5")) and "rpn_conv1" in h5lite.read_keras_weights(f("rpn1_model.h5"))
    train_det_step2.main([f("rpn1.h5")] + common + ["--save_weights_dest", f("det2.h5"), "--save_model_dest", f("det2_model.h5")])
    train_rpn_step3.main(common + ["--step2_weights_path", f("det2.h5"), "--save_weights_dest", f("rpn3.h5"), "--save_model_dest", f("rpn3_model.h5")])
    train_det_step4.main([f("rpn3.h5")] + common + ["--init_weights", f("det2.h5"), "--save_weights_dest", f("det4.h5"),
                                                    "--save_model_dest", f("det4_model.h5"), "--save_rpn_model_dest", f("rpn3_again.h5")])
    w3, w4 = h5lite.read_keras_weights(f("rpn3.h5")), h5lite.read_keras_weights(f("det4.h5"))
    assert "rpn_out_cls" in w3 and any(k.startswith("dense_class_") for k in w4)
    first = "block1_conv1" if network == "vgg16" else "conv1"
    assert np.array_equal(w3[first][0], h5lite.read_keras_weights(f("rpn1.h5"))[first][0])       # frozen layers never move
    dets = voc_dets.main([f("rpn3.h5"), f("det4.h5"), "--voc_path", voc, "--img_set", "trainval", "--network", network,
                          "--resize_dims", "208,288", "--out_dir", f("dets")])
    assert isinstance(dets, dict)
    capsys.readouterr()


# ------------------------------------------------------------------ Adam held tightly, one step at a time
def _slot_views(trainer, name):
    """(m, v) optimiser-slot views of layer `name`'s kernel inside the trainer's flat buffers."""
    p = trainer.params
    wv = p.views[name][0][0]
    off = (wv.data_ptr() - p.w.data_ptr()) // 4
    return [s[off:off + wv.numel()].view(wv.shape).cpu().numpy().astype(np.float64) for s in p.slots]


def test_rpn_adam_one_step_moments_and_update():
    """The reference's own check is a ONE-step Adam weight diff (train_rpn_test.py:32-46).  After one step nothing has
    accumulated, so each piece of the update is pinned separately against the f64 oracle:
      * the gradient itself (first moment / 0.1, L2 term included): relative Frobenius <= 1e-3, max <= 2e-2 of the largest
        entry -- the same bars the SGD step is held to;
      * the second moment / 0.001 against the squared oracle gradient;
      * the weight update wherever |g| > 1e-6 (below that Adam's 1e-8 epsilon amplifies f32 rounding of g): within 0.2 % of lr.
    Then a second compile() + step: the moments restart but the step counter runs on (t = 2), so the update is
    0.744 lr * sign(g), not lr -- Keras keeps `iterations` on the optimiser object (optimizers.py, Adam.get_updates)."""
    from faster_rcnn_amd import resnet, train
    from faster_rcnn_amd.weights import synthetic_resnet
    from oracle import keras_train_ref as kt
    A, lr = 9, 1e-3
    w0 = synthetic_resnet(50, anchors_per_loc=A, seed=31)
    x = image(112, 144, seed=6)
    rows, cols = resnet.get_conv_rows_cols(112, 144)
    y_class, y_bbreg = rpn_targets(rows, cols, A, seed=2)
    base = resnet.resnet50_base(weights={k: [a.copy() for a in v] for k, v in w0.items()},
                                weight_regularizer=resnet.WEIGHT_REGULARIZER, bias_regularizer=resnet.BIAS_REGULARIZER)
    rpn = resnet.resnet50_rpn(base, anchors_per_loc=A)
    opt = train.Adam(lr=lr)
    rpn.compile(opt)
    losses = rpn.train_on_batch(x, [y_class, y_bbreg])
    ref_opt = kt.Optim("adam", lr)
    ref_w, ref_losses, ref_g = kt.rpn_train_step(w0, x, y_class, y_bbreg, A, ref_opt, l2=1e-4)
    for a, b in zip(losses, ref_losses):
        assert abs(a - b) <= 1e-4 * max(1.0, abs(b)), (losses, ref_losses)
    names = kt.conv_layer_names(50, [4]) + ["rpn_conv1", "rpn_out_cls", "rpn_out_bbreg"]
    got_w = {n: rpn.get_layer(n).get_weights() for n in names}          # get_layer flushes the trainer (no manual sync)
    for n in names:
        g = ref_g[(n, 0)].numpy()
        m, v = _slot_views(rpn._trainer, n)
        eg = m / 0.1 - g
        assert np.sqrt((eg ** 2).sum() / (g ** 2).sum()) < 1e-3, (n, "gradient")
        assert np.abs(eg).max() < 2e-2 * np.abs(g).max(), (n, "gradient max")
        ev = v / 0.001 - g * g
        assert np.sqrt((ev ** 2).sum() / ((g * g) ** 2).sum()) < 2e-3, (n, "second moment")
        dg = got_w[n][0].astype(np.float64) - w0[n][0]
        dw = ref_w[n][0] - np.asarray(w0[n][0], np.float64)
        big = np.abs(g) > 1e-6
        assert big.mean() > 0.5
        ulp = 2.0 ** -23 * np.abs(w0[n][0])
        assert (np.maximum(np.abs(dg - dw) - 2 * ulp, 0)[big] < 2e-3 * lr).all(), (n, "update")
    # ---- next phase: same optimiser object, fresh slots, counter keeps running
    before = {n: got_w[n][0].astype(np.float64) for n in names}
    opt.lr = 1e-4
    rpn.compile(opt)
    rpn.train_on_batch(x, [y_class, y_bbreg])
    assert opt.iterations == 2
    ref_opt.recompile(lr=1e-4)
    ref_w2, _, ref_g2 = kt.rpn_train_step(ref_w, x, y_class, y_bbreg, A, ref_opt, l2=1e-4)
    scale_t2 = np.sqrt(1 - 0.999 ** 2) / (1 - 0.9 ** 2) * 0.1 / np.sqrt(0.001)          # 0.7444: |update| / lr at t = 2 with fresh moments
    for n in ("rpn_conv1", "res4f_branch2c"):
        g = ref_g2[(n, 0)].numpy()
        big = np.abs(g) > 1e-5
        step = rpn.get_layer(n).get_weights()[0].astype(np.float64) - before[n]
        ulp = 2.0 ** -23 * np.abs(before[n])
        # fresh moments at t = 2: m = 0.1 g, v = 0.001 g^2  ->  step = -lr * 0.7444 * |g| / (|g| + 1e-8 / sqrt(0.001)) * sign(g)
        want = -1e-4 * scale_t2 * g / (np.abs(g) + 1e-8 / np.sqrt(0.001))
        assert big.mean() > 0.3
        # lr alone (t restarted) would be off by 0.26 lr.  Per element: within 0.02 lr -- except for a handful of elements (fewer than
        # one in ten thousand) whose f32 gradient has the other SIGN than the f64 oracle's: Adam's step is lr-sized whatever |g| is, and
        # one pre-activation that lands within f32 rounding of zero takes the other ReLU branch than the oracle (check_updates' note:
        # which one depends on the summation order of the matrix engine in use)
        err = np.maximum(np.abs(step - want) - 2 * ulp, 0)[big]
        assert (err < 0.02 * 1e-4).mean() > 0.9999 and err.max() < 2 * 1e-4, (n, err.max(), (err < 0.02 * 1e-4).mean())
        assert np.abs(np.abs(step[np.abs(g) > 1e-4]).mean() / 1e-4 - scale_t2) < 0.01          # the 0.744, not 1.0


def test_weights_are_current_at_every_way_out_of_the_model(tmp_path):
    """get_layer().get_weights() straight after train_on_batch (the reference's usage, train_rpn_test.py:38-41),
    DetModel.predict after training, save_weights, and load_weights AFTER compile (training continues from the
    loaded values; the next flush must not overwrite them)."""
    from faster_rcnn_amd import resnet, train
    from faster_rcnn_amd.weights import load_npz, synthetic_resnet
    A, C, n = 9, 21, 8
    w0 = synthetic_resnet(50, anchors_per_loc=A, num_classes=C, seed=41)
    x = image(96, 128, seed=7)
    rows, cols = resnet.get_conv_rows_cols(96, 128)
    y_class, y_bbreg = rpn_targets(rows, cols, A, seed=3)
    base = resnet.resnet50_base(weights={k: [a.copy() for a in v] for k, v in w0.items()})
    rpn = resnet.resnet50_rpn(base, anchors_per_loc=A)
    rpn.compile(train.SGD(1e-2, 0.9))
    rpn.train_on_batch(x, [y_class, y_bbreg])
    k1 = rpn.get_layer("rpn_conv1").get_weights()[0]
    assert not np.array_equal(k1, w0["rpn_conv1"][0])                    # no stale pre-training weights
    p1 = rpn.predict_on_batch(x)[1]
    rpn.save_weights(str(tmp_path / "a.npz"))
    assert np.array_equal(load_npz(str(tmp_path / "a.npz"))["rpn_conv1"][0], k1)
    # load_weights after compile: the ORIGINAL weights come back and training continues from them
    from faster_rcnn_amd.weights import save_weights_file
    save_weights_file(str(tmp_path / "orig.npz"), w0)
    rpn.load_weights(str(tmp_path / "orig.npz"))
    assert np.array_equal(rpn.get_layer("rpn_conv1").get_weights()[0], w0["rpn_conv1"][0])
    l_again = rpn.train_on_batch(x, [y_class, y_bbreg])
    fresh = resnet.resnet50_rpn(resnet.resnet50_base(weights={k: [a.copy() for a in v] for k, v in w0.items()}), anchors_per_loc=A)
    fresh.compile(train.SGD(1e-2, 0.9))
    l_fresh = fresh.train_on_batch(x, [y_class, y_bbreg])
    assert l_again[1:] == l_fresh[1:]                                    # same forward: the loaded weights reached the trainer
    # detector: predict right after a step runs on the trained head
    dw = synthetic_resnet(50, anchors_per_loc=A, num_classes=C, seed=42)
    dense0 = dw["dense_class_%d" % C][0].copy()                          # the model trains INTO this dict
    det = resnet.resnet50_classifier(n, C, base_model=resnet.resnet50_base(weights=dw))
    rs = np.random.RandomState(1)
    rois = np.stack([rs.randint(0, 3, n), rs.randint(0, 2, n), rs.randint(4, cols - 1, n), rs.randint(3, rows - 1, n)], axis=1).astype(np.float32)[None]
    yc = np.zeros((1, n, C), np.int32); yc[0, np.arange(n), rs.randint(0, C, n)] = 1
    yb = np.zeros((1, n, 8 * (C - 1)), np.float32)
    before = det.predict([x, rois])[0]
    det.compile(train.SGD(1e-2, 0.9))
    det.train_on_batch([x, rois], [yc, yb])
    after = det.predict([x, rois])[0]
    assert not np.array_equal(before, after)
    assert not np.array_equal(det.get_layer("dense_class_%d" % C).get_weights()[0], dense0)


def test_deferred_steps_are_the_same_steps():
    """train_on_batch(defer=True) only enqueues the step (train_util's loops read the losses one step late, so the
    host prepares the next image while the GPU trains): three deferred steps on three DIFFERENT images -- staged into
    the alternating pinned sets while the previous step is still running -- give bit for bit the losses and weights of
    three synchronous steps, in any order of reading the results, also when the 8-slot loss ring wraps."""
    from faster_rcnn_amd import resnet, train
    from faster_rcnn_amd.weights import synthetic_resnet
    A = 9
    w0 = synthetic_resnet(50, anchors_per_loc=A, num_classes=21, seed=43)
    rows, cols = resnet.get_conv_rows_cols(96, 128)
    xs = [image(96, 128, seed=20 + i) for i in range(11)]
    ys = [rpn_targets(rows, cols, A, seed=30 + i) for i in range(11)]

    def run(defer):
        rpn = resnet.resnet50_rpn(resnet.resnet50_base(weights={k: [a.copy() for a in v] for k, v in w0.items()}), anchors_per_loc=A)
        rpn.compile(train.SGD(1e-2, 0.9))
        got = [rpn.train_on_batch(x, list(y), defer=defer) for x, y in zip(xs, ys)]
        if defer:
            assert all(isinstance(g, train.PendingLosses) for g in got)
            got = [g.result() for g in reversed(got)][::-1]             # last first; the first three were read when the ring wrapped
        return got, rpn.get_layer("rpn_conv1").get_weights()[0], rpn.get_layer("res4f_branch2c").get_weights()[0]

    l_sync, a_sync, b_sync = run(False)
    l_def, a_def, b_def = run(True)
    assert l_sync == l_def
    assert np.array_equal(a_sync, a_def) and np.array_equal(b_sync, b_def)

    # the detector step (frozen stages 1-3 of image i+1 on the second stream beside the backward pass of image i)
    C, n = 21, 8
    dw0 = synthetic_resnet(50, anchors_per_loc=A, num_classes=C, seed=44)
    rs = np.random.RandomState(5)
    batches = []
    for i in range(5):
        rois = np.stack([rs.randint(0, 3, n), rs.randint(0, 2, n), rs.randint(4, cols - 1, n), rs.randint(3, rows - 1, n)], axis=1).astype(np.float32)[None]
        yc = np.zeros((1, n, C), np.int32); yc[0, np.arange(n), rs.randint(0, C, n)] = 1
        yb = (rs.randn(1, n, 8 * (C - 1)) * (rs.rand(1, n, 8 * (C - 1)) < 0.1)).astype(np.float32)
        batches.append(([xs[i], rois], [yc, yb]))

    def run_det(defer):
        det = resnet.resnet50_classifier(n, C, base_model=resnet.resnet50_base(weights={k: [a.copy() for a in v] for k, v in dw0.items()}))
        det.compile(train.SGD(1e-2, 0.9))
        got = [det.train_on_batch(x, y, defer=defer) for x, y in batches]
        got = [g.result() for g in got] if defer else got
        return got, det.get_layer("res4a_branch2a").get_weights()[0], det.get_layer("dense_class_%d" % C).get_weights()[0]

    d_sync, d_def = run_det(False), run_det(True)
    assert d_sync[0] == d_def[0]
    assert np.array_equal(d_sync[1], d_def[1]) and np.array_equal(d_sync[2], d_def[2])


def test_many_steps_keep_device_memory_flat():
    """The step driver allocates on three streams (upload + frozen stages, main, loss read-back) and hands tensors across
    them (record_stream): 60 RPN steps, every other one deferred, must neither grow the allocator's reservation nor
    produce a non-finite loss."""
    from faster_rcnn_amd import resnet, train
    from faster_rcnn_amd.weights import synthetic_resnet
    A = 9
    w0 = synthetic_resnet(50, anchors_per_loc=A, num_classes=21, seed=45)
    rows, cols = resnet.get_conv_rows_cols(192, 256)
    rpn = resnet.resnet50_rpn(resnet.resnet50_base(weights=w0, dtype="bf16"), anchors_per_loc=A)
    rpn.compile(train.SGD(1e-4, 0.9))
    xs = [image(192, 256, seed=50 + i) for i in range(4)]
    ys = [rpn_targets(rows, cols, A, seed=60 + i) for i in range(4)]
    reserved = []
    for it in range(60):
        l = rpn.train_on_batch(xs[it % 4], list(ys[it % 4]), defer=bool(it & 1))
        v = l.result() if hasattr(l, "result") else l
        assert all(np.isfinite(v)), (it, v)
        if it in (19, 59):
            torch.cuda.synchronize()
            reserved.append(torch.cuda.memory_reserved())
    assert reserved[1] <= reserved[0], reserved


def test_rpn_steps_on_the_f16x3_engine_track_the_exact_split():
    """train.F32_ENGINE = "f16x3" (opt-in): forward and input-gradient launches on the f16x3 engine with per-step magnitude-record
    arenas and the batched plane refresh -- two SGD steps on a 320x480 image (stages 2-3 and the RPN layers pass the engine's policy)
    against the same steps on the default exact split: losses to 1e-5, weight updates to 1e-3 of each tensor's (Frobenius)."""
    from faster_rcnn_amd import ops, resnet, train
    from faster_rcnn_amd.weights import synthetic_resnet
    A = 9
    w0 = synthetic_resnet(50, anchors_per_loc=A, seed=7)
    x = image(320, 480)
    rows, cols = resnet.get_conv_rows_cols(320, 480)
    y_class, y_bbreg = rpn_targets(rows, cols, A)
    res = {}
    for eng in ("bf16x6", "f16x3"):
        prev, train.F32_ENGINE = train.F32_ENGINE, eng
        try:
            base = resnet.resnet50_base(weights={k: [a.copy() for a in v] for k, v in w0.items()},
                                        weight_regularizer=resnet.WEIGHT_REGULARIZER, bias_regularizer=resnet.BIAS_REGULARIZER)
            rpn = resnet.resnet50_rpn(base, anchors_per_loc=A)
            tr = train.RpnTrainer(rpn, l2=1e-4)
            tr.compile(train.SGD(lr=1e-3, momentum=0.9))
            measured0 = ops.AMAX_MEASURED
            losses = [tr.train_on_batch(x, [y_class, y_bbreg]) for _ in range(2)]
            tr.sync_weights()
            res[eng] = (losses, {k: [np.array(a, np.float64) for a in v] for k, v in rpn.weights.items()}, ops.AMAX_MEASURED - measured0)
            if eng == "f16x3":
                live = [pk for c in tr._tconvs() for pk in (c.pc, getattr(c, "pd", None)) if getattr(pk, "_h3", None) is not None]
                assert len(live) >= 10                               # trainable filters did run on the engine, forward and backward
                for pk in live[:6]:                                  # ... and their planes are those of the UPDATED weights
                    have = pk._h3.clone()
                    pk._h3 = None
                    assert torch.equal(have, pk.h3_planes())
        finally:
            train.F32_ENGINE = prev
    (la, wa, _), (lb, wb, measured) = res["bf16x6"], res["f16x3"]
    for a, b in zip(np.ravel(la), np.ravel(lb)):
        assert abs(a - b) <= 1e-5 * max(1.0, abs(a)), (la, lb)
    assert 0 < measured <= 2 * 12, measured                          # a few tensors per step have no producing conv launch (the image, loss gradients)
    for k in wa:
        for o, a, b in zip(w0[k], wa[k], wb[k]):
            da, db = a - np.asarray(o, np.float64), b - np.asarray(o, np.float64)
            if np.abs(da).max() > 0:
                assert np.sqrt(((da - db) ** 2).sum() / (da ** 2).sum()) < 1e-3, k
