"""Known-answer tests for the UNPINNED half of the oracle (oracle/keras_ref.py, oracle/keras_train_ref.py).

TensorFlow 1.3 / Keras 2.0.8 cannot run here and the reference's golden .h5 files are missing, so the conv / BN /
pool / RoI-resize / loss / optimiser restatement has no reference output to be checked against.  These vectors are
derived BY HAND from the published algorithms the reference calls (each test cites the call site in
/root/reference/faster_rcnn and the third-party routine): every expected number below is a literal or a closed
formula evaluated with Python's math module, never a call into the code under test.  Where a plausible wrong
implementation exists (symmetric SAME padding, half-pixel bilinear, mask inside the RPN regression sum, a step
counter that restarts per compile) the test states what THAT would give, and the expected value differs from it.

CPU tests pin the oracle; the ``gpu`` twins push the same inputs through the C ABI and require the same literals,
so the product is held to the hand-derived answers directly, not only to the oracle.
"""
import math

import numpy as np
import pytest

torch = pytest.importorskip("torch")

from oracle import keras_ref as kr
from oracle import keras_train_ref as kt


# ---------------------------------------------------------------------------------------------------------------
# 1. TF SAME padding (Keras padding='same'), used by conv1 7x7/2 (resnet.py:408), every 3x3 (resnet.py:161, 229) and
#    VGG (vgg.py:96-137).  TF: out = ceil(in / stride); pad_along = max((out-1)*stride + k - in, 0);
#    pad_before = pad_along // 2, pad_after = pad_along - pad_before  (the EXTRA pixel goes to the end).
#    All-ones kernels turn the conv into window sums of x = arange(h*w): integers, exact in f32.
SAME_CASES = {
    # 6x6, 3x3 stride 2: out 3, pad_along 1 -> (0, 1): windows start at 0, 2, 4 and run off the END only.
    # (symmetric-before padding would give out[0,0] = x[0:2,0:2] = 14, not 63)
    "even6_k3_s2": (6, 3, 2, [[63, 81, 63], [171, 189, 135], [168, 180, 126]]),
    # 5x5, 3x3 stride 2: out 3, pad_along 2 -> (1, 1)
    "odd5_k3_s2": (5, 3, 2, [[12, 27, 24], [63, 108, 81], [72, 117, 84]]),
    # 8x8, 7x7 stride 2 (the stem on an even size): out 4, pad_along 5 -> (2, 3): first window rows -2..4
    "even8_k7_s2_corners": (8, 7, 2, None),
    # 5x5, 3x3 stride 1: out 5, pad (1, 1)
    "odd5_k3_s1_corners": (5, 3, 1, None),
}


def _window_sums(n, k, stride, pad_before):
    """Independent restatement for the corner-only cases: sum of x = arange(n*n) over each window clipped to the image."""
    x = np.arange(n * n, dtype=np.int64).reshape(n, n)
    out_n = -(-n // stride)
    out = np.zeros((out_n, out_n), dtype=np.int64)
    for i in range(out_n):
        for j in range(out_n):
            r0, c0 = i * stride - pad_before, j * stride - pad_before
            out[i, j] = x[max(r0, 0):min(r0 + k, n), max(c0, 0):min(c0 + k, n)].sum()
    return out


def _same_expected(name):
    n, k, stride, lit = SAME_CASES[name]
    if lit is not None:
        return n, k, stride, np.array(lit, dtype=np.float64)
    if name == "even8_k7_s2_corners":
        want = _window_sums(8, 7, 2, 2)
        # hand-computed corners: rows/cols 0..4 -> 40*(0+1+2+3+4) + 5*10 = 450; rows/cols 4..7 -> 4*8*22 + 4*22 = 792
        assert want[0, 0] == 450 and want[3, 3] == 792
    else:
        want = _window_sums(5, 3, 1, 1)
        assert want[0, 0] == 0 + 1 + 5 + 6 and want[4, 4] == 18 + 19 + 23 + 24 and want[2, 2] == 9 * 12
    return n, k, stride, want.astype(np.float64)


@pytest.mark.parametrize("name", sorted(SAME_CASES))
def test_oracle_same_padding(name):
    n, k, stride, want = _same_expected(name)
    x = np.arange(n * n, dtype=np.float32).reshape(1, n, n, 1)
    w = np.ones((k, k, 1, 1), dtype=np.float32)
    y = kr.conv2d(x, w, None, stride, "same", torch.float64)
    assert np.array_equal(np.asarray(y)[0, :, :, 0], want)
    out, before, after = kr.same_pad(n, k, stride)
    assert out == want.shape[0] and before <= after and before + after == max((out - 1) * stride + k - n, 0)


def test_oracle_valid_conv_and_pools():
    """1x1 stride-2 convs are VALID (resnet.py:218, 238): out = (in-1)//2 + 1 and they sample pixels 0, 2, 4, ...;
    MaxPooling2D((3,3), strides=(2,2)) is VALID (resnet.py:412): out = (in-3)//2 + 1; VGG pools 2x2/2 floor (vgg.py:100)."""
    x = np.arange(49, dtype=np.float32).reshape(1, 7, 7, 1)
    y = np.asarray(kr.conv2d(x, np.ones((1, 1, 1, 1), np.float32), None, 2, "valid", torch.float64))[0, :, :, 0]
    assert np.array_equal(y, [[0, 2, 4, 6], [14, 16, 18, 20], [28, 30, 32, 34], [42, 44, 46, 48]])
    p = np.asarray(kr.pool2d(torch.as_tensor(x), 3, 2, True))[0, :, :, 0]
    assert np.array_equal(p, [[16, 18, 20], [30, 32, 34], [44, 46, 48]])          # window max = its bottom-right element
    p2 = np.asarray(kr.pool2d(torch.as_tensor(x), 2, 2, True))[0, :, :, 0]
    assert np.array_equal(p2, [[8, 10, 12], [22, 24, 26], [36, 38, 40]])           # 7 -> 3: the last row/column is dropped


# ---------------------------------------------------------------------------------------------------------------
# 2. TF-1.3 bilinear resize, align_corners=False, NO half-pixel offset (custom_layers.py:50 -> tf.image.resize_images ->
#    resize_bilinear_op.cc: scale = in/out; src = i*scale; lo = floor(src); hi = min(lo+1, in-1); lerp = src - lo).
#    A 3x5 crop -> 7x7.  Lerp table (exact fractions): rows  i*3/7 -> lo 0,0,0,1,1,2,2  hi 1,1,1,2,2,2,2
#                                                      cols  j*5/7 -> lo 0,0,1,2,2,3,4  hi 1,1,2,3,3,4,4
#    On the linear ramp f[y][x] = 10*y + x the result is 10*Y[i] + X[j] with the sampled coordinates below; at the
#    clamped taps (lo == hi) the coordinate is lo itself.  A half-pixel-centre resize (TF2 / cv2 style:
#    src = (i+.5)*scale - .5) would give Y[1] = 0.142857, not 0.428571.
Y7 = [0.0, 3 / 7, 6 / 7, 9 / 7, 12 / 7, 2.0, 2.0]
X7 = [0.0, 5 / 7, 10 / 7, 15 / 7, 20 / 7, 25 / 7, 4.0]


def _roi_case():
    feat = np.zeros((4, 6, 2), dtype=np.float32)
    for y in range(4):
        for x in range(6):
            feat[y, x, 0] = 10 * y + x
            feat[y, x, 1] = -(10 * y + x) * 0.5
    # rois are (x1, y1, x2, y2) in conv cells, x2 / y2 EXCLUSIVE (custom_layers.py:45-50): rows 1..3, cols 1..5
    rois = np.array([[1, 1, 6, 4]], dtype=np.float32)
    want = np.zeros((1, 7, 7, 2), dtype=np.float64)
    for i in range(7):
        for j in range(7):
            v = 10 * (Y7[i] + 1) + (X7[j] + 1)
            want[0, i, j] = (v, -0.5 * v)
    return feat, rois, want


def test_oracle_legacy_bilinear_3x5_to_7x7():
    feat, rois, want = _roi_case()
    got = kr.roi_resize(feat, rois, 7)
    assert np.abs(got - want).max() < 2e-5                       # f32 lerps on values up to 36
    got_t = kr.roi_resize_torch(torch.as_tensor(feat, dtype=torch.float64), rois, 7).numpy()
    assert np.abs(got_t - want).max() < 2e-6
    # a crop at least 7 wide is POINT-sampled where src is an integer and never reads row/col 'in' (the exclusive end)
    ramp = np.arange(14, dtype=np.float32).reshape(1, 14, 1) * np.ones((1, 1, 1), np.float32)
    got = kr.roi_resize(ramp, np.array([[0, 0, 14, 1]], np.float32), 7)
    assert np.allclose(got[0, 0, :, 0], [0, 2, 4, 6, 8, 10, 12], atol=1e-5)


# THIRD-PARTY vector (not derived here): TensorFlow 1.x's own unit test for the op the reference calls,
# tensorflow/python/ops/image_ops_test.py ResizeImagesTest.testResizeUp -- a 3x2 single-channel image resized to 6x4 with
# ResizeMethod.BILINEAR (align_corners=False, the legacy kernel without half-pixel centres that tf.image.resize_images
# reaches from custom_layers.py:50).  The first number in the Keras half of the oracle that someone other than this
# repository's author published.
TF_RESIZE_UP_IN = [64, 32, 32, 64, 50, 100]                               # [3, 2] row-major
TF_RESIZE_UP_BILINEAR = [64.0, 48.0, 32.0, 32.0,
                         48.0, 48.0, 48.0, 48.0,
                         32.0, 48.0, 64.0, 64.0,
                         41.0, 61.5, 82.0, 82.0,
                         50.0, 75.0, 100.0, 100.0,
                         50.0, 75.0, 100.0, 100.0]                        # [6, 4]


def test_oracle_tf_published_resize_up_vector():
    feat = np.array(TF_RESIZE_UP_IN, np.float32).reshape(3, 2, 1)
    got = kr.roi_resize(feat, np.array([[0, 0, 2, 3]], np.float32), (6, 4))
    assert np.array_equal(got.reshape(-1), np.array(TF_RESIZE_UP_BILINEAR, np.float32))
    # the wrong kernels give something else: half-pixel centres (TF2 / OpenCV convention) and align_corners=True
    t = torch.tensor(TF_RESIZE_UP_IN, dtype=torch.float32).reshape(1, 1, 3, 2)
    for kw in ({"align_corners": False}, {"align_corners": True}):
        other = torch.nn.functional.interpolate(t, size=(6, 4), mode="bilinear", **kw).reshape(-1).numpy()
        assert not np.allclose(other, TF_RESIZE_UP_BILINEAR)


# ---------------------------------------------------------------------------------------------------------------
# 3. BatchNormalization(training=False) with the two epsilons the reference uses: Keras default 1e-3 for bn_conv1
#    (resnet.py:410) and 1e-5 inside the blocks (resnet.py:148, 216).  y = gamma*(x-mean)/sqrt(var+eps) + beta.
def test_oracle_batchnorm_both_epsilons():
    x, g, b, m, v = 2.0, 1.5, 0.25, 0.5, 0.25
    for eps in (1e-3, 1e-5):
        want = g * (x - m) / math.sqrt(v + eps) + b
        got = kr.batchnorm_inference(torch.tensor([x], dtype=torch.float64), [g], [b], [m], [v], eps)
        assert abs(float(got[0]) - want) < 1e-12
    # literal values (2.25 / sqrt(0.251) + 0.25, 2.25 / sqrt(0.25001) + 0.25): the two epsilons are far apart at f32
    assert abs((g * 1.5 / math.sqrt(0.251) + b) - 4.741026910) < 1e-8 and abs((g * 1.5 / math.sqrt(0.25001) + b) - 4.749910003) < 1e-8
    # Scale (custom_layers.py:126-128) after BN: gamma2 * y + beta2
    kg = kr.KerasGraphs({"s": [np.array([2.0]), np.array([-1.0])]}, torch.float64)
    assert float(kg.scale(torch.tensor([3.0], dtype=torch.float64), "s")[0]) == 5.0


# ---------------------------------------------------------------------------------------------------------------
# 4. Losses (loss_functions.py:15-76) on hand-sized inputs.
LN2 = math.log(2.0)


def test_oracle_cls_loss_rpn_keras_bce_clip():
    """K.binary_crossentropy (Keras 2.0.8, TF backend): clip p to [1e-7, 1-1e-7], logits = log(p/(1-p)),
    sigmoid_cross_entropy_with_logits; the loss is sum(can_use * bce) / 256 (loss_functions.py:24)."""
    A = 1
    # cells: (selected, positive, p)
    y_true = torch.tensor([[[[1.0, 1.0]], [[1.0, 0.0]], [[0.0, 1.0]], [[1.0, 1.0]], [[1.0, 1.0]]]], dtype=torch.float64)   # (1,5,1,2A)
    y_pred = torch.tensor([[[[0.5]], [[0.25]], [[0.9]], [[0.0]], [[1.0]]]], dtype=torch.float64)
    want = (LN2 - math.log(0.75) + 0.0 - math.log(1e-7) - math.log(1.0 - 1e-7)) / 256.0
    assert abs(-math.log(1e-7) - 16.11809565095832) < 1e-12      # p = 0 with a positive target is clipped, not infinite
    assert abs(float(kt.cls_loss_rpn(y_true, y_pred, A)) - want) < 1e-12


def test_oracle_bbreg_loss_rpn_mask_outside_the_sum():
    """loss_functions.py:44 as written: 10 * mask * K.sum(smoothL1(t - p)) / 2400 is a TENSOR (the sum encloses only
    the smooth-L1 term), which Keras then averages: loss = mean(mask) * 10 * S / 2400 with S over ALL anchors."""
    A = 1
    y_true = torch.tensor([[[[1, 1, 1, 1, 0.5, -2.0, 0.0, 1.0]], [[0, 0, 0, 0, 0.0, 0.0, 0.0, 0.0]]]], dtype=torch.float64)   # (1,2,1,8A)
    y_pred = torch.tensor([[[[0.0, 0.0, 0.0, 0.0]], [[3.0, 0.0, 0.0, 0.0]]]], dtype=torch.float64)
    S = (0.5 * 0.25 + (2.0 - 0.5) + 0.0 + 0.5 * 1.0) + (3.0 - 0.5)        # 2.125 from the selected cell + 2.5 from the UNSELECTED one
    want = 0.5 * 10.0 * S / 2400.0                                         # mean(mask) = 4/8
    assert abs(want - 0.009635416666666667) < 1e-15
    assert abs(float(kt.bbreg_loss_rpn(y_true, y_pred, A)) - want) < 1e-15
    assert abs(want - 10.0 * 2.125 / 2400.0) > 5e-4                        # what the paper's masked sum would give


def test_oracle_detector_losses():
    """bbreg_loss_det (loss_functions.py:65): sum(mask*smoothL1) / sum(1e-4 + mask), the 1e-4 added PER ELEMENT;
    cls_loss_det (:76): mean over RoIs of categorical cross-entropy with Keras' renormalise + clip 1e-7."""
    y_true = torch.tensor([[1, 1, 1, 1, 1.0, 1.0, 1.0, 1.0], [0, 0, 0, 0, 9.0, 9.0, 9.0, 9.0]], dtype=torch.float64)
    y_pred = torch.tensor([[0.0, 0.0, 0.0, 3.0], [5.0, 5.0, 5.0, 5.0]], dtype=torch.float64)
    want = (0.5 + 0.5 + 0.5 + 1.5) / (8 * 1e-4 + 4.0)
    assert abs(want - 0.7498500299940012) < 1e-15
    assert abs(float(kt.bbreg_loss_det(y_true, y_pred, 1)) - want) < 1e-15
    yc = torch.tensor([[1.0, 0.0], [0.0, 1.0], [1.0, 0.0]], dtype=torch.float64)
    pc = torch.tensor([[0.25, 0.75], [0.0, 1.0], [2.0, 6.0]], dtype=torch.float64)    # last row un-normalised: Keras divides by its sum
    want = (-math.log(0.25) - math.log(1.0 - 1e-7) - math.log(0.25)) / 3.0
    assert abs(float(kt.cls_loss_det(yc, pc)) - want) < 1e-12
    assert abs(kt.smooth_l1(torch.tensor([1.0], dtype=torch.float64))[0] - 0.5) < 1e-15      # |d| == 1 takes the quadratic branch


# ---------------------------------------------------------------------------------------------------------------
# 5. Optimisers (args_util.py:56-59 -> Keras 2.0.8 optimizers.py).
def _adam_update(g, lr, t, m=0.0, v=0.0, b1=0.9, b2=0.999, eps=1e-8):
    m = b1 * m + (1 - b1) * g
    v = b2 * v + (1 - b2) * g * g
    lr_t = lr * math.sqrt(1 - b2 ** t) / (1 - b1 ** t)
    return -lr_t * m / (math.sqrt(v) + eps), m, v


def test_oracle_sgd_momentum_two_steps():
    """Keras SGD.get_updates: v = momentum*v - lr*g; w = w + v (no Nesterov, no decay)."""
    opt = kt.Optim("sgd", 0.1, momentum=0.9)
    w = {"w": torch.tensor([1.0], dtype=torch.float64)}
    g = {"w": torch.tensor([0.5], dtype=torch.float64)}
    w = opt.step(w, g)
    assert abs(float(w["w"][0]) - 0.95) < 1e-15                   # v = -0.05
    w = opt.step(w, g)
    assert abs(float(w["w"][0]) - 0.855) < 1e-15                  # v = 0.9*(-0.05) - 0.05 = -0.095


def test_oracle_adam_steps_and_recompile():
    """Keras Adam.get_updates: t = iterations + 1; lr_t = lr*sqrt(1-b2^t)/(1-b1^t); m, v as usual;
    w -= lr_t*m/(sqrt(v)+eps).  ``iterations`` belongs to the optimiser OBJECT and keeps counting when the reference
    re-compiles for the next phase (train_util.py:29-33), while m and v are re-created as zeros."""
    lr, g = 1e-3, 0.5
    opt = kt.Optim("adam", lr)
    w = {"w": torch.tensor([1.0], dtype=torch.float64)}
    gd = {"w": torch.tensor([g], dtype=torch.float64)}
    d1, m1, v1 = _adam_update(g, lr, 1)
    assert abs(d1 + lr) < 1e-9                                    # the first Adam step is ~ -lr * sign(g)
    w = opt.step(w, gd)
    assert abs(float(w["w"][0]) - (1.0 + d1)) < 1e-15
    d2, _, _ = _adam_update(g, lr, 2, m1, v1)
    w = opt.step(w, gd)
    assert abs(float(w["w"][0]) - (1.0 + d1 + d2)) < 1e-15
    # next phase: recompile -> fresh moments, t continues at 3
    opt.recompile()
    d3, _, _ = _adam_update(g, lr, 3)
    assert abs(d3 / lr + math.sqrt(1 - 0.999 ** 3) / (1 - 0.9 ** 3) * 0.1 / math.sqrt(0.001)) < 1e-6     # = -0.639 lr ...
    assert abs(d3 + lr) > 0.3 * lr                                                                        # ... not the -lr a restarted counter gives
    w = opt.step(w, gd)
    assert abs(float(w["w"][0]) - (1.0 + d1 + d2 + d3)) < 1e-15


# ---------------------------------------------------------------------------------------------------------------
# 6. Product functions that never met a golden in round 1 (VERDICT a1, a17).
def test_product_get_anchors_equals_the_reference_table(golden):
    from faster_rcnn_amd import util
    assert np.array_equal(util.get_anchors([128, 256, 512]), golden["anchors9"])
    assert np.array_equal(util.get_anchors([16, 32, 64, 128, 256, 512]), golden["anchors18"])
    assert util.get_anchors([128, 256, 512]).tolist() == [[128, 128], [90, 181], [181, 90], [256, 256], [181, 362], [362, 181],
                                                          [512, 512], [362, 724], [724, 362]]             # SURVEY 8(a) a1, util.py:249-253


def test_product_host_preprocess_is_bgr_minus_mean():
    """resnet.preprocess / vgg.preprocess (resnet.py:64-75, vgg.py:52-57): BGR->RGB, then Keras preprocess_input flips
    back to BGR and subtracts [103.939, 116.779, 123.68]: net effect BGR - mean in float64."""
    from faster_rcnn_amd import resnet, vgg
    img = np.array([[[0, 128, 255], [10, 20, 30]], [[255, 0, 1], [103, 116, 123]]], dtype=np.uint8)
    want = np.array([[[-103.939, 11.221, 131.32], [-93.939, -96.779, -93.68]],
                     [[151.061, -116.779, -122.68], [-0.939, -0.779, -0.68]]], dtype=np.float64)
    for fn in (resnet.preprocess, vgg.preprocess):
        got = np.asarray(fn(img))
        assert got.dtype == np.float64 and got.shape == (2, 2, 3)
        assert np.abs(got - want).max() < 1e-12


# ---------------------------------------------------------------------------------------------------------------
# 7. Kernel ORIENTATION (the all-ones kernels above cannot tell a convolution from a cross-correlation, a
#    transposed (r, t) or swapped (in, out) axes).  Keras Conv2D with a (kh, kw, in, out) kernel -> tf.nn.conv2d (NHWC,
#    HWIO) is a CROSS-CORRELATION: y[h, w, o] = sum_{r,t,i} x[h*s + r - p, w*s + t - q, i] * k[r, t, i, o]
#    (resnet.py:150, 229, 408; vgg.py:96).  Asymmetric everything: 5x6 input with 2 channels, 3x3x2x2 kernel with 36
#    distinct entries, stride 2 SAME -> rows pad (1, 1), columns pad (0, 1) (pad_along 2 and 1), output 3x3x2.
#        x[h, w, i]    = 1 + 12 h + 2 w + i                                (1 .. 60)
#        k[r, t, i, 0] = 1 + 2 ((3 r + t) 2 + i)                          (odd numbers 1 .. 35)
#        k[r, t, i, 1] = -(2 + 2 ((3 r + t) 2 + i)),  except k[1, 2, 0, 1] = 7
#    y[0, 0, 0] worked by hand: rows -1, 0, 1 -> taps r = 1, 2 on image rows 0, 1; columns 0, 1, 2 -> t = 0, 1, 2:
#        r=1: (1, 2).(13, 15) + (3, 4).(17, 19) + (5, 6).(21, 23) = 43 + 127 + 243
#        r=2: (13, 14).(25, 27) + (15, 16).(29, 31) + (17, 18).(33, 35) = 703 + 931 + 1191        sum = 3238
#    The other 17 literals come from the same formula evaluated with explicit Python loops (below, not the code under test).
#    What the plausible WRONG orientations give at y[0, 0] / y[1, 1]:
#        true                      [3238, -3207] / [12039, -11649]
#        kernel flipped (true convolution)      [878, -963] / [8391, -8117]
#        (r, t) transposed                      [2622, -2301] / [11079, -10399]
#        (in, out) swapped                      [-24, -210] / [545, -459]
#        column padding (1, 0) instead of (0, 1)  y[0, 0] = [2076, -2057]
ORIENT_WANT = [[[3238, -3207], [4390, -4291], [3212, -3344]],
               [[10743, -10397], [12039, -11649], [7806, -8220]],
               [[7054, -6063], [7630, -6571], [4508, -4928]]]
ORIENT_WRONG = {"flipped": ([878, -963], [8391, -8117]), "rt_transposed": ([2622, -2301], [11079, -10399]),
                "io_swapped": ([-24, -210], [545, -459])}


def _orient_case():
    x = np.zeros((5, 6, 2), np.float32)
    k = np.zeros((3, 3, 2, 2), np.float32)
    for h in range(5):
        for w in range(6):
            for i in range(2):
                x[h, w, i] = 1 + 12 * h + 2 * w + i
    for r in range(3):
        for t in range(3):
            for i in range(2):
                k[r, t, i, 0] = 1 + 2 * ((3 * r + t) * 2 + i)
                k[r, t, i, 1] = -(2 + 2 * ((3 * r + t) * 2 + i))
    k[1, 2, 0, 1] = 7
    return x, k


def _xcorr_loops(x, k, stride, pad_top, pad_left, ho, wo):
    y = np.zeros((ho, wo, k.shape[3]), np.float64)
    for h in range(ho):
        for w in range(wo):
            for o in range(k.shape[3]):
                for r in range(k.shape[0]):
                    for t in range(k.shape[1]):
                        hh, ww = h * stride + r - pad_top, w * stride + t - pad_left
                        if 0 <= hh < x.shape[0] and 0 <= ww < x.shape[1]:
                            y[h, w, o] += float(np.dot(x[hh, ww].astype(np.float64), k[r, t, :, o].astype(np.float64)))
    return y


def test_oracle_conv_orientation_asymmetric_kernel():
    x, k = _orient_case()
    want = np.array(ORIENT_WANT, np.float64)
    assert want[0, 0, 0] == 43 + 127 + 243 + 703 + 931 + 1191                    # the element worked out above
    assert np.array_equal(_xcorr_loops(x, k, 2, 1, 0, 3, 3), want)              # the literals follow the stated formula
    y = np.asarray(kr.conv2d(x[None], k, None, 2, "same", torch.float64))[0]
    assert np.array_equal(y, want)
    # and each wrong orientation is a DIFFERENT answer on this case (so agreement above means something)
    for name, kk in (("flipped", k[::-1, ::-1]), ("rt_transposed", k.transpose(1, 0, 2, 3)), ("io_swapped", k.transpose(0, 1, 3, 2))):
        bad = _xcorr_loops(x, np.ascontiguousarray(kk), 2, 1, 0, 3, 3)
        assert bad[0, 0].tolist() == ORIENT_WRONG[name][0] and bad[1, 1].tolist() == ORIENT_WRONG[name][1]
        assert not np.array_equal(bad, want)


# 8. Dense after the VGG head's Flatten (vgg.py:233-247: TimeDistributed(Flatten()) on the (7, 7, C) crop, channels last
#    -> flat index j = (h*7 + w)*C + c, then Dense kernels are (in, out): y = x @ K + b).  A 7x7 RoI makes RoiResizeConv
#    the identity (scale 1), so crop[h, w, c] = feat[y1 + h, x1 + w, c] with feat[y, x, c] = 100 y + 10 x + c, roi
#    (x1, y1, x2, y2) = (2, 1, 9, 8): crop[h, w, c] = 100 (1 + h) + 10 (2 + w) + c.  fc1 column 0 is one-hot at
#    (h, w, c) = (2, 5, 1) -> j = 77 -> 371; column 1 at (6, 0, 3) -> j = 171 -> 723.
#    A channels-first flatten (j = c*49 + h*7 + w) would read crop[4, 0, 1] = 521 and crop[3, 3, 3] = 453; a (w, h)
#    transposed one crop[5, 2, 1] = 641; a Dense kernel used as (out, in) does not even have the right shape.
#    fc2 = identity on the first two features; dense_reg_2 = [x0, x1, x0 - x1, 2 x0]; dense_class_2 logits = x / 128.
FLAT_REG_WANT = [371.0, 723.0, -352.0, 742.0]


def _flatten_case():
    C = 4
    feat = np.zeros((1, 9, 10, C), np.float32)
    for y in range(9):
        for xx in range(10):
            for c in range(C):
                feat[0, y, xx, c] = 100 * y + 10 * xx + c
    rois = np.array([[2, 1, 9, 8]], np.float32)
    fc1 = np.zeros((49 * C, 4), np.float32)
    fc1[(2 * 7 + 5) * C + 1, 0] = 1.0
    fc1[(6 * 7 + 0) * C + 3, 1] = 1.0
    fc2 = np.zeros((4, 4), np.float32)
    fc2[0, 0] = fc2[1, 1] = 1.0
    kreg = np.array([[1, 0, 1, 2], [0, 1, -1, 0], [0, 0, 0, 0], [0, 0, 0, 0]], np.float32)
    kcls = np.array([[1 / 128, 0], [0, 1 / 128], [0, 0], [0, 0]], np.float32)      # exact in f32
    z = lambda n: np.zeros(n, np.float32)
    w = {"fc1": [fc1, z(4)], "fc2": [fc2, z(4)], "dense_class_2": [kcls, z(2)], "dense_reg_2": [kreg, z(4)]}
    p1 = 1.0 / (1.0 + math.exp((371 - 723) / 128))
    return feat, rois, w, [1.0 - p1, p1]


def test_oracle_dense_after_hwc_flatten():
    feat, rois, w, cls_want = _flatten_case()
    cls, reg = kr.KerasGraphs(w, torch.float64).vgg_classifier(torch.as_tensor(feat), rois, 2)
    assert np.array_equal(np.asarray(reg)[0], FLAT_REG_WANT)
    assert np.abs(np.asarray(cls)[0] - cls_want).max() < 1e-12


# ---------------------------------------------------------------------------------------------------------------
# GPU twins: the same hand-derived answers through the C ABI.
gpu = pytest.mark.gpu


@gpu
@pytest.mark.parametrize("name", sorted(SAME_CASES))
def test_gpu_same_padding(name):
    from faster_rcnn_amd import ops
    n, k, stride, want = _same_expected(name)
    x = torch.arange(n * n, dtype=torch.float32).reshape(1, n, n, 1).cuda()
    pc = ops.PackedConv(np.ones((k, k, 1, 1), np.float32))
    y = ops.conv2d(x, pc, stride, "same")
    assert np.array_equal(y.cpu().numpy()[0, :, :, 0], want)
    # and on the 32-channel path (the MFMA main loop with halo handling): every channel carries the image / 32
    x32 = (torch.arange(n * n, dtype=torch.float32).reshape(1, n, n, 1) * torch.ones(32)).cuda().contiguous()
    w32 = np.zeros((k, k, 32, 4), np.float32)
    w32[:, :, :, 0] = 1.0 / 32
    w32[:, :, 0, 1] = 1.0
    y = ops.conv2d(x32, ops.PackedConv(w32), stride, "same").cpu().numpy()[0]
    assert np.array_equal(y[:, :, 0], want) and np.array_equal(y[:, :, 1], want) and not y[:, :, 2:].any()


@gpu
def test_gpu_valid_stride2_and_pools():
    from faster_rcnn_amd import ops
    x = (torch.arange(49, dtype=torch.float32).reshape(1, 7, 7, 1) * torch.ones(32)).cuda().contiguous()
    w = np.zeros((1, 1, 32, 4), np.float32)
    w[0, 0, 0, 0] = 1.0
    y = ops.conv2d(x, ops.PackedConv(w), 2, "valid").cpu().numpy()[0, :, :, 0]
    assert np.array_equal(y, [[0, 2, 4, 6], [14, 16, 18, 20], [28, 30, 32, 34], [42, 44, 46, 48]])
    x4 = (torch.arange(49, dtype=torch.float32).reshape(1, 7, 7, 1) * torch.ones(4)).cuda().contiguous()
    assert np.array_equal(ops.pool2d(x4, 3, 2, True).cpu().numpy()[0, :, :, 0], [[16, 18, 20], [30, 32, 34], [44, 46, 48]])
    assert np.array_equal(ops.pool2d(x4, 2, 2, True).cpu().numpy()[0, :, :, 0], [[8, 10, 12], [22, 24, 26], [36, 38, 40]])


@gpu
def test_gpu_legacy_bilinear_3x5_to_7x7():
    from faster_rcnn_amd import ops
    feat, rois, want = _roi_case()
    f4 = np.concatenate([feat, feat], axis=2)                     # the kernel moves channels four at a time
    got = ops.roi_crop_resize(torch.from_numpy(f4).cuda(), torch.from_numpy(rois).cuda(), 7).cpu().numpy()
    assert np.abs(got[..., :2] - want).max() < 2e-5 and np.array_equal(got[..., :2], got[..., 2:])
    assert np.array_equal(got, kr.roi_resize(f4, rois, 7))        # and bit-for-bit the oracle's f32 lerp order


@gpu
def test_gpu_tf_published_resize_up_vector():
    """The third-party vector through frcnn_roi_crop_resize_fwd.  The entry point resamples to a SQUARE grid (the reference
    only asks for 7x7), so the 3x2 -> 6x4 resize is made of two calls, each the identity along one axis (in == out there:
    scale 1, every lerp weight 0): 2 -> 4 columns on a 4-row strip, then 3 -> 6 rows on a 6-column strip.  That is TF's own
    order -- x lerp inside each source row, then the y lerp between the two results -- and every value in this vector is
    exact in f32, so the literals must come out bit for bit."""
    from faster_rcnn_amd import ops
    src = np.array(TF_RESIZE_UP_IN, np.float32).reshape(3, 2)
    strip = np.zeros((4, 2, 4), np.float32)                         # (rows, cols, channels): a 4th row of padding, channel 0 carries the image
    strip[:3, :, 0] = src
    wide = ops.roi_crop_resize(torch.from_numpy(strip).cuda(), torch.tensor([[0., 0., 2., 4.]]).cuda(), 4).cpu().numpy()[0, :3, :, 0]     # (3,4)
    strip2 = np.zeros((3, 6, 4), np.float32)
    strip2[:, :4, 0] = wide
    got = ops.roi_crop_resize(torch.from_numpy(strip2).cuda(), torch.tensor([[0., 0., 6., 3.]]).cuda(), 6).cpu().numpy()[0, :, :4, 0]     # (6,4)
    assert np.array_equal(got.reshape(-1), np.array(TF_RESIZE_UP_BILINEAR, np.float32))


@gpu
def test_gpu_batchnorm_fold_both_epsilons():
    """The product folds BN (+Scale) into the conv epilogue's scale/shift (nets.ConvUnit.lower): a 1x1 identity conv
    on x = 2 must return gamma*(x-mean)/sqrt(var+eps)+beta for both epsilons."""
    from faster_rcnn_amd import nets
    x = torch.full((1, 2, 2, 32), 2.0, dtype=torch.float32).cuda()
    for eps in (nets.BN_EPS_STEM, nets.BN_EPS_BLOCK):
        w = {"c": [np.eye(32, dtype=np.float32).reshape(1, 1, 32, 32), np.zeros(32, np.float32)],
             "bn": [np.full(32, 1.5), np.full(32, 0.25), np.full(32, 0.5), np.full(32, 0.25)]}
        y = nets.ConvUnit(w, "c", "bn", eps=eps)(x).cpu().numpy()
        want = 1.5 * 1.5 / math.sqrt(0.25 + eps) + 0.25
        assert np.abs(y - want).max() < 5e-7 * want
    assert (nets.BN_EPS_STEM, nets.BN_EPS_BLOCK) == (1e-3, 1e-5)


def _loss_call(name, y_true, y_pred, *dims):
    """frcnn_loss_* -> (loss, gradient) for small hand-sized inputs."""
    from faster_rcnn_amd import _lib
    from faster_rcnn_amd.ops import _p, _stream
    yt = torch.as_tensor(np.asarray(y_true, np.float32)).cuda().contiguous()
    yp = torch.as_tensor(np.asarray(y_pred, np.float32)).cuda().contiguous()
    loss = torch.zeros(1, dtype=torch.float32, device="cuda")
    g = torch.zeros_like(yp)
    if name in ("frcnn_loss_det_cls", "frcnn_loss_det_reg"):
        _lib.call(name, _p(yt), _p(yp), *dims, _p(loss), _p(g), g.shape[1], _stream())
    elif name.endswith("_ws"):
        ws = torch.empty(int(_lib.load().frcnn_loss_workspace_bytes()), dtype=torch.uint8, device="cuda")
        _lib.call(name, _p(yt), _p(yp), *dims, _p(loss), _p(g), _p(ws), _stream())
    else:
        _lib.call(name, _p(yt), _p(yp), *dims, _p(loss), _p(g), _stream())
    return float(loss.item()), g.cpu().numpy()


@gpu
def test_gpu_losses_known_answers():
    # cls_loss_rpn: cells x 2A / cells x A
    l, g = _loss_call("frcnn_loss_rpn_cls", [[1, 1], [1, 0], [0, 1], [1, 1], [1, 1]], [[0.5], [0.25], [0.9], [0.0], [1.0]], 5, 1)
    want = (LN2 - math.log(0.75) - math.log(1e-7) - math.log(1.0 - 1e-7)) / 256.0
    assert abs(l - want) < 1e-6 * want
    assert g[2, 0] == 0.0                                         # unselected anchor: no gradient
    # bbreg_loss_rpn with the mask outside the sum
    l, g = _loss_call("frcnn_loss_rpn_reg", [[1, 1, 1, 1, 0.5, -2.0, 0.0, 1.0], [0, 0, 0, 0, 0, 0, 0, 0]], [[0, 0, 0, 0], [3.0, 0, 0, 0]], 2, 1)
    assert abs(l - 0.009635416666666667) < 1e-8
    assert g[1, 0] != 0.0                                         # the unselected anchor's error IS in the sum, so it has a gradient
    # the many-workgroup forms the training step calls: same known answers, and on a full-size RPN map (2 394 cells x 9
    # anchors) the one-workgroup form's loss to 1e-6 relative and its gradient bit for bit
    l, g = _loss_call("frcnn_loss_rpn_cls_ws", [[1, 1], [1, 0], [0, 1], [1, 1], [1, 1]], [[0.5], [0.25], [0.9], [0.0], [1.0]], 5, 1)
    assert abs(l - want) < 1e-6 * want and g[2, 0] == 0.0
    l, g = _loss_call("frcnn_loss_rpn_reg_ws", [[1, 1, 1, 1, 0.5, -2.0, 0.0, 1.0], [0, 0, 0, 0, 0, 0, 0, 0]], [[0, 0, 0, 0], [3.0, 0, 0, 0]], 2, 1)
    assert abs(l - 0.009635416666666667) < 1e-8 and g[1, 0] != 0.0
    rs = np.random.RandomState(0)
    cells, A = 2394, 9
    yc = np.concatenate([rs.rand(cells, A) < 0.05, rs.rand(cells, A) < 0.02], axis=1).astype(np.float32)
    pc = rs.rand(cells, A).astype(np.float32)
    pc[rs.rand(cells, A) < 0.01] = 0.0                          # clipped probabilities
    yr = np.concatenate([np.repeat(rs.rand(cells, A) < 0.02, 4, axis=1), rs.randn(cells, 4 * A)], axis=1).astype(np.float32)
    pr = rs.randn(cells, 4 * A).astype(np.float32)
    for name, yt, yp in (("frcnn_loss_rpn_cls", yc, pc), ("frcnn_loss_rpn_reg", yr, pr)):
        l1, g1 = _loss_call(name, yt, yp, cells, A)
        l2, g2 = _loss_call(name + "_ws", yt, yp, cells, A)
        assert abs(l1 - l2) <= 1e-6 * abs(l1) and np.array_equal(g1, g2), name
        assert (l2, g2.tobytes()) == (lambda r: (r[0], r[1].tobytes()))(_loss_call(name + "_ws", yt, yp, cells, A))      # reproducible
    # detector losses
    l, _ = _loss_call("frcnn_loss_det_reg", [[1, 1, 1, 1, 1.0, 1.0, 1.0, 1.0], [0, 0, 0, 0, 9.0, 9.0, 9.0, 9.0]], [[0, 0, 0, 3.0], [5.0, 5.0, 5.0, 5.0]], 2, 1)
    assert abs(l - 0.7498500299940012) < 1e-6
    l, _ = _loss_call("frcnn_loss_det_cls", [[1.0, 0.0], [0.0, 1.0], [1.0, 0.0]], [[0.25, 0.75], [0.0, 1.0], [0.25, 0.75]], 3, 2)
    assert abs(l - (-2 * math.log(0.25) - math.log(1.0 - 1e-7)) / 3.0) < 1e-6


@gpu
def test_gpu_optimiser_known_answers():
    from faster_rcnn_amd import train
    from faster_rcnn_amd.ops import _p, _stream
    from faster_rcnn_amd import _lib
    w = torch.ones(8, dtype=torch.float32, device="cuda")
    g = torch.full((8,), 0.5, dtype=torch.float32, device="cuda")
    v = torch.zeros(8, dtype=torch.float32, device="cuda")
    for want in (0.95, 0.855):
        _lib.call("frcnn_sgd_momentum", _p(w), _p(g), _p(v), 8, 0.1, 0.9, 0.0, 1.0, _stream())
        assert abs(float(w[0]) - want) < 1e-6
    # L2 inside the optimiser: g_eff = g + 2*l2*w  (Keras adds l2*sum(w^2) to the loss, resnet.py:26-27)
    w.fill_(1.0); v.zero_()
    _lib.call("frcnn_sgd_momentum", _p(w), _p(g), _p(v), 8, 0.1, 0.9, 0.25, 1.0, _stream())
    assert abs(float(w[0]) - (1.0 - 0.1 * (0.5 + 2 * 0.25 * 1.0))) < 1e-6
    # Adam: steps t = 1, 2, then a re-compile (fresh moments) with t = 3
    lr, gv = 1e-3, 0.5
    w.fill_(1.0)
    m = torch.zeros(8, dtype=torch.float32, device="cuda"); vv = torch.zeros(8, dtype=torch.float32, device="cuda")
    d1, m1, v1 = _adam_update(gv, lr, 1)
    d2, _, _ = _adam_update(gv, lr, 2, m1, v1)
    d3, _, _ = _adam_update(gv, lr, 3)
    for t, want in ((1, 1 + d1), (2, 1 + d1 + d2)):
        _lib.call("frcnn_adam", _p(w), _p(g), _p(m), _p(vv), 8, lr, 0.9, 0.999, 1e-8, t, 0.0, 1.0, _stream())
        assert abs(float(w[0]) - want) < 2e-7
    m.zero_(); vv.zero_()
    _lib.call("frcnn_adam", _p(w), _p(g), _p(m), _p(vv), 8, lr, 0.9, 0.999, 1e-8, 3, 0.0, 1.0, _stream())
    assert abs(float(w[0]) - (1 + d1 + d2 + d3)) < 3e-7
    # the trainer's counter lives on the optimiser object and survives compile() (ParamSet.step)
    opt = train.Adam(lr=lr)
    ps = train.ParamSet({"a": [np.ones(8, np.float32)]}, ["a"])
    for phase in range(2):
        ps.reset_optimizer()
        ps.g.fill_(gv)
        ps.step(opt, 0.0)
        ps.g.fill_(gv)
        ps.step(opt, 0.0)
    d3b, m3, v3 = _adam_update(gv, lr, 3)
    d4b, _, _ = _adam_update(gv, lr, 4, m3, v3)
    assert opt.iterations == 4
    assert abs(float(ps.w[0]) - (1 + d1 + d2 + d3b + d4b)) < 4e-7


@gpu
def test_gpu_conv_orientation_asymmetric_kernel():
    """The hand-derived orientation vectors through frcnn_pack_conv_weights + frcnn_conv2d_fwd: once on the 2-channel
    shape itself (per-element k decode) and once embedded in 32 input / 4 output channels (the MFMA main loop with its
    [channel chunk][tap][32 channels] filter packing), the unused input channels carrying non-zero data under zero weights."""
    from faster_rcnn_amd import ops
    x, k = _orient_case()
    want = np.array(ORIENT_WANT, np.float32)
    y = ops.conv2d(torch.from_numpy(x[None]).cuda(), ops.PackedConv(k), 2, "same").cpu().numpy()[0]
    assert np.array_equal(y, want)
    x32 = np.full((1, 5, 6, 32), 7.0, np.float32)
    x32[0, :, :, 5], x32[0, :, :, 20] = x[:, :, 0], x[:, :, 1]
    k32 = np.zeros((3, 3, 32, 4), np.float32)
    k32[:, :, 5, 1], k32[:, :, 20, 1] = k[:, :, 0, 0], k[:, :, 1, 0]
    k32[:, :, 5, 3], k32[:, :, 20, 3] = k[:, :, 0, 1], k[:, :, 1, 1]
    y = ops.conv2d(torch.from_numpy(x32).cuda(), ops.PackedConv(k32), 2, "same").cpu().numpy()[0]
    assert np.array_equal(y[:, :, 1], want[:, :, 0]) and np.array_equal(y[:, :, 3], want[:, :, 1]) and not y[:, :, (0, 2)].any()
    # the bf16 engine (64-channel chunks): small integers are exact in bf16 operands with f32 accumulation
    x64 = np.full((1, 5, 6, 64), 3.0, np.float32)
    x64[0, :, :, 9], x64[0, :, :, 40] = x[:, :, 0], x[:, :, 1]
    k64 = np.zeros((3, 3, 64, 64), np.float32)
    k64[:, :, 9, 2], k64[:, :, 40, 2] = k[:, :, 0, 0], k[:, :, 1, 0]
    k64[:, :, 9, 33], k64[:, :, 40, 33] = k[:, :, 0, 1], k[:, :, 1, 1]
    yb = ops.conv2d_bf16(torch.from_numpy(x64).cuda().to(torch.bfloat16), ops.PackedConvBf16(k64), 2, "same", out_f32=True).cpu().numpy()[0]
    assert np.array_equal(yb[:, :, 2], want[:, :, 0]) and np.array_equal(yb[:, :, 33], want[:, :, 1])


@gpu
def test_gpu_dense_after_hwc_flatten():
    from faster_rcnn_amd import nets
    feat, rois, w, cls_want = _flatten_case()
    cls, reg = nets.VggHead(w, 2)(torch.from_numpy(feat).cuda(), torch.from_numpy(rois).cuda())
    assert np.array_equal(reg.cpu().numpy()[0], FLAT_REG_WANT)
    assert np.abs(cls.cpu().numpy()[0].astype(np.float64) - cls_want).max() < 1e-6
