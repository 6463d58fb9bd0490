"""Weight containers and initialisers, keyed by the reference's Keras layer names.

A weight set is ``{layer_name: [arrays in Keras get_weights() order]}``:
    Conv2D / Dense        [kernel (HWIO / IO), bias]   (no bias entry when use_bias=False)
    BatchNormalization    [gamma, beta, moving_mean, moving_variance]
    Scale                 [gamma, beta]                 (custom_layers.py:59-134)
Layer names follow resnet.py:145-147, 408-410 / vgg.py:96-137 so a converted Keras .h5
(f1 row in SURVEY 8(f)) drops in unchanged.

The reference initialises from downloaded ImageNet weights (resnet.py:481-485); there is no
network here, so ``synthetic_*`` draws seeded weights instead: He-normal conv
kernels keep activations O(1) through 50+ layers, BN statistics are non-trivial so folding
errors would show, and the new layers use the reference's own initialisers.
"""
import numpy as np

f32 = np.float32


def _trunc_normal(rs, shape, std):
    v = rs.randn(*shape)
    bad = np.abs(v) > 2
    while bad.any():
        v[bad] = rs.randn(int(bad.sum()))
        bad = np.abs(v) > 2
    return (v * std).astype(f32)


def _conv(rs, k, cin, cout, bias=True, std=None):
    std = np.sqrt(2.0 / (k * k * cin)) if std is None else std
    out = [(rs.randn(k, k, cin, cout) * std).astype(f32)]
    if bias:
        out.append((rs.randn(cout) * 0.01).astype(f32))
    return out


def _bn(rs, c, gamma=1.0):
    return [(gamma * (1 + 0.05 * rs.randn(c))).astype(f32), (0.05 * rs.randn(c)).astype(f32),
            (0.1 * rs.randn(c)).astype(f32), (1 + 0.1 * rs.rand(c)).astype(f32)]


def _scale(rs, c):
    return [(1 + 0.05 * rs.randn(c)).astype(f32), (0.05 * rs.randn(c)).astype(f32)]


def resnet_block_names(depth):
    """(stage, block label, is_conv_block) for stages 2-4 (resnet.py:414-446 / 570-598)."""
    if depth == 50:
        plan = {2: "abc", 3: "abcd", 4: "abcdef"}
        return [(s, b, b == "a") for s in (2, 3, 4) for b in plan[s]]
    plan = {2: ["a", "b", "c"], 3: ["a"] + ["b%d" % i for i in range(1, 4)], 4: ["a"] + ["b%d" % i for i in range(1, 23)]}
    return [(s, b, b == "a") for s in (2, 3, 4) for b in plan[s]]


STAGE_FILTERS = {2: (64, 64, 256), 3: (128, 128, 512), 4: (256, 256, 1024), 5: (512, 512, 2048)}


def _bottleneck(rs, w, stage, block, cin, has_shortcut, bias, scale):
    f1, f2, f3 = STAGE_FILTERS[stage]
    for suffix, k, ci, co in (("2a", 1, cin, f1), ("2b", 3, f1, f2), ("2c", 1, f2, f3)) + ((("1", 1, cin, f3),) if has_shortcut else ()):
        w["res%d%s_branch%s" % (stage, block, suffix)] = _conv(rs, k, ci, co, bias)
        # small gamma on the residual branch's last BN keeps the activation scale O(1) through
        # 16-33 residual adds (otherwise the variance doubles per block)
        w["bn%d%s_branch%s" % (stage, block, suffix)] = _bn(rs, co, 0.25 if suffix == "2c" else 1.0)
        if scale:
            w["scale%d%s_branch%s" % (stage, block, suffix)] = _scale(rs, co)
    return f3


def synthetic_resnet(depth=50, anchors_per_loc=9, num_classes=21, seed=1):
    """Base (conv1..stage 4) + RPN heads + stage-5 classifier, ResNet-50 or -101."""
    assert depth in (50, 101)
    rs = np.random.RandomState(seed)
    r101 = depth == 101
    # conv1 sees mean-subtracted pixels (rms ~70): scale its kernel so the stem output is O(1)
    w = {"conv1": _conv(rs, 7, 3, 64, bias=not r101, std=np.sqrt(2.0 / 147) / 70.0), "bn_conv1": _bn(rs, 64)}
    if r101:
        w["scale_conv1"] = _scale(rs, 64)
    cin = 64
    for stage, block, is_conv in resnet_block_names(depth):
        cin = _bottleneck(rs, w, stage, block, cin, is_conv, bias=not r101, scale=r101)
    # RPN (resnet.py:464-474).  The reference draws N(0,.05) / TruncatedNormal(.01) and then
    # TRAINS them; untrained they give near-constant scores and zero deltas, which would make
    # the proposal/NMS stage trivial.  The synthetic set widens them so that decoded boxes move
    # by up to ~half their size and scores spread over (0,1).
    w["rpn_conv1"] = [(rs.randn(3, 3, 1024, 512) * np.sqrt(2.0 / (9 * 1024))).astype(f32), (rs.randn(512) * 0.01).astype(f32)]
    w["rpn_out_cls"] = [_trunc_normal(rs, (1, 1, 512, anchors_per_loc), 0.02), (rs.randn(anchors_per_loc) * 0.1).astype(f32)]
    w["rpn_out_bbreg"] = [_trunc_normal(rs, (1, 1, 512, 4 * anchors_per_loc), 0.03), (rs.randn(4 * anchors_per_loc) * 0.1).astype(f32)]
    # classifier (resnet.py:508-533)
    cin = 1024
    for block in "abc":
        cin = _bottleneck(rs, w, 5, block, cin, block == "a", bias=not r101, scale=r101)
    w["dense_class_%d" % num_classes] = [_trunc_normal(rs, (2048, num_classes), 0.03), (rs.randn(num_classes) * 0.1).astype(f32)]
    w["dense_reg_%d" % num_classes] = [_trunc_normal(rs, (2048, 4 * (num_classes - 1)), 0.02), (rs.randn(4 * (num_classes - 1)) * 0.1).astype(f32)]
    return w


def calibrate_classifier(weights, num_classes, probs, spread=0.5):
    """Synthetic-weight helper: make an UNTRAINED ``dense_class_C`` layer fire many classes.

    Random-init heads are degenerate: the pooled stage-5 features of all RoIs share a large common component, so one class
    wins every arg-max and detection-set comparisons (mAP deltas, per-class NMS) exercise a single class.  Given the class
    probabilities ``probs`` (n, C) the CURRENT weights produce on a calibration image, the logits are known up to a per-row
    constant (log p); this returns new ``[kernel, bias]`` with each class's logit centred over the RoIs and the layer scaled
    so that a class's logit varies by ``spread`` (standard deviation, averaged over classes) from RoI to RoI:
    W' = g W,  b'_c = g (b_c - mean_r L_rc).  A linear re-parametrisation of the same layer -- nothing about the kernels
    under test changes.  ``spread`` is kept small on purpose: which class wins does not depend on it, the gain (~3 on the
    seeded ResNet-50 set) multiplies the f32 noise of the pooled features as little as possible, and a frame unlike the
    calibration frame (a photograph against uniform noise) shifts every RoI's logits together by gain x offset -- a large
    gain saturates its softmax at exactly 1.0 for all RoIs, whose tied scores then leave the NMS order undefined."""
    name = "dense_class_%d" % num_classes
    kernel, bias = (np.asarray(a, dtype=np.float64) for a in weights[name])
    logits = np.log(np.clip(np.asarray(probs, dtype=np.float64), 1e-30, None))
    logits -= logits.mean(axis=1, keepdims=True)                  # drop the per-row softmax constant
    centre = logits.mean(axis=0)
    gain = spread / max(float((logits - centre).std(axis=0).mean()), 1e-6)
    return [(kernel * gain).astype(f32), ((bias - centre) * gain).astype(f32)]


VGG_CONVS = [("block1_conv1", 3, 64), ("block1_conv2", 64, 64), ("block2_conv1", 64, 128), ("block2_conv2", 128, 128),
             ("block3_conv1", 128, 256), ("block3_conv2", 256, 256), ("block3_conv3", 256, 256),
             ("block4_conv1", 256, 512), ("block4_conv2", 512, 512), ("block4_conv3", 512, 512),
             ("block5_conv1", 512, 512), ("block5_conv2", 512, 512), ("block5_conv3", 512, 512)]


def synthetic_vgg16(anchors_per_loc=9, num_classes=21, seed=1, with_classifier=True):
    rs = np.random.RandomState(seed)
    w = {}
    for name, ci, co in VGG_CONVS:
        w[name] = _conv(rs, 3, ci, co, std=(np.sqrt(2.0 / 27) / 70.0) if ci == 3 else None)
    w["rpn_conv1"] = [(rs.randn(3, 3, 512, 512) * np.sqrt(2.0 / (9 * 512))).astype(f32), (rs.randn(512) * 0.01).astype(f32)]   # vgg.py:171-174
    w["rpn_out_cls"] = [_trunc_normal(rs, (1, 1, 512, anchors_per_loc), 0.02), (rs.randn(anchors_per_loc) * 0.1).astype(f32)]
    w["rpn_out_bbreg"] = [_trunc_normal(rs, (1, 1, 512, 4 * anchors_per_loc), 0.03), (rs.randn(4 * anchors_per_loc) * 0.1).astype(f32)]
    if with_classifier:
        w["fc1"] = [(rs.randn(7 * 7 * 512, 4096) * np.sqrt(2.0 / (7 * 7 * 512))).astype(f32), np.zeros(4096, f32)]
        w["fc2"] = [(rs.randn(4096, 4096) * np.sqrt(2.0 / 4096)).astype(f32), np.zeros(4096, f32)]
        w["dense_class_%d" % num_classes] = [_trunc_normal(rs, (4096, num_classes), 0.02), (rs.randn(num_classes) * 0.1).astype(f32)]
        w["dense_reg_%d" % num_classes] = [_trunc_normal(rs, (4096, 4 * (num_classes - 1)), 0.01), (rs.randn(4 * (num_classes - 1)) * 0.1).astype(f32)]
    return w


def save_weights_file(path, weights, full_model=False):
    """``.h5`` / ``.hdf5`` -> a Keras 2.0.x HDF5 file Keras itself can load (h5lite writer); anything else -> .npz."""
    if str(path).lower().endswith((".h5", ".hdf5")):
        from . import h5lite
        h5lite.write_keras_weights(path, {k: [np.asarray(a, dtype=np.float32) for a in v] for k, v in weights.items()}, full_model=full_model)
    else:
        save_npz(path, weights)


def save_npz(path, weights):
    flat = {}
    for name, arrs in weights.items():
        for i, a in enumerate(arrs):
            flat["%s/%d" % (name, i)] = a
    np.savez(path, **flat)


def load_weights_file(path):
    """{layer_name: [arrays in get_weights() order]} from either a Keras 2.0.x ``.h5`` (save_weights or a full
    model.save file, read in-process by h5lite) or the ``.npz`` this package's save_weights writes."""
    from . import h5lite
    if h5lite.is_hdf5(path):
        return h5lite.read_keras_weights(path)
    return _load_npz(path)


def load_npz(path):
    return load_weights_file(path)


def _load_npz(path):
    z = np.load(path)
    out = {}
    for key in z.files:
        name, i = key.rsplit("/", 1)
        out.setdefault(name, {})[int(i)] = z[key]
    return {n: [d[i] for i in sorted(d)] for n, d in out.items()}
