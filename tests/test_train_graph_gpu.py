"""The training step replayed from hipGraphs (train._StepGraph) IS the eager step: same kernels, same arguments, same order per
tensor.  Every test runs one sequence of steps twice -- train.STEP_GRAPHS off and on -- and compares losses and weights bit for bit.
(Reference: one train_on_batch per image, train_util.py:50-54, 110-118.)"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")



def image(h, w, seed=0):
    rs = np.random.RandomState(seed)
    return (rs.randint(0, 256, (h, w, 3)).astype(np.float64) - np.array([103.939, 116.779, 123.68]))[None]


def rpn_targets(rows, cols, A, seed=1):
    rs = np.random.RandomState(seed)
    can_use = rs.rand(1, rows, cols, A) < 0.25
    is_pos = rs.rand(1, rows, cols, A) < 0.15
    y_class = np.concatenate([can_use, is_pos], axis=3)
    sel = np.repeat(can_use & is_pos, 4, axis=3).astype(np.float32)
    tg = (rs.randn(1, rows, cols, 4 * A) * is_pos.repeat(4, axis=3)).astype(np.float32)
    return y_class, np.concatenate([sel, tg], axis=3)


def _copy(w):
    return {k: [a.copy() for a in v] for k, v in w.items()}


class graphs:
    def __init__(self, on, after=2, shapes=8):
        self.v = (on, after, shapes)

    def __enter__(self):
        from faster_rcnn_amd import train
        self.prev = (train.STEP_GRAPHS, train.STEP_GRAPH_AFTER, train.STEP_GRAPH_SHAPES)
        train.STEP_GRAPHS, train.STEP_GRAPH_AFTER, train.STEP_GRAPH_SHAPES = self.v

    def __exit__(self, *exc):
        from faster_rcnn_amd import train
        train.STEP_GRAPHS, train.STEP_GRAPH_AFTER, train.STEP_GRAPH_SHAPES = self.prev


def _captured(model):
    return len(model._trainer._graphs["graphs"])


@pytest.mark.parametrize("dtype,opt_kind", [("f32", "sgd"), ("bf16", "sgd"), ("f32", "adam")])
def test_rpn_steps_replayed_equal_eager_steps(dtype, opt_kind):
    """Twelve RPN steps walking over TWO image shapes (so two captured steps and eager steps interleave), every other one deferred:
    losses and trained weights equal the all-eager run's bit for bit; a second phase (compile: new slots, new learning rate) keeps
    replaying the same graphs, since the optimiser is not part of them."""
    from faster_rcnn_amd import resnet, train
    from faster_rcnn_amd.weights import synthetic_resnet
    A = 9
    w0 = synthetic_resnet(50, anchors_per_loc=A, num_classes=21, seed=71)
    sizes = [(96, 128), (128, 160)]
    feed = []
    for i in range(12):
        h, w = sizes[(i // 2) % 2] if i < 8 else sizes[i % 2]
        rows, cols = resnet.get_conv_rows_cols(h, w)
        feed.append((image(h, w, seed=100 + i), rpn_targets(rows, cols, A, seed=200 + i)))

    def run(on):
        with graphs(on):
            rpn = resnet.resnet50_rpn(resnet.resnet50_base(weights=_copy(w0), weight_regularizer=resnet.WEIGHT_REGULARIZER,
                                                           bias_regularizer=resnet.BIAS_REGULARIZER, dtype=dtype), anchors_per_loc=A)
            opt = train.SGD(1e-2, 0.9) if opt_kind == "sgd" else train.Adam(1e-3)
            rpn.compile(opt)
            got = []
            for i, (x, y) in enumerate(feed):
                if i == 8:
                    opt.lr = opt.lr * 0.1
                    rpn.compile(opt)
                got.append(rpn.train_on_batch(x, list(y), defer=bool(i & 1)))
            got = [g.result() if hasattr(g, "result") else g for g in got]
            n = _captured(rpn)
            return got, [rpn.get_layer(k).get_weights() for k in ("rpn_conv1", "rpn_out_bbreg", "res4a_branch2a", "res4f_branch2c")], n

    l_e, w_e, n_e = run(False)
    l_g, w_g, n_g = run(True)
    assert n_e == 0 and n_g == 2
    assert l_e == l_g
    for a, b in zip(w_e, w_g):
        for u, v in zip(a, b):
            assert np.array_equal(u, v)


def _det_batches(rs, xs, rows, cols, n, C, count):
    out = []
    for i in range(count):
        rois = np.stack([rs.randint(0, 3, n), rs.randint(0, 2, n), rs.randint(4, cols - 1, n), rs.randint(3, rows - 1, n)], axis=1).astype(np.float32)[None]
        yc = np.zeros((1, n, C), np.int32)
        yc[0, np.arange(n), rs.randint(0, C, n)] = 1
        yb = (rs.randn(1, n, 8 * (C - 1)) * (rs.rand(1, n, 8 * (C - 1)) < 0.1)).astype(np.float32)
        out.append(([xs[i], rois], [yc, yb]))
    return out


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_detector_step2_replayed_equals_eager(dtype):
    """Seven detector steps fed with images (step 2: stages 1-3 frozen on the prefix stream, RoI crop forward and backward, head,
    stage 4): replayed == eager, bit for bit."""
    from faster_rcnn_amd import resnet, train
    from faster_rcnn_amd.weights import synthetic_resnet
    C, n, A = 21, 8, 9
    w0 = synthetic_resnet(50, anchors_per_loc=A, num_classes=C, seed=72)
    rows, cols = resnet.get_conv_rows_cols(96, 128)
    xs = [image(96, 128, seed=300 + i) for i in range(7)]
    batches = _det_batches(np.random.RandomState(6), xs, rows, cols, n, C, 7)

    def run(on):
        with graphs(on):
            det = resnet.resnet50_classifier(n, C, base_model=resnet.resnet50_base(weights=_copy(w0), weight_regularizer=resnet.WEIGHT_REGULARIZER,
                                                                                    bias_regularizer=resnet.BIAS_REGULARIZER, dtype=dtype))
            det.compile(train.SGD(1e-3, 0.9))
            got = [det.train_on_batch(x, y, defer=bool(i & 1)) for i, (x, y) in enumerate(batches)]
            got = [g.result() if hasattr(g, "result") else g for g in got]
            return got, [det.get_layer(k).get_weights() for k in ("res4a_branch2a", "res5c_branch2c", "dense_class_%d" % C, "dense_reg_%d" % C)], _captured(det)

    l_e, w_e, n_e = run(False)
    l_g, w_g, n_g = run(True)
    assert (n_e, n_g) == (0, 1)
    assert np.isfinite(np.asarray(l_e)).all()
    assert l_e == l_g
    for a, b in zip(w_e, w_g):
        for u, v in zip(a, b):
            assert np.array_equal(u, v)


def test_detector_step4_on_conv_features_replayed_equals_eager():
    """Step 4 (train_util.py:133-193): a base-less detector on cached conv features -- no prefix graph, the feature map is the
    main part's own input."""
    from faster_rcnn_amd import resnet, train
    from faster_rcnn_amd.weights import synthetic_resnet
    C, n, A = 21, 8, 9
    w0 = synthetic_resnet(50, anchors_per_loc=A, num_classes=C, seed=73)
    rows, cols = 6, 8
    rs = np.random.RandomState(7)
    feats = [np.maximum(rs.randn(1, rows, cols, 1024), 0).astype(np.float32) for _ in range(6)]
    batches = _det_batches(rs, feats, rows, cols, n, C, 6)

    def run(on):
        with graphs(on):
            det = resnet.resnet50_classifier(n, C, base_model=None, weights=_copy(w0))
            det.compile(train.SGD(1e-3, 0.9))
            got = [det.train_on_batch(x, y) for x, y in batches]
            return got, [det.get_layer(k).get_weights() for k in ("res5a_branch2a", "res5c_branch2c", "dense_reg_%d" % C)], _captured(det)

    l_e, w_e, n_e = run(False)
    l_g, w_g, n_g = run(True)
    assert (n_e, n_g) == (0, 1)
    assert np.isfinite(np.asarray(l_e)).all()
    assert l_e == l_g
    for a, b in zip(w_e, w_g):
        for u, v in zip(a, b):
            assert np.array_equal(u, v)


def test_step_graph_cache_is_bounded_and_dropped_on_load_weights(tmp_path):
    """At most STEP_GRAPH_SHAPES captured shapes (least recently used first out); load_weights after compile rebuilds the trainer and
    destroys the old trainer's graphs; training goes on from the loaded values."""
    from faster_rcnn_amd import resnet, train
    from faster_rcnn_amd.weights import synthetic_resnet
    A = 9
    w0 = synthetic_resnet(50, anchors_per_loc=A, num_classes=21, seed=74)
    with graphs(True, after=0, shapes=2):
        rpn = resnet.resnet50_rpn(resnet.resnet50_base(weights=_copy(w0)), anchors_per_loc=A)
        rpn.compile(train.SGD(1e-2, 0.9))
        for k, (h, w) in enumerate([(96, 128), (128, 160), (96, 160), (96, 128)]):
            rows, cols = resnet.get_conv_rows_cols(h, w)
            l = rpn.train_on_batch(image(h, w, seed=k), list(rpn_targets(rows, cols, A, seed=k)))
            assert all(np.isfinite(l))
            assert _captured(rpn) == min(k + 1, 2)
        old = rpn._trainer
        path = str(tmp_path / "w.npz")
        rpn.save_weights(path)
        rpn.load_weights(path)
        assert rpn._trainer is not old and _captured(rpn) == 0 and len(old._graphs["graphs"]) == 0
        rows, cols = resnet.get_conv_rows_cols(96, 128)
        assert all(np.isfinite(rpn.train_on_batch(image(96, 128, seed=9), list(rpn_targets(rows, cols, A, seed=9)))))


def test_evicted_step_graphs_give_their_memory_back():
    """Five image shapes against a cache of two captured steps: every few steps a captured step is evicted and another captured.  The
    device reservation must not grow with the number of evictions (an evicted step's private pools go back to the device) and the
    losses stay finite (scripts/dev/r6_soak_graphs.py is the long form: 12 shapes, two trainers, 720 steps each)."""
    from faster_rcnn_amd import resnet, train
    from faster_rcnn_amd.weights import synthetic_resnet
    A = 9
    w0 = synthetic_resnet(50, anchors_per_loc=A, num_classes=21, seed=75)
    shapes = [(96, 128), (112, 128), (96, 160), (128, 160), (112, 176)]
    with graphs(True, after=0, shapes=2):
        rpn = resnet.resnet50_rpn(resnet.resnet50_base(weights=_copy(w0), dtype="bf16"), anchors_per_loc=A)
        rpn.compile(train.SGD(1e-4, 0.9))
        reserved = []
        for it in range(60):
            h, w = shapes[it % len(shapes)]
            rows, cols = resnet.get_conv_rows_cols(h, w)
            l = rpn.train_on_batch(image(h, w, seed=it), list(rpn_targets(rows, cols, A, seed=it)))
            assert all(np.isfinite(l)), (it, l)
            if it in (19, 59):
                torch.cuda.synchronize()
                reserved.append(torch.cuda.memory_reserved())
        assert _captured(rpn) == 2
        assert reserved[1] <= reserved[0] + (64 << 20), reserved
