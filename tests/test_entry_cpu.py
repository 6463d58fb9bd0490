"""Host logic of the captured inference entry (faster_rcnn_amd/entry.py) that needs no GPU: the per-size graph cache's
recency order, byte budget and busy-slot rule, and which (manager, detector) pairs take the captured path."""
import types

from faster_rcnn_amd import entry


def fake(nbytes):
    s = entry._Slot()
    s.nbytes, s.busy = nbytes, False
    return s


def test_graph_cache_lru_budget_and_busy_slots():
    made = []

    def make(n):
        def f():
            made.append(n)
            return fake(n)
        return f
    c = entry.GraphCache(byte_budget=250)
    a = c.acquire((600, 1000), make(100))
    b = c.acquire((600, 800), make(100))
    assert c.acquire((600, 1000), make(100)) is a and c.hits == 1            # idle slot of that size: reused, now most recent
    assert c.keys() == [(600, 800), (600, 1000)]
    a.busy = True
    a2 = c.acquire((600, 1000), make(100))                                   # same size while the first is in flight: a second instance
    assert a2 is not a and len(c) == 2 and c.evictions == 1                  # 300 > 250: the idle least-recent size (600x800) went
    assert c.keys() == [(600, 1000)] and c.nbytes == 200
    a2.busy = True
    d = c.acquire((375, 500), make(100))                                     # both 600x1000 slots busy: nothing evictable but over budget
    assert len(c) == 3 and c.nbytes == 300 and c.evictions == 1
    a.busy = a2.busy = False
    e = c.acquire((500, 375), make(100))                                     # now the oldest idle ones go until the budget holds
    assert c.nbytes <= 250 and e in [s for v in c._slots.values() for s in v] and d in [s for v in c._slots.values() for s in v]
    assert made == [100] * 5 and c.captures == 5
    c.clear()
    assert len(c) == 0 and c.nbytes == 0
    assert b is not None


def test_only_this_packages_models_take_the_captured_path():
    mgr = types.SimpleNamespace(rpn_model=object(), conv_only=True)
    assert not entry.DetectionEntry.usable(mgr, object(), 64)                # foreign Keras-style models: eager path
    assert entry.for_models(mgr, object()) is None
    assert entry.default_in_flight("bf16") == 4


def test_cubic_tap_tables_from_the_c_abi_equal_the_numpy_restatement():
    """frcnn_resize_cubic_taps is a HOST function of the library (no GPU): OpenCV's f32 tap arithmetic written in C must give
    the integers shapes._cubic_taps derives in numpy, for enlarging, shrinking, odd and degenerate sizes."""
    import numpy as np
    from faster_rcnn_amd import ops, shapes
    rs = np.random.RandomState(0)
    pairs = [(800, 500), (600, 375), (375, 600), (1000, 353), (901, 500), (7, 3), (3, 7), (1, 5), (5, 1), (640, 640)]
    pairs += [(int(rs.randint(1, 1600)), int(rs.randint(1, 1600))) for _ in range(60)]
    for dst, src in pairs:
        tab = ops.resize_cubic_taps(dst, src)
        idx, coef = shapes._cubic_taps(dst, src)
        assert tab.shape == (dst, 8) and np.array_equal(tab[:, :4], idx) and np.array_equal(tab[:, 4:], coef), (dst, src)
