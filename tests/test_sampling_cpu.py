"""The host half of the training loops' fast feed: frcnn_host_mt_sample_range replays ``random.sample(range(n), k)`` -- the draw
rpn_util._apply_sampling makes twice per image (rpn_util.py:336-348) -- on the interpreter's own Mersenne-Twister state: the same
list, the same state afterwards, both branches of Lib/random.py's sample().  Host code of the C-ABI library: runs without a GPU."""
import random

import numpy as np
import pytest

from faster_rcnn_amd import dp, rpn_util


@pytest.mark.parametrize("seed", [0, 1, 12345])
def test_sample_range_is_random_sample(seed):
    cases = [(5, 5), (10, 3), (21, 21), (22, 6), (22, 5), (100, 7), (400, 272), (21546, 21290), (64296, 64040), (30000, 5), (1, 1), (2, 1), (1000, 0),
             (85, 64), (86, 64), (2 ** 20, 1000), (2 ** 16, 2 ** 16), (33, 32), (129, 1)]
    for n, k in cases:
        random.seed(seed)
        want = random.sample(range(n), k)
        state_want = random.getstate()
        follow_want = (random.random(), random.randint(0, 10 ** 9), random.gauss(0, 1))
        random.seed(seed)
        got = rpn_util.sample_range(n, k)
        assert got.dtype == np.int32 and list(got) == want, (n, k)
        assert random.getstate() == state_want, (n, k)
        assert (random.random(), random.randint(0, 10 ** 9), random.gauss(0, 1)) == follow_want       # the stream goes on where the interpreter's would


def test_sample_range_mid_stream_and_refusals():
    random.seed(7)
    for _ in range(1000):                                    # generator index in the middle of its 624-word block, then across a refill
        random.random()
    for n, k in ((3000, 2800), (50000, 49872), (700, 699)):
        st = random.getstate()
        want = random.sample(range(n), k)
        after = random.getstate()
        random.setstate(st)
        assert list(rpn_util.sample_range(n, k)) == want and random.getstate() == after
    with pytest.raises(ValueError):
        rpn_util.sample_range(5, 6)
    rpn_util.FAST_SAMPLE = False
    try:
        random.seed(3); a = rpn_util.sample_range(100, 30)
    finally:
        rpn_util.FAST_SAMPLE = True
    random.seed(3); b = rpn_util.sample_range(100, 30)
    assert list(a) == list(b)


def test_apply_sampling_draws_what_the_reference_loop_draws():
    """_apply_sampling on masks: the literal reference statements next to it (random.sample on ranges, fancy-index clears)."""
    rs = np.random.RandomState(2)
    n = 21546
    is_pos = (rs.rand(n) < 0.02)
    can_use = (rs.rand(n) < 0.6)

    def literal(is_pos, can_use):                            # rpn_util.py:324-350, statement by statement
        pos_locs = np.where(np.logical_and(is_pos == 1, can_use == 1))[0]
        neg_locs = np.where(np.logical_and(is_pos == 0, can_use == 1))[0]
        num_pos, num_neg = len(pos_locs), len(neg_locs)
        if num_pos > 128:
            locs_off = random.sample(range(num_pos), num_pos - 128)
            can_use[pos_locs[locs_off]] = 0
            num_pos = 128
        if num_neg + num_pos > 256:
            locs_off = random.sample(range(num_neg), num_neg + num_pos - 256)
            can_use[neg_locs[locs_off]] = 0
        return can_use
    random.seed(11); want = literal(is_pos.copy(), can_use.copy()); st = random.getstate()
    random.seed(11); got = rpn_util._apply_sampling(is_pos.copy(), can_use.copy())
    assert np.array_equal(got, want) and random.getstate() == st
    assert int((got & is_pos).sum()) == 128 and int(got.sum()) == 256


def test_image_schedule_peek_never_shuffles():
    imgs = list(range(5))
    random.seed(5)
    sch = dp.ImageSchedule(list(imgs), rank_=0, world_=1)
    sch.begin_phase(0, 12)
    seen = []
    for i in range(12):
        st = random.getstate()
        ahead = sch.peek(i)
        assert random.getstate() == st                       # peeking consumes nothing
        cur = sch.image(i)
        if ahead is not None:
            assert ahead == cur
        else:
            assert (i % 5) == 0                              # None exactly where image(i) has to shuffle first
        seen.append(cur)
    assert sch.peek(12) is None                              # past the phase
    random.seed(5)
    ref = dp.ImageSchedule(list(imgs), rank_=0, world_=1)
    ref.begin_phase(0, 12)
    assert seen == [ref.image(i) for i in range(12)]
