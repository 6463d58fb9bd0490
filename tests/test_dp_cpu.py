"""CPU, world_size 2, gloo: the data-parallel host logic (image sharding schedule, the single
flat-buffer all-reduce and its 1/world scale, identical initial weights via broadcast)."""
import os
import socket
import sys

import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from faster_rcnn_amd import dp
    r, w = dp.init_from_env(backend="gloo")
    assert (r, w) == (rank, world) and dp.rank() == rank and dp.world() == world
    # identical weights after broadcast
    flat_w = torch.full((1000,), float(rank + 1))
    dp.broadcast_(flat_w)
    assert bool((flat_w == 1.0).all())
    # one all-reduce of the flat gradient buffer; a "skipped image" rank contributes zeros
    g = torch.arange(1000, dtype=torch.float32) * (rank + 1) if rank == 0 else torch.zeros(1000)
    scale = dp.allreduce_sum_(g)
    assert scale == 0.5
    assert torch.allclose(g * scale, torch.arange(1000, dtype=torch.float32) * 0.5)
    # the asynchronous form the training step uses (started behind the last gradient kernel, waited for when the optimiser
    # is enqueued): same sum, same scale
    g2 = torch.arange(1000, dtype=torch.float32) * (rank + 1)
    handle, scale2 = dp.allreduce_sum_begin(g2)
    assert handle is not None and scale2 == 0.5
    handle.wait()
    assert torch.equal(g2, torch.arange(1000, dtype=torch.float32) * 3)
    # schedule: the two ranks cover consecutive images of the reference schedule, no overlap
    idx = [dp.image_index(i, 1, 10, 7) for i in range(10)]
    out[rank] = idx
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_dp_gloo_world2():
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    a, b = out[0], out[1]
    ref = [(j + 10 * 1 * 2) % 7 for j in range(20)]          # the reference schedule with 2 images per global step
    assert [v for pair in zip(a, b) for v in pair] == ref


def test_single_process_is_identity():
    sys.path.insert(0, ROOT)
    from faster_rcnn_amd import dp
    assert dp.world() == 1 and dp.rank() == 0
    g = torch.ones(5)
    assert dp.allreduce_sum_(g) == 1.0 and bool((g == 1).all())
    assert dp.allreduce_sum_begin(g) == (None, 1.0)                      # nothing to wait for: the step updates at once
    # world 1 reproduces the reference schedule exactly (train_util.py:39)
    assert [dp.image_index(i, 2, 10, 7) for i in range(10)] == [(i + 10 * 2) % 7 for i in range(10)]


# ---------------------------------------------------------------------------------------------------------------
# dp.ImageSchedule: the reference's image walk + shuffles (train_util.py:39-41) under data parallelism
def _reference_walk(n_images, phases, seed):
    """What the reference's single process visits: random.shuffle(images) whenever img_idx == 0."""
    import random
    rng = random.Random(seed)
    images = list(range(n_images))
    seen = []
    for phase_num, num_iterations in enumerate(phases):
        for i in range(num_iterations):
            idx = (i + num_iterations * phase_num) % n_images
            if idx == 0:
                rng.shuffle(images)
            seen.append(images[idx])
    return seen


def test_image_schedule_world1_is_the_reference_walk():
    import random
    sys.path.insert(0, ROOT)
    from faster_rcnn_amd import dp
    for n_images, phases in ((7, [10, 10]), (5, [12, 3, 9]), (1, [4]), (6, [6, 6])):
        random.seed(1234)
        sched = dp.ImageSchedule(list(range(n_images)), rank_=0, world_=1)
        got = []
        for phase_num, num_iterations in enumerate(phases):
            sched.begin_phase(phase_num, num_iterations)
            got += [sched.image(i) for i in range(num_iterations)]
        assert got == _reference_walk(n_images, phases, 1234)          # global `random` stream, same calls in the same order


def _dp_walk(n_images, phases, world, seed):
    """Every rank's ImageSchedule side by side; returns per-step tuples of image ids and each rank's final order."""
    from faster_rcnn_amd import dp
    scheds = [dp.ImageSchedule(list(range(n_images)), rank_=r, world_=world, seed=seed) for r in range(world)]
    steps = []
    for phase_num, num_iterations in enumerate(phases):
        for s in scheds:
            s.begin_phase(phase_num, num_iterations)
        for i in range(num_iterations):
            steps.append(tuple(s.image(i) for s in scheds))
    return steps, scheds


def test_image_schedule_ranks_share_every_permutation():
    """world 2..8, image counts that do and do not divide: flattened over ranks the walk equals the single-process
    walk of the reference schedule with `world` images per step, drawn from ONE shuffle stream -- so an epoch is
    partitioned (no image duplicated or dropped), also when the wrap falls between two ranks of one step."""
    import random
    sys.path.insert(0, ROOT)
    for world in (2, 3, 8):
        for n_images, phases in ((7, [10, 10]), (16, [5, 9]), (5, [6]), (9, [4, 4, 4])):
            steps, scheds = _dp_walk(n_images, phases, world, seed=99)
            flat = [v for st in steps for v in st]
            # single-process restatement: positions offset + i*world + r, shuffle when position % n == 0
            rng = random.Random(99)
            images = list(range(n_images))
            want = []
            for phase_num, num_iterations in enumerate(phases):
                off = num_iterations * phase_num * world
                for q in range(num_iterations * world):
                    if (off + q) % n_images == 0:
                        rng.shuffle(images)
                    want.append(images[(off + q) % n_images])
            assert flat == want, (world, n_images, phases)
            # inside one pass over the data (between two wraps) no image repeats
            run = []
            off = 0
            for q, v in enumerate(flat[:phases[0] * world]):
                if q % n_images == 0:
                    run = []
                assert v not in run
                run.append(v)
    # ranks that have applied the same number of shuffles hold the same order
    steps, scheds = _dp_walk(10, [10], 2, seed=5)
    assert scheds[0].applied == scheds[1].applied and scheds[0].images == scheds[1].images


def _sched_worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import random
    from faster_rcnn_amd import dp
    dp.init_from_env(backend="gloo")
    random.seed(1000 + rank)                        # ranks deliberately disagree on the GLOBAL stream (sampling uses it)
    sched = dp.ImageSchedule(list(range(7)))
    seen = []
    for phase_num, num_iterations in enumerate([6, 6]):
        sched.begin_phase(phase_num, num_iterations)
        for i in range(num_iterations):
            seen.append(sched.image(i))
            random.random()                         # per-image sampling draws from the global stream in between
    out[rank] = (seen, list(sched.images))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_image_schedule_gloo_world2_same_permutation_after_wrap():
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_sched_worker, args=(2, port, out), nprocs=2, join=True)
    (a, order_a), (b, order_b) = out[0], out[1]
    assert order_a == order_b                       # same permutation on both ranks after the wraps (7 images, 12 per phase)
    for x, y in zip(a, b):
        assert x != y                               # 7 is odd: the two ranks of a step never hold the same image
    steps, _ = _dp_walk(7, [6, 6], 2, seed=__import__("faster_rcnn_amd.dp", fromlist=["x"]).DP_SHUFFLE_SEED)
    assert [s[0] for s in steps] == a and [s[1] for s in steps] == b
