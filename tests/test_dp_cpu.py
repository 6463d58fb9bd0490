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
    # world 1 reproduces the reference schedule exactly (train_util.py:39)
    assert [dp.image_index(i, 2, 10, 7) for i in range(10)] == [(i + 10 * 2) % 7 for i in range(10)]
