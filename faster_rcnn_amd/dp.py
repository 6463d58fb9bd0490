"""Data-parallel host logic: one process per GPU, images sharded by rank, ONE all-reduce of the
flat gradient buffer per step (RCCL over xGMI through torch.distributed, backend "nccl"; the same
code runs on the "gloo" backend for the CPU tests).  The reference is single-process
(train_util.py:38-54); with world_size == 1 every function here is the identity."""
import os
import random

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Join the process group described by RANK / WORLD_SIZE / LOCAL_RANK / MASTER_*; returns (rank, world)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            local = int(os.environ.get("LOCAL_RANK", "0"))
            torch.cuda.set_device(local)
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend=backend)
    return rank, world


def world():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def image_index(i, phase_num, num_iterations, num_train, rank_=None, world_=None):
    """The reference's schedule is img_idx = (i + num_iterations*phase_num) % num_train
    (train_util.py:39).  Under DP, global step i consumes ``world`` consecutive images of that
    schedule; this rank takes the one at offset ``rank``."""
    r = rank() if rank_ is None else rank_
    w = world() if world_ is None else world_
    return ((i * w + r) + num_iterations * phase_num * w) % num_train


def allreduce_sum_(flat):
    """In-place sum over ranks of the flat gradient buffer; returns the 1/world scale to apply."""
    w = world()
    if w > 1:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    return 1.0 / w


def allreduce_sum_begin(flat):
    """Start the in-place sum over ranks of the flat gradient buffer without waiting for it: returns
    (handle, 1/world scale).  ``handle.wait()`` orders the CURRENT stream behind the collective (RCCL: stream-wise, the
    host does not block; gloo: the host blocks until the exchange has happened).  world == 1: (None, 1.0)."""
    w = world()
    if w == 1 and not (FORCE_COLLECTIVE and dist.is_available() and dist.is_initialized()):
        return None, 1.0
    return dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True), 1.0 / w


# test hook: run the collective even in a one-rank process group (tests/test_dp_gpu.py drives the step's asynchronous
# all-reduce through REAL RCCL on a one-GPU box that way: the sum over one rank is the identity, so the step must stay
# bit-identical to the plain single-process step)
FORCE_COLLECTIVE = os.environ.get("FRCNN_DP_FORCE_COLLECTIVE", "0") != "0"


def broadcast_(flat, src=0):
    """Identical initial weights on every rank."""
    if world() > 1:
        dist.broadcast(flat, src=src)
    return flat


DP_SHUFFLE_SEED = 20171204      # every rank seeds its shuffle stream with this: identical permutations everywhere


class ImageSchedule:
    """Which image a rank trains on at global step ``i`` of a phase, with the reference's shuffles.

    The reference walks ``img_idx = (i + num_iterations*phase_num) % num_train`` and calls
    ``random.shuffle(images)`` whenever that index is 0, before using image 0 (train_util.py:39-41, 100-103).
    Under data parallelism the ``world`` ranks of step ``i`` take the consecutive positions
    ``i*world + rank`` of that walk.  A wrap can fall BETWEEN two ranks of one step (``num_train % world != 0``):
    the ranks before it still use the old order, the ranks at and after it the reshuffled one.  So every rank
    counts how many positions with index 0 lie at or before its own position -- previous phases included, a
    closed form, no communication -- and applies exactly that many shuffles from a stream all ranks seed
    identically (``random.Random(DP_SHUFFLE_SEED)``): rank-independent permutations, no duplicated or dropped
    image.  The global ``random`` stream is left to the per-image sampling (rpn_util._apply_sampling), and with
    world == 1 the shuffles come from it exactly as in the reference (seeded runs reproduce its order).
    """

    def __init__(self, images, rank_=None, world_=None, seed=DP_SHUFFLE_SEED):
        self.images = images
        self.rank = rank() if rank_ is None else rank_
        self.world = world() if world_ is None else world_
        self.rng = random if self.world == 1 else random.Random(seed)
        self.applied = 0            # shuffles applied to self.images so far
        self.before_phase = 0       # index-0 positions in the phases already finished
        self.offset = self.steps = 0

    @staticmethod
    def _zero_hits(a, b, n):
        """positions x in [a, b] with x % n == 0"""
        return b // n - (a - 1) // n if b >= a else 0

    def begin_phase(self, phase_num, num_iterations):
        self.before_phase += self._zero_hits(self.offset, self.offset + self.steps * self.world - 1, len(self.images))
        self.offset, self.steps = num_iterations * phase_num * self.world, num_iterations

    def image(self, i):
        n = len(self.images)
        pos = self.offset + i * self.world + self.rank
        hits = self.before_phase + self._zero_hits(self.offset, pos, n)
        while self.applied < hits:
            self.rng.shuffle(self.images)
            self.applied += 1
        return self.images[pos % n]
