"""Data-parallel host logic: one process per GPU, images sharded by rank, ONE all-reduce of the
flat gradient buffer per step (RCCL over xGMI through torch.distributed, backend "nccl"; the same
code runs on the "gloo" backend for the CPU tests).  The reference is single-process
(train_util.py:38-54); with world_size == 1 every function here is the identity."""
import os

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


def broadcast_(flat, src=0):
    """Identical initial weights on every rank."""
    if world() > 1:
        dist.broadcast(flat, src=src)
    return flat
