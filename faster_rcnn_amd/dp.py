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

    def peek(self, i):
        """The image ``image(i)`` will return, when getting it needs no shuffle (None otherwise): what a loop may start
        preparing AHEAD of time without touching the shuffle stream before the reference would."""
        n = len(self.images)
        if i >= self.steps:
            return None
        pos = self.offset + i * self.world + self.rank
        hits = self.before_phase + self._zero_hits(self.offset, pos, n)
        return self.images[pos % n] if self.applied >= hits else None

    def image(self, i):
        n = len(self.images)
        pos = self.offset + i * self.world + self.rank
        hits = self.before_phase + self._zero_hits(self.offset, pos, n)
        while self.applied < hits:
            self.rng.shuffle(self.images)
            self.applied += 1
        return self.images[pos % n]


# ----------------------------------------------------------------------------- launch-side host logic (no HIP call anywhere below)
KFD_NODES = "/sys/class/kfd/kfd/topology/nodes"


def _kfd_gpu_nodes(root=KFD_NODES):
    """The GPU nodes of the KFD topology in node order = the order the HIP runtime enumerates them: a list of property dicts
    (CPU nodes have simd_count 0).  Reads sysfs only: a launcher must count devices WITHOUT initialising the HIP runtime (a
    process that has must not start GPU children by exec/fork, and torch.cuda.device_count() is only runtime-free when
    torch's amdsmi path is available: VERDICT r3)."""
    nodes = []
    try:
        names = sorted((n for n in os.listdir(root) if n.isdigit()), key=int)
    except OSError:
        return None
    for n in names:
        props = {}
        try:
            with open(os.path.join(root, n, "properties")) as f:
                for line in f:
                    parts = line.split()
                    if len(parts) == 2:
                        props[parts[0]] = parts[1]
        except OSError:
            continue                                   # (a node this user may not read: not ours to use either)
        if int(props.get("simd_count", "0")) > 0:
            props["node"] = n
            nodes.append(props)
    return nodes


def _apply_visibility(nodes, env):
    """ROCR_VISIBLE_DEVICES filters the runtime's list, HIP_VISIBLE_DEVICES (CUDA_VISIBLE_DEVICES is its alias) indexes
    what is left.  Integer lists only; anything else (UUIDs) -> None: the caller falls back to asking the runtime."""
    for key in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES" if "HIP_VISIBLE_DEVICES" in env else "CUDA_VISIBLE_DEVICES"):
        val = env.get(key)
        if val is None:
            continue
        picked = []
        for tok in (t.strip() for t in val.split(",")):
            if tok == "":
                continue
            if not tok.lstrip("-").isdigit():
                return None
            i = int(tok)
            if i < 0 or i >= len(nodes):
                break                                   # the runtime stops at the first invalid index
            picked.append(nodes[i])
        nodes = picked
    return nodes


def visible_gpus(root=KFD_NODES, env=None):
    """GPUs a child process of this one will see, as KFD property dicts in HIP device order; None when sysfs cannot tell."""
    nodes = _kfd_gpu_nodes(root)
    if nodes is None:
        return None
    return _apply_visibility(nodes, os.environ if env is None else env)


def count_gpus(root=KFD_NODES, env=None):
    """Number of HIP devices a child will see, without touching the runtime; falls back to torch's own count only when the
    KFD topology is unreadable or the visibility variables name devices by UUID."""
    nodes = visible_gpus(root, env)
    return torch.cuda.device_count() if nodes is None else len(nodes)


def _parse_cpulist(text):
    cpus = []
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus += list(range(int(lo), int(hi or lo) + 1))
    return cpus


def gpu_local_cpus(props, pci_root="/sys/bus/pci/devices"):
    """(cpus, numa_node) next to a GPU: KFD gives the PCI address (domain, location_id = bus << 8 | devfn), PCI sysfs the
    CPUs local to that slot.  (None, None) when the platform does not say."""
    try:
        loc, dom = int(props["location_id"]), int(props.get("domain", "0"))
        bdf = "%04x:%02x:%02x.%x" % (dom, (loc >> 8) & 0xFF, (loc >> 3) & 0x1F, loc & 7)
        with open(os.path.join(pci_root, bdf, "local_cpulist")) as f:
            cpus = _parse_cpulist(f.read())
        try:
            with open(os.path.join(pci_root, bdf, "numa_node")) as f:
                numa = int(f.read().strip())
        except OSError:
            numa = -1
        return (cpus or None), numa
    except (OSError, KeyError, ValueError):
        return None, None


def rank_cpu_slice(local_rank, local_world, root=KFD_NODES, pci_root="/sys/bus/pci/devices", env=None, min_cpus=4):
    """The cores rank `local_rank` of `local_world` should run on: the CPUs local to ITS GPU's PCIe slot, shared out evenly
    among the ranks whose GPUs hang off the same set of CPUs (8 GPUs on 2 sockets: 4 ranks split each socket's cores).
    Returns (sorted cpu list, numa node) or (None, None): no topology, a single rank, or fewer than `min_cpus` per rank (the
    step's host side runs a handful of threads: staging casts, the RCCL proxy)."""
    nodes = visible_gpus(root, env)
    if not nodes or local_world <= 1 or local_rank >= len(nodes) or local_world > len(nodes):
        return None, None
    local = [gpu_local_cpus(nodes[r], pci_root) for r in range(local_world)]
    mine, numa = local[local_rank]
    if not mine:
        return None, None
    allowed = set(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else set(mine)
    mine = sorted(c for c in mine if c in allowed)
    peers = [r for r in range(local_world) if local[r][0] and sorted(c for c in local[r][0] if c in allowed) == mine]
    per = len(mine) // len(peers)
    if per < min_cpus:
        return None, None
    k = peers.index(local_rank)
    return mine[k * per:(k + 1) * per], numa


def pin_rank(local_rank, local_world, **kw):
    """sched_setaffinity to rank_cpu_slice(); returns a short description for logs (None = left alone).  A plain syscall:
    safe before and after the HIP runtime starts; threads created later inherit the mask."""
    try:
        cpus, numa = rank_cpu_slice(local_rank, local_world, **kw)
        if not cpus:
            return None
        os.sched_setaffinity(0, cpus)
        return "cpus %d-%d (%d cores, numa node %s)" % (cpus[0], cpus[-1], len(cpus), numa)
    except (OSError, AttributeError, ValueError):
        return None
