"""One process per GPU over torch.distributed (backend "nccl" = RCCL over xGMI on ROCm; "gloo" on
CPU for tests).  Takes the role mpi4py's COMM_WORLD plays in the reference
(gaussianprocess.py:15-20, calculator.py:6-8, RBF_mb.py:257-301): rank / size, broadcast of small
host objects, barriers -- plus the tensor collectives the sharded covariance assembly needs."""
import os

import torch
import torch.distributed as dist


class Comm:
    """World communicator; degrades to a serial no-op when torch.distributed is not initialised."""

    def __init__(self, group=None):
        self.group = group

    @property
    def active(self):
        return dist.is_available() and dist.is_initialized()

    @property
    def rank(self):
        return dist.get_rank(self.group) if self.active else 0

    @property
    def size(self):
        return dist.get_world_size(self.group) if self.active else 1

    def Get_rank(self):
        return self.rank

    def Get_size(self):
        return self.size

    def barrier(self):
        if self.active:
            dist.barrier(self.group)

    Barrier = barrier

    def bcast(self, obj, root=0):
        if not self.active or self.size == 1:
            return obj
        box = [obj]
        dist.broadcast_object_list(box, src=root, group=self.group)
        return box[0]

    def allreduce(self, value, op=None):
        """sum of a python float / numpy array over ranks (gaussianprocess.py:246-247)"""
        if not self.active or self.size == 1:
            return value
        box = [None] * self.size
        dist.all_gather_object(box, value, group=self.group)
        out = box[0]
        for v in box[1:]:
            out = out + v
        return out

    # ---- tensor collectives (device tensors on nccl, cpu tensors on gloo)
    def all_gather_rows(self, local, counts):
        """concatenate row slabs [counts[r], ncols] of every rank into [sum(counts), ncols]"""
        if not self.active or self.size == 1:
            return local
        # all_gather wants equal shapes: pad every slab to the largest one, then cut the padding off
        ncols, cmax = local.shape[1], int(max(counts))
        padded = torch.zeros((cmax, ncols), dtype=local.dtype, device=local.device)
        padded[:local.shape[0]] = local
        parts = [torch.empty_like(padded) for _ in range(self.size)]
        dist.all_gather(parts, padded, group=self.group)
        return torch.cat([p[:int(c)] for p, c in zip(parts, counts)], dim=0)

    def all_reduce_sum_(self, t):
        if self.active and self.size > 1:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t

    def broadcast_(self, t, root=0):
        if self.active and self.size > 1:
            dist.broadcast(t, src=root, group=self.group)
        return t


def init_from_env(backend=None):
    """torchrun-style initialisation (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*); returns Comm"""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            # GPR_DIST_BACKEND=gloo lets several ranks share one GPU (rehearsal of the N > 1 path on a
            # one-GPU box: RCCL refuses two ranks on the same device)
            backend = os.environ.get("GPR_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if torch.cuda.is_available():
            ndev = max(torch.cuda.device_count(), 1)
            torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) % ndev)
        dist.init_process_group(backend=backend)
    return Comm()


def shard_range(n, rank, size):
    """contiguous chunk [start, end) of n items for `rank`, the reference's ceil split
    (RBF_mb.py:471-475: chunk_size = (n + size - 1) // size)"""
    chunk = (n + size - 1) // size
    start = min(rank * chunk, n)
    return start, min(start + chunk, n)


def balanced_ranges(weights, size):
    """contiguous ranges of items with about equal total weight (rows) per rank"""
    import numpy as np
    w = np.asarray(weights, dtype=np.float64)
    total = w.sum()
    edges = [0]
    if total <= 0:
        return [(0, len(w)) if r == 0 else (len(w), len(w)) for r in range(size)]
    cum = np.cumsum(w)
    for r in range(1, size):
        edges.append(int(np.searchsorted(cum, total * r / size, side="left")))
    edges.append(len(w))
    for k in range(1, len(edges)):
        edges[k] = max(edges[k], edges[k - 1])
    return [(edges[r], edges[r + 1]) for r in range(size)]
