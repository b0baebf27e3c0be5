"""One process per GPU over torch.distributed (backend "nccl" = RCCL over xGMI on ROCm; "gloo" on
CPU for tests).  Takes the role mpi4py's COMM_WORLD plays in the reference
(gaussianprocess.py:15-20, calculator.py:6-8, RBF_mb.py:257-301): rank / size, broadcast of small
host objects, barriers -- plus the tensor collectives the sharded covariance assembly needs."""
import os

import torch
import torch.distributed as dist


# GPR_DIST_FORCE=1: a world of ONE still goes through the backend (process group initialised, no single-rank short cuts in the
# tensor collectives) -- the rehearsal of the RCCL code paths a one-GPU box allows (tools/rccl_one_rank.py, tests/test_gpu_ranks.py)
_FORCE = os.environ.get("GPR_DIST_FORCE") == "1"


def multi(comm):
    """True when the multi-rank code paths are to run on `comm`: several ranks -- or the rehearsal switch with an initialised
    backend and one rank (the sharded assembly, the block-cyclic factorisation, the weight-space all-reduce and the strong-scaled
    prediction then run with a world of one, every exchange a call into the backend)"""
    return comm is not None and (comm.size > 1 or (_FORCE and bool(getattr(comm, "active", False))))


class Comm:
    """World communicator; degrades to a serial no-op when torch.distributed is not initialised."""

    def __init__(self, group=None):
        self.group = group

    @property
    def _solo(self):
        """nothing to exchange: not initialised, or one rank (unless the rehearsal switch sends that through the backend too)"""
        return not self.active or (self.size == 1 and not _FORCE)

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
        if self._solo:
            return obj
        box = [obj]
        dist.broadcast_object_list(box, src=root, group=self.group)
        return box[0]

    def allreduce(self, value, op=None):
        """sum of a python float / numpy array over ranks (gaussianprocess.py:246-247): one tensor all-reduce, nothing
        pickled"""
        if self._solo:
            return value
        import numpy as np
        arr = np.asarray(value, dtype=np.float64)
        t = torch.from_numpy(np.ascontiguousarray(arr).reshape(-1).copy()).to(self.tensor_device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        out = t.cpu().numpy().reshape(arr.shape)
        return float(out) if arr.ndim == 0 else out

    # ---- tensor collectives (device tensors on nccl, cpu tensors on gloo)
    @property
    def tensor_device(self):
        """where a tensor has to live to go through this communicator's backend: RCCL moves device memory only"""
        if self.active and dist.get_backend(self.group) == "nccl":
            return torch.device("cuda", torch.cuda.current_device())
        return torch.device("cpu")

    @property
    def device_collectives(self):
        """True when the backend moves device memory itself (RCCL over xGMI); gloo stages through the host"""
        return not self.active or dist.get_backend(self.group) == "nccl"

    def all_gather_rows(self, local, counts, out=None):
        """row slabs [counts[r], ncols] of every rank -> [sum(counts), ncols], rank-major, written straight into `out`
        (allocated here when None).  Equal slabs: one all_gather_into_tensor.  Ragged slabs: rank r's slab is broadcast
        into its rows of `out` -- every element crosses a link once and lands where it stays; there is no padded
        staging copy and no concatenation (a 24 250^2 matrix is 4.7 GB: each avoided copy is one of those)."""
        if self._solo:
            if out is None:
                return local
            out[:local.shape[0]].copy_(local)
            return out
        counts = [int(c) for c in counts]
        total, ncols = sum(counts), local.shape[1]
        assert local.shape[0] == counts[self.rank], (local.shape, counts, self.rank)
        if out is None:
            out = torch.empty((total, ncols), dtype=local.dtype, device=local.device)
        assert out.is_contiguous() and tuple(out.shape) == (total, ncols), (out.shape, total, ncols)
        if total == 0 or ncols == 0:
            return out
        if min(counts) == max(counts):
            dist.all_gather_into_tensor(out, local.contiguous(), group=self.group)
            return out
        off, works = 0, []
        for r, c in enumerate(counts):
            if c:
                view = out[off:off + c]
                if r == self.rank and view.data_ptr() != local.data_ptr():
                    view.copy_(local)
                works.append(dist.broadcast(view, src=self._global_rank(r), group=self.group, async_op=True))
            off += c
        for w in works:
            w.wait()
        return out

    def gather_slabs_inplace(self, full, row0, counts, owners):
        """`full` [n, ncols] already holds this rank's row slabs (slab i = rows row0[i] : row0[i] + counts[i], computed by
        rank owners[i]); on return it holds everyone's.  Each element crosses a link once and lands where it stays.
        The slabs come in rounds of one slab per rank, in rank order (the energy rows of every rank, then the force
        rows).  A round is ONE collective where the backend has one for it:
          * equal, adjacent slabs: all_gather_into_tensor straight into their rows of `full`;
          * unequal slabs on RCCL: all_gather with a list of views -- torch issues the G broadcasts of such a call as one
            grouped launch (ncclGroupStart / End), every broadcast writing its rows of `full` directly;
          * otherwise (gloo wants equal sizes; empty slabs): one asynchronous broadcast per slab, waited for together."""
        if self._solo:
            return full
        G = self.size
        slabs = [(int(o), int(c), int(r)) for o, c, r in zip(row0, counts, owners)]
        rounds, loose = [], []
        for i in range(0, len(slabs), G):
            rnd = slabs[i:i + G]
            (rounds if len(rnd) == G and [r for _, _, r in rnd] == list(range(G)) else loose).append(rnd)
        nccl = dist.get_backend(self.group) == "nccl"
        works = []
        for rnd in rounds:
            cs = [c for _, c, _ in rnd]
            views = [full[o:o + c] for o, c, _ in rnd]
            adjacent = all(rnd[q][0] + rnd[q][1] == rnd[q + 1][0] for q in range(G - 1))
            if min(cs) > 0 and min(cs) == max(cs) and adjacent and full.is_contiguous():
                mine = views[self.rank].clone()                      # (not the in-place form: one copy of 1 / G of the rows)
                works.append(dist.all_gather_into_tensor(full[rnd[0][0]:rnd[0][0] + G * cs[0]], mine, group=self.group, async_op=True))
            elif nccl and min(cs) > 0:
                works.append(dist.all_gather(views, views[self.rank].clone(), group=self.group, async_op=True))
            else:
                loose.append(rnd)
        for rnd in loose:
            for o, c, r in rnd:
                if c:
                    works.append(dist.broadcast(full[o:o + c], src=self._global_rank(r), group=self.group, async_op=True))
        for w in works:
            w.wait()
        return full

    def _global_rank(self, r):
        return r if self.group is None else dist.get_global_rank(self.group, r)

    def all_reduce_sum_(self, t):
        if not self._solo:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t

    def all_reduce_max_(self, t):
        if not self._solo:
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        return t

    def broadcast_(self, t, root=0):
        if not self._solo:
            dist.broadcast(t, src=self._global_rank(root), group=self.group)
        return t


def init_from_env(backend=None):
    """torchrun-style initialisation (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*); returns Comm"""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if (world > 1 or _FORCE) and not dist.is_initialized():
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
