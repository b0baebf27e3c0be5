"""Test-side implementations of the two interfaces gpr_calculator_amd/distributed.py is written against:
  * HostOps     the local arithmetic in numpy / scipy on CPU tensors (the schedule then runs without a GPU);
  * ThreadComm  the collectives of parallel.Comm between Python threads of one process (several 'ranks' sharing one
                GPU, or none) -- the same call sequence as over torch.distributed."""
import threading

import numpy as np
import torch
from scipy.linalg import cholesky, solve_triangular

from gpr_calculator_amd.distributed import OpsBase


class HostOps(OpsBase):
    def __init__(self):
        self.info = torch.zeros(1, dtype=torch.int32)

    def zeros(self, shape):
        return torch.zeros(shape, dtype=torch.float64)

    def mirror_lower_(self, M):
        a = M.numpy()
        iu = np.triu_indices(a.shape[0], 1)
        a[iu] = a.T[iu]
        return M

    def new_tiles(self):
        return torch.zeros(256 * 256 + 1, dtype=torch.float64)        # the factored diagonal block itself (+ its size)

    def diag_(self, Aloc, r, c, kb, buf):
        a = Aloc.numpy()
        blk = a[r:r + kb, c:c + kb]
        try:
            L = cholesky(np.tril(blk) + np.tril(blk, -1).T, lower=True, check_finite=False)     # only the lower triangle is meaningful
        except np.linalg.LinAlgError:
            # like the device kernel: record the failing block (first failure wins), carry on with NaNs
            if int(self.info.item()) == 0:
                self.info[0] = c + 1
            L = np.full((kb, kb), np.nan)
        blk[:] = L
        buf[:kb * kb] = torch.from_numpy(L.reshape(-1))
        buf[-1] = kb

    def panel_(self, buf, Aloc, r0, c, kb, m):
        L = buf[:kb * kb].numpy().reshape(kb, kb)
        a = Aloc.numpy()
        a[r0:r0 + m, c:c + kb] = solve_triangular(L, a[r0:r0 + m, c:c + kb].T, lower=True, check_finite=False).T

    def update_(self, Aloc, r0, m, c, kb, panel, c1, ncols, dist, rank):
        a = Aloc.numpy()
        upd = a[r0:r0 + m, c:c + kb] @ panel.numpy().T                  # [m, ncols]
        grow = dist.global_rows(rank)[r0:r0 + m]
        mask = (c1 + np.arange(ncols))[None, :] <= grow[:, None]        # lower triangle of the global matrix only
        a[r0:r0 + m, c1:c1 + ncols] -= np.where(mask, upd, 0.0)

    def finish_(self, L):
        a = L.numpy()
        a[np.triu_indices(a.shape[0], 1)] = 0.0
        return L, None, self.info


class ThreadWorld:
    def __init__(self, size):
        self.size = size
        self.barrier = threading.Barrier(size)
        self.slots = [None] * size


class ThreadComm:
    """parallel.Comm between the threads of one process"""
    active = True

    def __init__(self, world, rank):
        self.w, self._rank = world, rank

    rank = property(lambda self: self._rank)
    size = property(lambda self: self.w.size)

    def Get_rank(self):
        return self._rank

    def Get_size(self):
        return self.w.size

    def barrier(self):
        self.w.barrier.wait()

    Barrier = barrier

    @staticmethod
    def _sync():
        """Thread ranks share one device but not one stream (the look-ahead schedule moves its chain to a side stream per
        rank): what a rank hands to the others must be complete when they read it, and what it read must have been read
        before the owner reuses the memory.  A real backend orders this with stream events; here the host waits."""
        if torch.cuda.is_available() and torch.cuda.is_initialized():
            torch.cuda.current_stream().synchronize()

    def _exchange(self, value):
        self._sync()
        self.w.slots[self._rank] = value
        self.w.barrier.wait()
        out = list(self.w.slots)
        self.w.barrier.wait()
        return out

    def _done(self):
        """every rank has finished reading what the others handed over"""
        self._sync()
        self.w.barrier.wait()

    def bcast(self, obj, root=0):
        return self._exchange(obj)[root]

    def allreduce(self, value, op=None):
        vals = self._exchange(value)
        out = vals[0]
        for v in vals[1:]:
            out = out + v
        return out

    def all_gather_rows(self, local, counts, out=None):
        parts = self._exchange(local)
        cat = torch.cat([p[:int(c)] for p, c in zip(parts, counts)], dim=0)
        if out is None:
            out = cat.clone()
        else:
            out.copy_(cat)
        self._done()
        return out

    def gather_slabs_inplace(self, full, row0, counts, owners):
        mine = [(int(o), int(c), full[int(o):int(o) + int(c)].clone()) for o, c, r in zip(row0, counts, owners) if int(r) == self._rank]
        for part in self._exchange(mine):
            for o, c, rows in part:
                full[o:o + c].copy_(rows)
        self._done()
        return full

    def all_reduce_max_(self, t):
        parts = self._exchange(t.clone())
        total = parts[0].clone()
        for p in parts[1:]:
            total = torch.maximum(total, p)
        t.copy_(total)
        self._done()
        return t

    def all_reduce_sum_(self, t):
        parts = self._exchange(t.clone())
        total = parts[0].clone()
        for p in parts[1:]:
            total += p
        t.copy_(total)
        self._done()
        return t

    def broadcast_(self, t, root=0):
        parts = self._exchange(t)
        if self._rank != root:
            t.copy_(parts[root])
        self._done()                   # the root must not overwrite its tensor before everybody has copied it
        return t


def run_ranks(size, fn):
    """fn(comm) on `size` threads; returns the list of results; re-raises the first failure"""
    world = ThreadWorld(size)
    out, err = [None] * size, [None] * size

    def body(r):
        try:
            out[r] = fn(ThreadComm(world, r))
        except BaseException as e:          # noqa: BLE001 - surfaced below
            err[r] = e
            world.barrier.abort()

    ts = [threading.Thread(target=body, args=(r,)) for r in range(size)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    for e in err:
        if e is not None and not isinstance(e, threading.BrokenBarrierError):
            raise e
    for e in err:
        if e is not None:
            raise e
    return out
