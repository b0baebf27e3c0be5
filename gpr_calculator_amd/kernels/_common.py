"""Machinery shared by RBF_mb and Dot_mb: packing of the reference's data dictionaries onto the
device (with a cache keyed on the identity of the descriptor arrays) and assembly of the block
matrix [[K_ee, K_ef], [K_fe, K_ff]] in place on the GPU (gpr_calc/kernels/RBF_mb.py:135-171,
Dot_mb.py:87-119, base.py:3-30)."""
import weakref

import numpy as np
import torch

from .. import _lib
from .. import device as dev
from ..parallel import multi
from ..utilities import list_to_tuple


def as_energy_tuple(e):
    """accept the (X, ELE, counts) tuple or a list of (x, ele) (kee_C: rbf_kernel.py:26-27), or an
    already packed device SetPair (its raw device tensors are returned)"""
    if isinstance(e, dev.SetPair):
        x, _, ele = e.raw_rows()
        return (x, ele, e.counts)
    if isinstance(e, tuple) and len(e) == 3 and not isinstance(e[0], tuple):
        return e
    return list_to_tuple(list(e), mode="energy")


def as_force_tuple(f, stress=False):
    """accept (X, dXdR, ELE, counts) or a list / object array of (x, dxdr, ele) (kef_C: rbf_kernel.py:112-113)"""
    if isinstance(f, dev.SetPair):
        x, dxdr, ele = f.raw_rows()
        return (x, dxdr, ele, f.counts)
    if isinstance(f, tuple) and len(f) == 4 and not isinstance(f[0], tuple):
        return f
    return list_to_tuple(list(f), stress=stress)


def n_points(entry):
    """number of points of a data-dict entry in either representation"""
    if isinstance(entry, dev.SetPair):
        return entry.npts
    if isinstance(entry, tuple) and len(entry) in (3, 4) and not isinstance(entry[0], tuple):
        return len(entry[-1])
    return len(entry)


class PackCache:
    """id(X)-keyed cache of packed sets; entries die with their numpy array (the reference replaces
    the training arrays on every append -- gaussianprocess.py:590-595 -- so identity is a safe key)."""

    def __init__(self, capacity=8):
        self.capacity = capacity
        self._items = {}

    def get(self, key_array, extra, build):
        key = (id(key_array), key_array.shape, extra)
        hit = self._items.get(key)
        if hit is not None and hit[0]() is key_array:
            return hit[1]
        packed = build()
        try:
            ref = weakref.ref(key_array)
        except TypeError:
            return packed
        if len(self._items) >= self.capacity:
            dead = [k for k, v in self._items.items() if v[0]() is None]
            for k in dead or list(self._items)[:1]:
                self._items.pop(k, None)
        self._items[key] = (ref, packed)
        return packed

    def clear(self):
        self._items.clear()


class DeviceKernel:
    """Base of the kernel objects.  Subclasses provide name, hyper-parameters and `_kp(block)`."""

    def __init__(self):
        self._cache = PackCache()

    # ---- packing
    def pack_energy(self, e, flags=0):
        if isinstance(e, dev.SetPair):
            return e.variant(flags)
        X, ELE, counts = as_energy_tuple(e)
        if isinstance(X, torch.Tensor):
            return dev.SetPair(X, None, ELE, counts, flags=flags)
        return self._cache.get(X, ("E", flags), lambda: dev.SetPair(X, None, ELE, counts, flags=flags))

    def pack_force(self, f, flags=0):
        if isinstance(f, dev.SetPair):
            return f.variant(flags)
        X, dXdR, ELE, counts = as_force_tuple(f)
        if isinstance(X, torch.Tensor):
            return dev.SetPair(X, dXdR[:, :, :3].contiguous(), ELE, counts, flags=flags)
        if dXdR.shape[-1] != 3:
            dXdR = np.ascontiguousarray(dXdR[:, :, :3])
        return self._cache.get(X, ("F", flags), lambda: dev.SetPair(X, dXdR, ELE, counts, flags=flags))

    # ---- block assembly
    def _assemble(self, data1, data2, kp_ee, kp_ef, kp_ff, grad=False, transpose=False, comm=None):
        """K (and dK/dl when grad) of data1 x data2 on the device.

        transpose=True returns K(data1, data2)^T, computed directly in that layout, with data1
        still on the streamed side (predict: data1 = structures, data2 = training set).
        comm (parallel.Comm of size > 1), covariance of one set with itself (fit, log-marginal likelihood): the points
        are cut into contiguous shards, every rank computes the lower-trapezoid row slab of its shard and the slabs are
        all-gathered (distributed.assemble_sharded).  Two different sets: the point-pair tasks of every block are dealt
        round-robin to the ranks and the partial matrices summed with one all-reduce (each element has one owner)."""
        if multi(comm) and data2 is None and not transpose:
            from ..distributed import DeviceOps, assemble_sharded

            def block_fn(d1, d2):
                out = self._assemble(d1, d2, kp_ee, kp_ef, kp_ff, grad=grad)
                return out[:2] if grad else (out,)
            mats = assemble_sharded(data1, comm, block_fn, DeviceOps(), nmat=2 if grad else 1)
            nE = n_points(data1["energy"]) if "energy" in data1 and len(data1["energy"]) > 0 else 0
            return (mats[0], mats[1], nE) if grad else mats[0]
        if multi(comm):
            dev.set_shard(comm.size, comm.rank)
            try:
                out = self._assemble(data1, data2, kp_ee, kp_ef, kp_ff, grad=grad, transpose=transpose)
            finally:
                dev.set_shard(1, 0)
            for t in (out[:2] if grad else (out,)):
                comm.all_reduce_sum_(t)
            return out
        same = data2 is None
        if same:
            data2 = data1
        E1 = self.pack_energy(data1["energy"]) if "energy" in data1 and n_points(data1["energy"]) > 0 else None
        F1 = self.pack_force(data1["force"]) if "force" in data1 and n_points(data1["force"]) > 0 else None
        if same:
            E2, F2 = E1, F1
        else:
            E2 = self.pack_energy(data2["energy"]) if "energy" in data2 and n_points(data2["energy"]) > 0 else None
            F2 = self.pack_force(data2["force"]) if "force" in data2 and n_points(data2["force"]) > 0 else None
        nE1, nF1 = (E1.npts if E1 else 0), (F1.npts if F1 else 0)
        nE2, nF2 = (E2.npts if E2 else 0), (F2.npts if F2 else 0)
        rows, cols = nE1 + 3 * nF1, nE2 + 3 * nF2
        shape = (cols, rows) if transpose else (rows, cols)
        K = torch.zeros(shape, dtype=dev.F64, device=dev.device())
        Kl = torch.zeros(shape, dtype=dev.F64, device=dev.device()) if grad else None

        def sub(M, r0, r1, c0, c1):
            return M[c0:c1, r0:r1] if transpose else M[r0:r1, c0:c1]

        if E1 and E2:
            dev.block_ee(kp_ee, E1.stream, E2.resident, symmetric=same and not transpose, transpose=transpose,
                         out=sub(K, 0, nE1, 0, nE2), grad=grad, dl_out=sub(Kl, 0, nE1, 0, nE2) if grad else None)
        if E1 and F2:
            dev.block_ef(kp_ef, E1.stream, F2.resident, transpose=transpose,
                         out=sub(K, 0, nE1, nE2, cols), grad=grad, dl_out=sub(Kl, 0, nE1, nE2, cols) if grad else None)
        if F1 and E2:
            if same:       # C_fe = C_ef.T (RBF_mb.py:165, Dot_mb.py:115)
                sub(K, nE1, rows, 0, nE2).copy_(sub(K, 0, nE1, nE2, cols).T)
                if grad:
                    sub(Kl, nE1, rows, 0, nE2).copy_(sub(Kl, 0, nE1, nE2, cols).T)
            else:          # kef_C(E2, F1, transpose=True)
                dev.block_ef(kp_ef, E2.stream, F1.resident, transpose=not transpose,
                             out=sub(K, nE1, rows, 0, nE2), grad=grad,
                             dl_out=sub(Kl, nE1, rows, 0, nE2) if grad else None)
        if F1 and F2:
            dev.block_ff(kp_ff, F1.stream, F2.resident, symmetric=same and not transpose, transpose=transpose,
                         out=sub(K, nE1, rows, nE2, cols), grad=grad,
                         dl_out=sub(Kl, nE1, rows, nE2, cols) if grad else None)
        return (K, Kl, nE1) if grad else K
