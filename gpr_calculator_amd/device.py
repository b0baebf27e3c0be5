"""Thin Python layer over the L1 C-ABI (include/gpr_hip.h): torch owns device memory and streams,
libgprhip.so does the arithmetic.  Nothing here computes on the CPU."""
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import KernelParams, check

F64 = torch.float64


_DEVICES = {}


def device():
    """current HIP device (raises when there is none: the package computes on the GPU only)"""
    if not _DEVICES:
        _lib.require_gpu()
    idx = torch.cuda.current_device()
    d = _DEVICES.get(idx)
    if d is None:
        d = _DEVICES[idx] = torch.device("cuda", idx)
        # first use of this device: the prediction session's slot streams take their place in the process's stream order now,
        # before the factorisation's look-ahead context, RCCL or anything else creates streams (csrc/predict.hip: the pool)
        _lib.lib().gpr_dot2_session_reserve_streams()
    return d


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_cur_device = getattr(torch._C, "_cuda_getDevice", None)


def stream_ptr():
    """torch's current stream of the current device as a raw handle.  Asked some twenty times per prediction of the
    pair-kernel route: the two C entry points cost 0.3 us where building a torch.cuda.Stream object cost 8."""
    if _raw_stream is not None and _cur_device is not None:
        return C.c_void_p(_raw_stream(_cur_device()))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class DevArray:
    """A device array owned by a native handle (raw pointer + length + dtype); `owner` keeps the handle alive.
    Accepted wherever this module takes device tensors that are only passed on to the library."""

    def __init__(self, pointer, n, dtype, owner):
        self.pointer, self.n, self.dtype, self.owner = int(pointer), int(n), dtype, owner
        self.shape = (self.n,)

    def reshape(self, *_):
        return self

    def data_ptr(self):
        return self.pointer

    def tensor(self):
        """a torch copy (debug / fallback paths)"""
        out = torch.empty(self.n, dtype=self.dtype, device=device())
        check(_lib.lib().gpr_memcpy_dtod(stream_ptr(), C.c_void_p(out.data_ptr()), C.c_void_p(self.pointer),
                                         self.n * out.element_size()), "gpr_memcpy_dtod")
        return out


def to_dev(a, dtype=F64):
    """numpy / tensor -> contiguous device tensor of dtype"""
    if isinstance(a, DevArray):
        if a.dtype != dtype:
            raise TypeError(f"device array of {a.dtype} where {dtype} is needed")
        return a
    if isinstance(a, torch.Tensor):
        if a.is_cuda and a.dtype == dtype and a.is_contiguous():
            return a
        return a.to(device=device(), dtype=dtype).contiguous()
    return torch.as_tensor(np.ascontiguousarray(a), dtype=dtype).to(device())


def ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(None)


def set_shard(nshards, shard):
    """restrict the following block calls of this thread to one shard of the point-pair tasks"""
    check(_lib.lib().gpr_set_shard(int(nshards), int(shard)), "gpr_set_shard")


def kernel_params(kind, zeta, sigma, l=1.0, sigma0=0.0, tol=0.0):
    return KernelParams(int(kind), float(zeta), float(sigma), float(l), float(sigma0), float(tol))


class PackedSet:
    """A list of energy / force points packed on the device (gpr_set).

    x[nrows, d], dxdr[nrows, d, width] or None, ele[nrows], counts[npts] follow the reference's
    `list_to_tuple` tuple (gpr_calc/utilities.py:340-390)."""

    def __init__(self, role, x, dxdr, ele, counts, col0=0, flags=0, x_index=None, dxdr_index=None):
        counts = np.ascontiguousarray(np.asarray(counts, dtype=np.int32))
        self.role, self.flags, self.col0 = role, flags, col0
        self.npts = int(len(counts))
        self.counts = counts
        self.x = to_dev(x)
        if self.x.dim() != 2:
            raise ValueError("x must be [nrows, d]")
        self.d = int(self.x.shape[1])
        self.ele = to_dev(ele, torch.int32).reshape(-1)
        if dxdr is not None:
            self.dxdr = to_dev(dxdr)
            self.width = int(self.dxdr.shape[-1])
        else:
            self.dxdr, self.width = None, 0
        # optional gather (gpr_set_create_gather): packed row r reads x[x_index[r]], dxdr[dxdr_index[r]]
        self.x_index = None if x_index is None else to_dev(x_index, torch.int32).reshape(-1)
        self.dxdr_index = None if dxdr_index is None else to_dev(dxdr_index, torch.int32).reshape(-1)
        idx = self.x_index if self.x_index is not None else self.dxdr_index
        self.nrows = int(self.x.shape[0]) if idx is None else int(idx.shape[0])
        nsum = int(counts.sum())
        if nsum != self.nrows:
            raise ValueError(f"counts sum to {nsum} but there are {self.nrows} rows")
        self._h = C.c_void_p()
        check(_lib.lib().gpr_set_create_gather(
            C.byref(self._h), stream_ptr(), role, self.d, self.width, col0, self.nrows, self.npts,
            counts.ctypes.data_as(_lib.c_int_p), ptr(self.x), ptr(self.dxdr), ptr(self.ele),
            ptr(self.x_index), ptr(self.dxdr_index), flags), "gpr_set_create_gather")

    @property
    def handle(self):
        return self._h

    def info(self):
        a, b, c, d = C.c_int(), C.c_int(), C.c_int(), C.c_longlong()
        check(_lib.lib().gpr_set_info(self._h, C.byref(a), C.byref(b), C.byref(c), C.byref(d)), "gpr_set_info")
        return dict(npts=a.value, nrows=b.value, rows_padded=c.value, bytes=d.value)

    def close(self):
        if self._h:
            _lib.lib().gpr_set_destroy(self._h, stream_ptr())
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class SetPair:
    """The same points packed in both roles (a set that appears on both sides of a block needs both); each
    role is packed on first use.  With x_index / dxdr_index the rows are gathered inside the pack kernel
    (force points regrouped from the descriptor kernel's output, gpr_set_create_gather)."""

    def __init__(self, x, dxdr, ele, counts, flags=0, x_index=None, dxdr_index=None):
        self.flags = flags
        self.counts = np.ascontiguousarray(np.asarray(counts, dtype=np.int32))
        self.x = to_dev(x)
        self.dxdr = None if dxdr is None else to_dev(dxdr)
        self.ele = to_dev(ele, torch.int32).reshape(-1)
        self.x_index = None if x_index is None else to_dev(x_index, torch.int32).reshape(-1)
        self.dxdr_index = None if dxdr_index is None else to_dev(dxdr_index, torch.int32).reshape(-1)
        idx = self.x_index if self.x_index is not None else self.dxdr_index
        self.npts = int(len(self.counts))
        self.nrows = int(self.x.shape[0]) if idx is None else int(idx.shape[0])
        self._sets = {}
        self._variants = {}

    def _role(self, role):
        s = self._sets.get(role)
        if s is None:
            s = self._sets[role] = PackedSet(role, self.x, self.dxdr, self.ele, self.counts, flags=self.flags,
                                             x_index=self.x_index, dxdr_index=self.dxdr_index)
        return s

    @property
    def stream(self):
        return self._role(_lib.ROLE_STREAM)

    @property
    def resident(self):
        return self._role(_lib.ROLE_RESIDENT)

    def row0_dev(self):
        """exclusive prefix of the per-point row counts, int32 [npts + 1] on the device"""
        if getattr(self, "_row0_dev", None) is None:
            r0 = np.zeros(self.npts + 1, dtype=np.int32)
            np.cumsum(self.counts, out=r0[1:])
            self._row0_dev = to_dev(r0, torch.int32)
        return self._row0_dev

    def variant(self, flags):
        """the same points under other packing flags (sharing the device tensors)"""
        if flags == self.flags:
            return self
        v = self._variants.get(flags)
        if v is None:
            v = self._variants[flags] = SetPair(self.x, self.dxdr, self.ele, self.counts, flags=flags,
                                                x_index=self.x_index, dxdr_index=self.dxdr_index)
        return v

    def raw_rows(self):
        """(x, dxdr, ele) row tensors with the gather applied (materialised; fallback paths only)"""
        x, dxdr = self.x, self.dxdr
        ele = self.ele.tensor() if isinstance(self.ele, DevArray) else self.ele
        if self.x_index is not None:
            x, ele = x.index_select(0, self.x_index.long()), ele.index_select(0, self.x_index.long())
        if dxdr is not None and self.dxdr_index is not None:
            dxdr = dxdr.index_select(0, self.dxdr_index.long())
        return x, dxdr, ele

    def close(self):
        for s in self._sets.values():
            s.close()
        self._sets = {}
        for v in self._variants.values():
            v.close()
        self._variants = {}


def _view(out, nrows, ncols):
    if out is None:
        out = torch.empty((nrows, ncols), dtype=F64, device=device())
    assert out.dtype == F64 and out.is_cuda and out.stride(-1) == 1 and tuple(out.shape) == (nrows, ncols), \
        (out.shape, (nrows, ncols), out.stride())
    return out


BLOCK_SYMMETRIC, BLOCK_TRANSPOSE = 1, 2


def block_ee(kp, s1, s2, symmetric=False, out=None, grad=False, dl_out=None, transpose=False):
    shape = (s2.npts, s1.npts) if transpose else (s1.npts, s2.npts)
    o = _view(out, *shape)
    dl = _view(dl_out, *shape) if grad else None
    flags = (BLOCK_SYMMETRIC if symmetric else 0) | (BLOCK_TRANSPOSE if transpose else 0)
    check(_lib.lib().gpr_block_ee(stream_ptr(), C.byref(kp), s1.handle, s2.handle, flags, ptr(o),
                                  max(int(o.stride(0)), shape[1]), ptr(dl), max(int(dl.stride(0)), shape[1]) if grad else 0),
          "gpr_block_ee")
    return (o, dl) if grad else o


def block_ef(kp, es1, fs2, transpose=False, out=None, grad=False, dl_out=None):
    shape = (3 * fs2.npts, es1.npts) if transpose else (es1.npts, 3 * fs2.npts)
    o = _view(out, *shape)
    dl = _view(dl_out, *shape) if grad else None
    check(_lib.lib().gpr_block_ef(stream_ptr(), C.byref(kp), es1.handle, fs2.handle, int(transpose), ptr(o),
                                  max(int(o.stride(0)), shape[1]), ptr(dl), max(int(dl.stride(0)), shape[1]) if grad else 0),
          "gpr_block_ef")
    return (o, dl) if grad else o


def block_ff(kp, s1, s2, symmetric=False, out=None, grad=False, dl_out=None, transpose=False):
    shape = (3 * s2.npts, 3 * s1.npts) if transpose else (3 * s1.npts, 3 * s2.npts)
    o = _view(out, *shape)
    dl = _view(dl_out, *shape) if grad else None
    flags = (BLOCK_SYMMETRIC if symmetric else 0) | (BLOCK_TRANSPOSE if transpose else 0)
    check(_lib.lib().gpr_block_ff(stream_ptr(), C.byref(kp), s1.handle, s2.handle, flags, ptr(o),
                                  max(int(o.stride(0)), shape[1]), ptr(dl), max(int(dl.stride(0)), shape[1]) if grad else 0),
          "gpr_block_ff")
    return (o, dl) if grad else o


def diag_ff(kp, pair, out=None):
    o = out if out is not None else torch.empty(3 * pair.npts, dtype=F64, device=device())
    check(_lib.lib().gpr_diag_ff(stream_ptr(), C.byref(kp), pair.stream.handle, pair.resident.handle, ptr(o)),
          "gpr_diag_ff")
    return o


def diag_ee(kp, pair, out=None):
    o = out if out is not None else torch.empty(pair.npts, dtype=F64, device=device())
    check(_lib.lib().gpr_diag_ee(stream_ptr(), C.byref(kp), pair.stream.handle, pair.resident.handle, ptr(o)),
          "gpr_diag_ee")
    return o


# ------------------------------------------------------------------ dense linear algebra
def potrf_lower_(a, want_dinv=False):
    """in-place lower Cholesky of a square row-major device matrix; returns info (0 = ok), and with
    want_dinv also the inverses of the 64 x 64 diagonal blocks of L (for trsm_lower_)"""
    n = a.shape[0]
    info = torch.zeros(1, dtype=torch.int32, device=a.device)
    dinv = torch.empty(((n + 63) // 64) * 4096, dtype=F64, device=a.device)
    check(_lib.lib().gpr_potrf_lower_dinv(stream_ptr(), n, ptr(a), int(a.stride(0)) if n > 1 else n, ptr(info),
                                          ptr(dinv)), "gpr_potrf_lower_dinv")
    return (info, dinv) if want_dinv else info


class PipelineTimeout(RuntimeError):
    """the look-ahead schedule of the factorisation gave up waiting for a hand-over (info = -1): a statement about the
    schedule, never about the matrix"""


_timeout_logged = False


def potrf_factor(make, log=None, comm=None):
    """(L, info, dinv) of the matrix `make()` builds.  info = -1 means that a polled hand-over between two running kernels
    of the factorisation timed out (gpr_hip.h, gpr_potrf_lower_dinv) and says nothing about the matrix, which is partly
    overwritten by then: the hand-overs are switched to stream events for the rest of the process (said once), the
    matrix is built again and factored on that schedule.  A second time-out raises PipelineTimeout -- it is never
    reported as 'not positive definite'.  With a communicator of several ranks `make` may contain collectives, so the
    ranks agree (one small all-reduce) on whether anyone timed out and repeat together."""
    global _timeout_logged
    a = make()
    info, dinv = potrf_lower_(a, want_dinv=True)
    timed_out = int(info.item()) == -1
    if comm is not None and getattr(comm, "size", 1) > 1:
        flag = torch.tensor([1.0 if timed_out else 0.0], dtype=F64, device=a.device)
        timed_out = float(comm.all_reduce_sum_(flag).item()) > 0.0
    if timed_out:
        check(_lib.lib().gpr_potrf_set_polling(0), "gpr_potrf_set_polling")
        if not _timeout_logged:
            _timeout_logged = True
            msg = ("gpr_potrf_lower: a polled hand-over of the look-ahead timed out; the factorisation now orders its "
                   "kernels by stream events only and is repeated")
            (log.warning if log is not None else print)(msg)
        del a
        a = make()
        info, dinv = potrf_lower_(a, want_dinv=True)
        if int(info.item()) == -1:
            raise PipelineTimeout("gpr_potrf_lower: hand-over time-out on the event-ordered schedule")
    return a, info, dinv


def tri_inv64(l):
    """inverses of the 64 x 64 diagonal blocks of a lower-triangular factor (the `dinv` of potrf_lower_), for a factor
    that was put together block-wise; also zeroes the strict upper triangle of l"""
    n = l.shape[0]
    dinv = torch.empty(((n + 63) // 64) * 4096, dtype=F64, device=l.device)
    check(_lib.lib().gpr_potrf_finish(stream_ptr(), n, ptr(l), int(l.stride(0)) if n > 1 else n, ptr(dinv)), "gpr_potrf_finish")
    return dinv


def trtri_lower(l):
    """L^-1 of a lower-triangular device matrix (zeros above the diagonal), 256-row blocks on the matrix cores"""
    n = l.shape[0]
    x = torch.zeros((n, n), dtype=F64, device=l.device)
    check(_lib.lib().gpr_trtri_lower(stream_ptr(), n, ptr(l), int(l.stride(0)) if n > 1 else n, ptr(x), n), "gpr_trtri_lower")
    return x


def trsm_lower_(l, b, trans=False, dinv=None):
    """b <- L^-1 b (trans False) or L^-T b (trans True); b is [n, nrhs] row-major, in place"""
    n, nrhs = b.shape
    check(_lib.lib().gpr_trsm_lower_dinv(stream_ptr(), int(trans), n, nrhs, ptr(l), int(l.stride(0)) if n > 1 else n,
                                         ptr(dinv), ptr(b), int(b.stride(0)) if n > 1 else nrhs), "gpr_trsm_lower_dinv")
    return b


def gemm(a, b, transa=False, transb=False, alpha=1.0, beta=0.0, out=None):
    m = a.shape[1] if transa else a.shape[0]
    k = a.shape[0] if transa else a.shape[1]
    n = b.shape[0] if transb else b.shape[1]
    kb = b.shape[1] if transb else b.shape[0]
    assert k == kb, (a.shape, b.shape, transa, transb)
    if out is None:
        out = torch.empty((m, n), dtype=F64, device=a.device)
        beta = 0.0
    check(_lib.lib().gpr_gemm(stream_ptr(), int(transa), int(transb), m, n, k, float(alpha), ptr(a),
                              max(int(a.stride(0)), a.shape[1]), ptr(b), max(int(b.stride(0)), b.shape[1]),
                              float(beta), ptr(out), max(int(out.stride(0)), n)), "gpr_gemm")
    return out


def xtx_lower(x):
    """x^T x of a lower-triangular device matrix (zeros above the diagonal), the full symmetric result: K^-1 from L^-1"""
    n = x.shape[0]
    out = torch.empty((n, n), dtype=F64, device=x.device)
    check(_lib.lib().gpr_xtx_lower(stream_ptr(), n, ptr(x), max(int(x.stride(0)), n), ptr(out), n), "gpr_xtx_lower")
    return out


def trmm_lower(t, b, out=None):
    """t @ b with t lower triangular [n, n], b [n, nrhs] (row-major device matrices)"""
    n, nrhs = b.shape
    if out is None:
        out = torch.empty((n, nrhs), dtype=F64, device=b.device)
    check(_lib.lib().gpr_trmm_lower(stream_ptr(), n, nrhs, ptr(t), max(int(t.stride(0)), n), ptr(b),
                                    max(int(b.stride(0)), nrhs), ptr(out), max(int(out.stride(0)), nrhs)),
          "gpr_trmm_lower")
    return out


def trmm_lower_sumsq(t, b):
    """column sums of squares of t @ b, t lower triangular [n, n], b [n, nrhs]: |L^-1 k*|^2 per predicted quantity"""
    n, nrhs = b.shape
    out = torch.empty(nrhs, dtype=F64, device=b.device)
    check(_lib.lib().gpr_trmm_lower_sumsq(stream_ptr(), n, nrhs, ptr(t), max(int(t.stride(0)), n), ptr(b),
                                          max(int(b.stride(0)), nrhs), ptr(out)), "gpr_trmm_lower_sumsq")
    return out


def add_diag_(a, n_first, add_first, add_rest):
    n = a.shape[0]
    check(_lib.lib().gpr_add_diag(stream_ptr(), n, ptr(a), max(int(a.stride(0)), n), int(n_first), float(add_first),
                                  float(add_rest)), "gpr_add_diag")
    return a


def colsumsq(v):
    n, ncols = v.shape
    out = torch.empty(ncols, dtype=F64, device=v.device)
    check(_lib.lib().gpr_colsumsq(stream_ptr(), n, ncols, ptr(v), max(int(v.stride(0)), ncols), ptr(out)),
          "gpr_colsumsq")
    return out


def sum_log_diag(a):
    n = a.shape[0]
    out = torch.empty(1, dtype=F64, device=a.device)
    check(_lib.lib().gpr_sum_log_diag(stream_ptr(), n, ptr(a), max(int(a.stride(0)), n), ptr(out)),
          "gpr_sum_log_diag")
    return out


def frob_dot(a, b):
    """sum(a * b) of two equally shaped row-major device matrices -> 1-element device tensor"""
    assert a.shape == b.shape
    a2 = a if a.dim() == 2 else a.reshape(-1, 1)
    b2 = b if b.dim() == 2 else b.reshape(-1, 1)
    m, n = a2.shape
    out = torch.empty(1, dtype=F64, device=a.device)
    check(_lib.lib().gpr_frob_dot(stream_ptr(), m, n, ptr(a2), max(int(a2.stride(0)), n), ptr(b2),
                                  max(int(b2.stride(0)), n), ptr(out)), "gpr_frob_dot")
    return out


def profile_enable(on=True):
    check(_lib.lib().gpr_profile_enable(int(on)), "gpr_profile_enable")


def profile_fetch(scaled=True):
    """{'ff': (ms, launches), 'ef': ..., 'ee': ..., 'diag': ...} kernel-only times since profile_enable().  With a
    sampling period (profile_enable(n), n > 1) only every n-th launch of a kind carries events; scaled=True returns
    the recorded time scaled to all launches seen (mean recorded launch x launches seen) and the launches seen."""
    names = ("ff", "ef", "ee", "diag", "trmm", "gemm", "so3", "features", "pack", "gemv", "so3_radial")
    seen = (C.c_int * len(names))()
    check(_lib.lib().gpr_profile_counts(seen), "gpr_profile_counts")
    ms = (C.c_double * len(names))()
    n = (C.c_int * len(names))()
    check(_lib.lib().gpr_profile_fetch(ms, n), "gpr_profile_fetch")
    if not scaled:
        return {k: (ms[i], n[i]) for i, k in enumerate(names)}
    return {k: ((ms[i] / n[i] * seen[i]) if n[i] else 0.0, seen[i]) for i, k in enumerate(names)}


def dot2_features(pset, species, sigma0, out=None):
    """Psi rows (Dot kernel, zeta = 2) of a STREAM-packed set: [npts, F] for energies, [3*npts, F] for forces"""
    species = np.ascontiguousarray(np.asarray(species, dtype=np.int32))
    ns = len(species)
    F = ns * (pset.d * (pset.d + 1) // 2 + 1)
    rows = pset.npts * (3 if pset.width else 1)
    if out is None:
        out = torch.empty((rows, F), dtype=F64, device=device())
    check(_lib.lib().gpr_dot2_features(stream_ptr(), pset.handle, ns, species.ctypes.data_as(_lib.c_int_p),
                                       float(sigma0), ptr(out), max(int(out.stride(0)), F)), "gpr_dot2_features")
    return out


def ws_finish(P, w, Q=None, diag=None, tail=0):
    """[mean | var] = [P w | max(diag - rowsum(Q o P), 0)] as one [2, m] (or [1, m]) device tensor (gpr_ws_finish);
    with tail > 0 returns (out, flat) where flat is the 1-D buffer [mean | var | tail words]: status words that should
    ride along with the result download go into its last `tail` slots"""
    m, n = P.shape
    k = 2 if Q is not None else 1
    flat = torch.empty(k * m + tail, dtype=F64, device=P.device)
    out = flat[:k * m].view(k, m)
    check(_lib.lib().gpr_ws_finish(stream_ptr(), m, n, ptr(P), int(P.stride(0)), ptr(Q),
                                   int(Q.stride(0)) if Q is not None else 0, ptr(w), ptr(diag), ptr(out), 0.0), "gpr_ws_finish")
    return (out, flat) if tail else out


def dot2_features_direct(pair, species, sigma, sigma0, want_diag=True, psi_out=None, diag_out=None, species_all=None):
    """(Psi rows, prior variances) of the points of an unpacked SetPair straight from its row tensors and
    index arrays (gpr_dot2_features_direct): [npts, F] / [npts] for energies, [3 npts, F] / [3 npts] for forces;
    psi_out / diag_out: preallocated row windows to write into"""
    species = np.ascontiguousarray(np.asarray(species, dtype=np.int32))
    ns = len(species)
    # species of these points that the model does not know (prior variance only)
    if species_all is None:
        ele = pair.ele.tensor() if isinstance(pair.ele, DevArray) else pair.ele
        species_all = torch.unique(ele).cpu().tolist()
    extra = np.ascontiguousarray([z for z in sorted(set(int(v) for v in species_all)) if z not in set(species.tolist())],
                                 dtype=np.int32)
    d = int(pair.x.shape[1])
    width = 0 if pair.dxdr is None else 3
    if width and int(pair.dxdr.shape[-1]) != 3:
        raise ValueError("direct features take dxdr[rows, d, 3]")
    F = ns * (d * (d + 1) // 2 + 1)
    rows = pair.npts * (3 if width else 1)
    psi = psi_out if psi_out is not None else torch.empty((rows, F), dtype=F64, device=device())
    diag = (diag_out if diag_out is not None else torch.empty(rows, dtype=F64, device=device())) if want_diag else None
    assert tuple(psi.shape) == (rows, F) and psi.stride(1) == 1 and (diag is None or diag.shape[0] == rows)
    check(_lib.lib().gpr_dot2_features_direct(
        stream_ptr(), d, width, pair.npts, ptr(pair.row0_dev()), ptr(pair.x_index), ptr(pair.dxdr_index),
        ptr(pair.x), ptr(pair.dxdr), ptr(pair.ele), ns, species.ctypes.data_as(_lib.c_int_p), len(extra),
        extra.ctypes.data_as(_lib.c_int_p), float(sigma),
        float(sigma0), ptr(psi), int(psi.stride(0)), ptr(diag)), "gpr_dot2_features_direct")
    return psi, diag


def rowdot(a, b):
    """out[i] = sum_j a[i, j] * b[i, j] for two equally shaped row-major device matrices"""
    assert a.shape == b.shape and a.dim() == 2
    m, n = a.shape
    out = torch.empty(m, dtype=F64, device=a.device)
    check(_lib.lib().gpr_rowdot(stream_ptr(), m, n, ptr(a), max(int(a.stride(0)), n), ptr(b), max(int(b.stride(0)), n),
                                ptr(out)), "gpr_rowdot")
    return out


def gemv_t(a, x):
    """a^T x for a row-major [n, ncols] device matrix and an [n] / [n, 1] vector -> [ncols]"""
    n, ncols = a.shape
    out = torch.empty(ncols, dtype=F64, device=a.device)
    check(_lib.lib().gpr_gemv_t(stream_ptr(), n, ncols, ptr(a), max(int(a.stride(0)), ncols), ptr(x.reshape(-1)),
                                ptr(out)), "gpr_gemv_t")
    return out
