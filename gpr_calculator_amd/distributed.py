"""Multi-GPU fit: one process per GPU (parallel.Comm over torch.distributed: RCCL on xGMI, gloo in the CPU tests).

What is sharded, and what moves:

  * covariance assembly (`assemble_sharded`): the training points -- energy structures and force environments --
    are cut into contiguous shards, balanced by the work their rows cost.  Rank g computes the row slab of its points,
    lower trapezoid only (K[shard_g, columns of the points up to the end of shard_g]); the slabs are ALL-GATHERED into
    the full matrix on every rank and the upper triangle is filled by mirroring.  The reference splits the same way
    (contiguous chunks of the first set's points, gather to rank 0, broadcast: RBF_mb.py:257-301, 351-431, 471-521).
    Each matrix element crosses the links once (an all-reduce of zero-padded partial matrices moves it twice and adds it).

  * Cholesky (`cholesky_block_cyclic`), for matrices large enough that the N^3/3 matters (cfg4: N = 24 250): 1-D
    block-cyclic rows (256-row blocks, block b on rank b % G), right-looking.  Per block step: the owner factors the
    diagonal block and BROADCASTS its operand tiles (311 KB); every rank solves its own rows of the panel; the panel
    rows are ALL-GATHERED (N x 256 doubles at most); every rank updates its own rows of the trailing matrix on the matrix
    cores -- with one step of look-ahead: the broadcast and the all-gather of step k + 1 run on a second stream beside
    the bulk of step k's update.  The factor is gathered at the end (predictions keep using a replicated L).

Both routines take the local arithmetic as an `ops` object: `DeviceOps` (libgprhip.so) in production, a numpy / scipy
implementation in tests/test_distributed_gloo.py, which runs the very same schedule over gloo with world size 2.
"""
import numpy as np
import torch

from .parallel import balanced_ranges

NB = 256        # row-block size of the distributed factorisation (= the diagonal block of csrc/potrf.hip)


# ------------------------------------------------------------------------------------------------ point shards
def slice_energy(e, p0, p1):
    X, ELE, counts = e
    r0, r1 = int(np.sum(counts[:p0])), int(np.sum(counts[:p1]))
    return (X[r0:r1], ELE[r0:r1], list(counts[p0:p1]))


def slice_force(f, p0, p1):
    X, dXdR, ELE, counts = f
    r0, r1 = int(np.sum(counts[:p0])), int(np.sum(counts[:p1]))
    return (X[r0:r1], dXdR[r0:r1], ELE[r0:r1], list(counts[p0:p1]))


def shard_points(counts_e, counts_f, size):
    """[(e0, e1, f0, f1)] per rank.  Force points are weighted by rows x (rows of all points up to and including
    themselves) -- the cost of their lower-trapezoid slab; energy points (cheap) by rows."""
    ce, cf = np.asarray(counts_e, dtype=np.float64), np.asarray(counts_f, dtype=np.float64)
    er = balanced_ranges(ce, size) if len(ce) else [(0, 0)] * size
    fr = balanced_ranges(cf * (ce.sum() + np.cumsum(cf)), size) if len(cf) else [(0, 0)] * size
    return [(er[r][0], er[r][1], fr[r][0], fr[r][1]) for r in range(size)]


def assemble_sharded(train_x, comm, block_fn, ops, nmat=1):
    """Full symmetric K(train, train) on every rank from per-rank row slabs.

    block_fn(data1, data2) -> tuple of `nmat` matrices [rows(data1), rows(data2)] (K, or K and dK/dl) on this rank's
    device; ops: zeros((r, c)), mirror_lower_(M).  Returns the tuple of gathered matrices."""
    E, F = train_x.get("energy") or (None, None, []), train_x.get("force") or (None, None, None, [])
    nE, nF = len(E[-1]), len(F[-1])
    N = nE + 3 * nF
    shards = shard_points(E[-1], F[-1], comm.size)
    e0, e1, f0, f1 = shards[comm.rank]
    # every rank holds the whole matrix and fills its own rows of it: the slabs are produced where they stay, the
    # others' rows arrive in place (Comm.gather_rows_inplace), nothing is padded, stacked or concatenated
    full = [ops.zeros((N, N)) for _ in range(nmat)]
    if e1 > e0:         # energy rows: columns of the energy points up to the shard's end
        out = block_fn({"energy": slice_energy(E, e0, e1)}, {"energy": slice_energy(E, 0, e1)})
        for K, o in zip(full, out):
            K[e0:e1, :e1] = o
    if f1 > f0:         # force rows: all energy columns, force columns up to the shard's end
        d2 = {"force": slice_force(F, 0, f1)}
        if nE:
            d2 = {"energy": E, "force": d2["force"]}
        out = block_fn({"force": slice_force(F, f0, f1)}, d2)
        for K, o in zip(full, out):
            K[nE + 3 * f0:nE + 3 * f1, :nE + 3 * f1] = o
    row0 = [s[0] for s in shards] + [nE + 3 * s[2] for s in shards]
    cnt = [s[1] - s[0] for s in shards] + [3 * (s[3] - s[2]) for s in shards]
    owners = list(range(comm.size)) * 2
    for K in full:
        comm.gather_slabs_inplace(K, row0, cnt, owners)
        ops.mirror_lower_(K)
    return tuple(full)


# ------------------------------------------------------------------------------------------------ row distribution
class RowBlockCyclic:
    """rows of an n x n matrix in blocks of nb, block b on rank b % size; a rank stores its blocks in increasing order"""

    def __init__(self, n, size, nb=NB):
        self.n, self.size, self.nb = int(n), int(size), int(nb)
        self.nblk = (self.n + self.nb - 1) // self.nb

    def block_rows(self, b):
        return min(self.nb, self.n - b * self.nb)

    def blocks_upto(self, rank, k):
        """number of `rank`'s blocks with id <= k"""
        return 0 if k < rank else (k - rank) // self.size + 1

    def local_row0(self, rank, b):
        """local row at which block b (owned by rank) starts; only the last block can be short, and it is the last local one"""
        return ((b - rank) // self.size) * self.nb

    def local_nrows(self, rank):
        return sum(self.block_rows(b) for b in range(rank, self.nblk, self.size))

    def rows_after(self, rank, k):
        """(first local row, number of local rows) of `rank` in blocks with id > k"""
        r0 = self.blocks_upto(rank, k) * self.nb
        return min(r0, self.local_nrows(rank)), max(self.local_nrows(rank) - r0, 0)

    def global_rows(self, rank):
        return np.concatenate([np.arange(b * self.nb, b * self.nb + self.block_rows(b)) for b in
                               range(rank, self.nblk, self.size)] or [np.zeros(0, dtype=np.int64)]).astype(np.int64)

    def perm_after(self, k):
        """index into the rank-major concatenation of every rank's rows-after-k that lists them in global row order
        (global rows (k + 1) * nb .. n - 1); k = -1: all rows"""
        g = np.arange((k + 1) * self.nb, self.n, dtype=np.int64)
        b = g // self.nb
        r = b % self.size
        cnt = np.array([self.rows_after(q, k)[1] for q in range(self.size)], dtype=np.int64)
        off = np.concatenate(([0], np.cumsum(cnt)[:-1]))
        done = np.array([self.blocks_upto(q, k) for q in range(self.size)], dtype=np.int64)
        local = ((b - r) // self.size - done[r]) * self.nb + g % self.nb
        return off[r] + local


def cholesky_block_cyclic(Aloc, dist, comm, ops):
    """In-place lower Cholesky of a symmetric matrix held as block-cyclic row blocks: Aloc [local rows, n] holds, on
    return, this rank's rows of L (entries above the diagonal are unspecified).  Only the lower triangle is read.
    A diagonal block that is not positive definite sets the owner's info word and the loop carries on (on NaNs) through
    the same collectives on every rank; potrf_distributed makes the ranks agree on info afterwards.

    Schedule (one step of look-ahead).  The work of step k falls into a *chain* -- factor the diagonal block (owner),
    broadcast its tiles, solve the own panel rows, all-gather the panel -- and the trailing update A22 -= P P^T of the
    rank's own rows.  The chain of step k + 1 needs only column block k + 1 of that update, so the update is cut in two:

        side lane:  U_k[column block k + 1]  ->  chain(k + 1): diag, broadcast, panel rows, all-gather
        main lane:  U_k[column blocks k + 2 ...]                      (the bulk of the flops)

    Both lanes run at the same time: the collectives of step k + 1 (311 KB of tiles, then up to n x 256 doubles of panel)
    cross the links while the matrix cores work on the rest of step k.  `ops.side()` is the lane switch (a second stream
    on the device, on which torch.distributed also orders its collectives; nothing at all in the numpy test double, where
    the same calls simply run in this order), `ops.fork_side()` makes the side lane wait for what the main lane has queued,
    `ops.join_side()` the reverse.  Panels land in two buffers used in turn: the all-gather of step k + 1 must not write
    what the update of step k still reads."""
    rank, size, nb, n = comm.rank, comm.size, dist.nb, dist.n
    # row orders of every step, on the device, built once: the gathered panel is rank-major, the update wants global order
    perms = ops.index_tables([dist.perm_after(k) for k in range(dist.nblk - 1)])
    buf = ops.new_tiles()
    lands = [ops.zeros((max(n - nb, 0), nb)) for _ in range(2)] if dist.nblk > 1 else None      # rank-major landing zones

    def chain(k):
        """diag(k) .. all-gather(k) on the current lane; (panel in global row order, r0, m, k0, kb, c1) or None (last block)"""
        k0, kb, owner = k * nb, dist.block_rows(k), k % size
        if rank == owner:
            ops.diag_(Aloc, dist.local_row0(rank, k), k0, kb, buf)
        comm.broadcast_(buf, root=owner)
        c1 = k0 + kb
        if c1 >= n:
            return None
        r0, m = dist.rows_after(rank, k)
        if m > 0:
            ops.panel_(buf, Aloc, r0, k0, kb, m)
        counts = [dist.rows_after(q, k)[1] for q in range(size)]
        mine = Aloc[r0:r0 + m, k0:c1].contiguous()
        land = lands[k % 2][:n - c1] if kb == nb else ops.zeros((n - c1, kb))
        panel = comm.all_gather_rows(mine, counts, out=land)         # rank-major
        panel = ops.take_rows(panel, perms[k])                      # global order: rows c1 .. n - 1
        return panel, r0, m, k0, kb, c1

    with ops.side():
        ops.fork_side()
        state = chain(0)
    for k in range(dist.nblk):
        if state is None:
            break
        panel, r0, m, k0, kb, c1 = state
        ops.join_side(panel)                                         # main lane: panel k and the own panel rows are there
        nxt = dist.block_rows(k + 1)                                 # width of column block k + 1
        with ops.side():
            ops.fork_side()                                          # column block k + 1 was last written by U_{k-1} on the main lane
            if m > 0:
                ops.update_(Aloc, r0, m, k0, kb, panel[:nxt], c1, nxt, dist, rank)
            state = chain(k + 1)
        rest = n - c1 - nxt
        if m > 0 and rest > 0:
            ops.update_(Aloc, r0, m, k0, kb, panel[nxt:], c1 + nxt, rest, dist, rank)
    ops.join_side()
    return dist.nblk


def gather_factor(Aloc, dist, comm, ops):
    """the full n x n factor on every rank, rows in global order"""
    counts = [dist.local_nrows(q) for q in range(comm.size)]
    full = comm.all_gather_rows(Aloc, counts)
    return ops.take_rows(full, ops.index_tables([dist.perm_after(-1)])[0])


def potrf_distributed(K, comm, ops, nb=NB):
    """Cholesky of the replicated symmetric matrix K with the work spread over the ranks; returns the replicated L
    (strict upper triangle zero), the diagonal-block inverses (see device.potrf_lower_) and info -- the SAME info on
    every rank (0, or the smallest failing pivot index any rank met), so that a matrix that is not positive definite
    makes every rank raise, as the reference does, instead of leaving some of them in the next collective."""
    n = K.shape[0]
    dist = RowBlockCyclic(n, comm.size, nb)
    if hasattr(ops, "lookahead") and "GPR_DIST_LOOKAHEAD" not in __import__("os").environ:
        # two lanes where the collectives move device memory themselves (RCCL); a backend that stages through the host
        # (gloo: rehearsals with several ranks on one device) serialises the lanes anyway and only pays for the second stream
        ops.lookahead = bool(getattr(comm, "device_collectives", True))
    Aloc = ops.take_rows(K, ops.index_tables([dist.global_rows(comm.rank)])[0])
    cholesky_block_cyclic(Aloc, dist, comm, ops)
    L = gather_factor(Aloc, dist, comm, ops)
    L, dinv, info = ops.finish_(L)
    return L, dinv, ops.agree_info_(info, comm)


# ------------------------------------------------------------------------------------------------ device arithmetic
class OpsBase:
    """what the schedule needs besides arithmetic, the same for every implementation of the local arithmetic"""

    # lanes of cholesky_block_cyclic: one lane here (calls run in program order); DeviceOps overrides with two streams
    def side(self):
        import contextlib
        return contextlib.nullcontext()

    def fork_side(self):
        pass

    def join_side(self, *tensors):
        pass

    def index_tables(self, tables):
        """several host index arrays -> index tensors where the matrices live, one upload for all of them"""
        if not tables:
            return []
        lens = [len(t) for t in tables]
        flat = np.concatenate([np.asarray(t, dtype=np.int64) for t in tables]) if sum(lens) else np.zeros(0, dtype=np.int64)
        devt = torch.from_numpy(np.ascontiguousarray(flat)).to(self.zeros((0, 0)).device)
        return list(torch.split(devt, lens))

    def take_rows(self, M, idx):
        if not torch.is_tensor(idx):
            idx = torch.from_numpy(np.ascontiguousarray(idx)).to(M.device)
        return M.index_select(0, idx)

    def agree_info_(self, info, comm):
        """smallest non-zero info over the ranks (0 if none), identical everywhere"""
        if comm.size == 1:
            return info
        big = 2 ** 30
        t = torch.where(info != 0, big - info.to(torch.int64), torch.zeros_like(info, dtype=torch.int64)).to(torch.float64)
        comm.all_reduce_max_(t)
        v = t.to(torch.int64)
        return torch.where(v != 0, big - v, v).to(torch.int32)


class DeviceOps(OpsBase):
    """local arithmetic of the routines above on the GPU (libgprhip.so)"""

    def __init__(self):
        from . import _lib, device as dev
        self.lib, self.dev = _lib.lib(), dev
        self.info = torch.zeros(1, dtype=torch.int32, device=dev.device())
        self.ntile = int(self.lib.gpr_potrf_ops_doubles())

    def zeros(self, shape):
        return torch.zeros(shape, dtype=self.dev.F64, device=self.dev.device())

    # ---- the two lanes of cholesky_block_cyclic.  The main lane is the stream that is current when the factorisation
    # starts; the side lane is a stream of this object.  Kernels go where torch's current stream points
    # (device.stream_ptr), and torch.distributed orders a collective behind the current stream and lets the current
    # stream wait for it -- so `with ops.side():` moves the chain of a step, collectives included, off the main lane.
    _side = _main = None
    # GPR_DIST_LOOKAHEAD=0: one lane -- the chain and the update of a step run one after the other in the order they are
    # written (what HostOps does; for A/B runs and for rehearsals in which several ranks share ONE device, where a second
    # stream per process only adds context switches between the processes)
    lookahead = __import__("os").environ.get("GPR_DIST_LOOKAHEAD", "1") != "0"

    def side(self):
        if not self.lookahead:
            return OpsBase.side(self)
        if self._side is None:
            self._side = torch.cuda.Stream(device=self.dev.device())
        self._main = torch.cuda.current_stream(self.dev.device())
        return torch.cuda.stream(self._side)

    def fork_side(self):
        """(called on the side lane) wait for everything the main lane has queued so far"""
        if self.lookahead:
            self._side.wait_stream(self._main)

    def join_side(self, *tensors):
        """(called on the main lane) wait for everything the side lane has queued so far; tensors made there are about
        to be read here"""
        if self._side is None or not self.lookahead:
            return
        torch.cuda.current_stream(self.dev.device()).wait_stream(self._side)
        for t in tensors:
            t.record_stream(torch.cuda.current_stream(self.dev.device()))

    def mirror_lower_(self, M):
        self.dev.check(self.lib.gpr_mirror_lower(self.dev.stream_ptr(), M.shape[0], self.dev.ptr(M), int(M.stride(0))), "gpr_mirror_lower")
        return M

    def new_tiles(self):
        return torch.empty(self.ntile, dtype=self.dev.F64, device=self.dev.device())

    def _at(self, M, r, c):
        return M.data_ptr() + 8 * (r * int(M.stride(0)) + c)

    def diag_(self, Aloc, r, c, kb, buf):
        self.dev.check(self.lib.gpr_potrf_diag_block(self.dev.stream_ptr(), kb, self._at(Aloc, r, c), int(Aloc.stride(0)), c,
                                                     self.dev.ptr(self.info), self.dev.ptr(buf)), "gpr_potrf_diag_block")

    def panel_(self, buf, Aloc, r0, c, kb, m):
        self.dev.check(self.lib.gpr_potrf_panel_rows(self.dev.stream_ptr(), kb, self.dev.ptr(buf), self._at(Aloc, r0, c),
                                                     int(Aloc.stride(0)), m), "gpr_potrf_panel_rows")

    def update_(self, Aloc, r0, m, c, kb, panel, c1, ncols, dist, rank):
        ld = int(Aloc.stride(0))
        self.dev.check(self.lib.gpr_gemm_nt_cyclic_lower(
            self.dev.stream_ptr(), m, ncols, kb, -1.0, self._at(Aloc, r0, c), ld, self.dev.ptr(panel), int(panel.stride(0)), 1.0,
            self._at(Aloc, r0, c1), ld, dist.nb, dist.size, rank, r0 // dist.nb, c1), "gpr_gemm_nt_cyclic_lower")

    def finish_(self, L):
        n = L.shape[0]
        dinv = torch.empty(((n + 63) // 64) * 4096, dtype=self.dev.F64, device=L.device)
        self.dev.check(self.lib.gpr_potrf_finish(self.dev.stream_ptr(), n, self.dev.ptr(L), int(L.stride(0)), self.dev.ptr(dinv)),
                       "gpr_potrf_finish")
        return L, dinv, self.info
