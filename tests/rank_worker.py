"""One rank of the multi-GPU path, started by `python -m torch.distributed.run --nproc-per-node N tests/rank_worker.py`
(tests/test_gpu_ranks.py).  With N GPUs the backend is "nccl" (RCCL over xGMI), one rank per device; with
GPR_DIST_BACKEND=gloo the ranks share the box's one GPU and the same calls travel through gloo.  Every collective the
fit and the prediction use is exercised against the single-rank result computed in the same process:

  1. equal and ragged row all-gathers (all_gather_into_tensor / one broadcast per slab), in-place slab exchange (one
     all-gather per round where the backend has one), tensor all-reduce, broadcast;
  2. covariance assembly sharded by training point (row slabs in place + mirror) against the single-rank matrix;
  3. the block-cyclic Cholesky with look-ahead (chain of step k + 1 on a second stream beside the update of step k)
     against the single-device factor, at sizes with ragged last blocks, and an indefinite matrix (same info everywhere);
  4. fit (dense RBF, weight-space Dot), log-marginal likelihood and gradient, and one strong-scaled prediction (a batch dealt
     over the ranks by structure, results through one tensor all-reduce) against the single-rank model.
Rank 0 prints one JSON line; any mismatch raises on the rank that sees it (the launcher then fails the job)."""
import json
import math
import os
import sys
import time

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import torch.distributed as dist


def rel(a, b):
    return float((a - b).abs().max() / b.abs().max())


def build(kind, n_train, comm):
    from gpr_calculator_amd.gaussianprocess import GP
    from gpr_calculator_amd.SO3 import SO3
    from gpr_calculator_amd.kernels.Dot_mb import Dot_mb
    from gpr_calculator_amd.kernels.RBF_mb import RBF_mb
    from gpr_calculator_amd.parallel import Comm
    from gpr_calculator_amd.synthetic import fcc_cell, lj_energy_forces
    kern = Dot_mb(para=[2.0, 2.0], zeta=2.0) if kind == "Dot" else RBF_mb(para=[2.5, 1.3], zeta=2.0)
    gp = GP(kernel=kern, descriptor=SO3(nmax=3, lmax=4, rcut=5.0), noise_e=0.002, noise_f=0.1,
            log_file=f"/tmp/gpr_rank{os.environ.get('RANK', '0')}.log")
    if comm is None:
        gp.comm = _Serial()
        gp.rank, gp.size = 0, 1
    rows, left = [], n_train
    for s in range(math.ceil(n_train / 32)):
        a = fcc_cell(seed=s, species=(46, 78) if s % 3 == 0 else (46,))
        E, F = lj_energy_forces(a)
        take = min(32, left)
        rows.append((a, E, F, True, list(range(take))))
        left -= take
    gp.set_train_pts(gp.points_from_structures(rows))
    return gp


class _Serial:
    rank, size, active = 0, 1, False

    def barrier(self):
        pass

    def bcast(self, obj, root=0):
        return obj

    def all_reduce_sum_(self, t):
        return t

    def allreduce(self, v, op=None):
        return v


def main():
    from gpr_calculator_amd import _lib, device as dev
    from gpr_calculator_amd.distributed import DeviceOps, potrf_distributed
    from gpr_calculator_amd.parallel import init_from_env, multi
    from gpr_calculator_amd.synthetic import fcc_cell, neb_images
    comm = init_from_env()
    _lib.require_gpu()
    rank, size = comm.rank, comm.size
    backend = dist.get_backend() if dist.is_initialized() else None
    d = dev.device()
    report = {"ranks": size, "backend": backend, "devices": torch.cuda.device_count()}
    t_all = time.perf_counter()

    # ---- 1. the collectives themselves
    eq = torch.full((5, 7), float(rank), dtype=torch.float64, device=d)
    got = comm.all_gather_rows(eq, [5] * size)
    assert got.shape == (5 * size, 7) and all(bool((got[5 * r:5 * r + 5] == r).all()) for r in range(size))
    counts = [3 + 2 * r for r in range(size)]
    rag = torch.full((counts[rank], 4), float(10 + rank), dtype=torch.float64, device=d)
    got = comm.all_gather_rows(rag, counts)
    off = 0
    for r, c in enumerate(counts):
        assert bool((got[off:off + c] == 10 + r).all())
        off += c
    for equal in (True, False):                         # slab exchange in place, two rounds (energy rows, force rows)
        c1 = [4] * size if equal else [2 + r for r in range(size)]
        c2 = [6] * size if equal else [5 + 2 * ((r + 1) % size) for r in range(size)]
        row0 = list(np.concatenate(([0], np.cumsum(c1 + c2)[:-1])))
        full = torch.zeros((sum(c1) + sum(c2), 9), dtype=torch.float64, device=d)
        for i in (rank, size + rank):
            full[row0[i]:row0[i] + (c1 + c2)[i]] = 100.0 * (i // size) + rank + 1
        comm.gather_slabs_inplace(full, row0, c1 + c2, list(range(size)) * 2)
        for i in range(2 * size):
            assert bool((full[row0[i]:row0[i] + (c1 + c2)[i]] == 100.0 * (i // size) + (i % size) + 1).all()), (equal, i)
    t = torch.tensor([1.0 + rank, 2.0], dtype=torch.float64, device=d)
    comm.all_reduce_sum_(t)
    assert t.tolist() == [size * (size + 1) / 2.0, 2.0 * size]
    b = torch.full((3,), float(rank), dtype=torch.float64, device=d)
    comm.broadcast_(b, root=size - 1)
    assert b.tolist() == [size - 1.0] * 3
    assert comm.bcast({"k": rank}, root=0) == {"k": 0}

    # ---- 2. sharded covariance assembly
    gp1 = build("RBF", 150, None)
    K1 = gp1.kernel.k_total_dev(gp1.train_x)
    Kg1 = gp1.kernel.k_total_with_grad_dev(gp1.train_x)
    K = gp1.kernel.k_total_dev(gp1.train_x, comm=comm)
    Kg = gp1.kernel.k_total_with_grad_dev(gp1.train_x, comm=comm)
    # (exact symmetry is what the mirrored slabs of the sharded assembly give; an unforced world of one takes the single-rank launch)
    assert rel(K, K1) < 1e-13 and (not multi(comm) or bool(torch.equal(K, K.T))) and rel(Kg[0], Kg1[0]) < 1e-13 and rel(Kg[1], Kg1[1]) < 1e-13
    report["assembly_rel"] = rel(K, K1)

    # ---- 3. block-cyclic Cholesky with look-ahead
    torch.manual_seed(1)
    chol = {}
    for n in (300, 769, 1516, 3000, 6063):
        B = torch.randn(n, n // 2 + 5, dtype=torch.float64, device=d)
        comm.broadcast_(B, root=0)                      # the same matrix everywhere
        A = B @ B.T + torch.eye(n, dtype=torch.float64, device=d) * (0.02 * n + 1.0)
        ref = A.clone()
        info, dinv_ref = dev.potrf_lower_(ref, want_dinv=True)
        torch.cuda.synchronize(); comm.barrier(); t0 = time.perf_counter()
        L, dinv, info_d = potrf_distributed(A.clone(), comm, DeviceOps())
        torch.cuda.synchronize(); comm.barrier()
        chol[n] = round((time.perf_counter() - t0) * 1e3, 2)
        assert int(info.item()) == 0 and int(info_d.item()) == 0
        assert rel(L, ref) < 1e-12, (n, rel(L, ref))
        assert float((dinv - dinv_ref).abs().max()) < 1e-10 * float(dinv_ref.abs().max())
        assert float(torch.triu(L, 1).abs().max()) == 0.0
    report["block_cyclic_ms"] = chol
    bad = A.clone()
    bad[700, 700] = -1.0
    _, _, info_d = potrf_distributed(bad, comm, DeviceOps())
    infos = comm.allreduce(np.eye(size)[rank] * float(info_d.item()))
    assert int(info_d.item()) == 701 and all(int(v) == 701 for v in np.atleast_1d(infos)), infos

    # ---- 4. fit, log-marginal likelihood, strong-scaled prediction
    test = neb_images(6, seed=500)
    theta = np.array([1.7, 1.1])
    for kind in ("RBF", "Dot"):
        single = build(kind, 120, None)
        single.fit(show=False, opt=False)
        lml1, g1 = single.log_marginal_likelihood(theta.copy(), eval_gradient=True)
        single.kernel.update([2.0, 2.0] if kind == "Dot" else [2.5, 1.3])
        want = single.predict_structures(test, return_std=True)
        gp = build(kind, 120, comm)
        gp.dist_chol_min_n = 0                          # the distributed factorisation also at this size
        gp.fit(show=False, opt=False)
        lml, g = gp.log_marginal_likelihood(theta.copy(), eval_gradient=True)
        gp.kernel.update([2.0, 2.0] if kind == "Dot" else [2.5, 1.3])
        got = gp.predict_structures(test, return_std=True)            # dealt over the ranks by structure
        assert abs(lml - lml1) < 1e-9 * abs(lml1) and np.abs(g - g1).max() < 1e-7 * np.abs(g1).max(), (kind, lml, lml1)
        assert np.abs(gp.alpha_ - single.alpha_).max() < 1e-7 * np.abs(single.alpha_).max()
        for k in range(len(test)):
            assert abs(got[0][k] - want[0][k]) < 1e-7 and np.abs(np.asarray(got[1][k]) - np.asarray(want[1][k])).max() < 1e-7
            assert abs(got[2][k] - want[2][k]) < 1e-7 and np.abs(np.asarray(got[3][k]) - np.asarray(want[3][k])).max() < 1e-7
    report["seconds"] = round(time.perf_counter() - t_all, 2)
    comm.barrier()
    if rank == 0:
        print(json.dumps({"rank_worker": "ok", **report}))
    if size > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
