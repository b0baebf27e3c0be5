"""GPU, one device, several 'ranks' as threads (dist_helpers.ThreadComm stands where torch.distributed stands on a
multi-GPU node): the sharded fit of gpr_calculator_amd/distributed.py with the real device arithmetic --
row-slab covariance assembly + all-gather + mirror, the block-cyclic Cholesky on the potrf kernels, the row-sharded
weight-space fit / log-marginal likelihood -- against the single-rank results."""
import math

import numpy as np
import pytest
import torch

from dist_helpers import run_ranks

pytestmark = pytest.mark.gpu


def rel(a, b):
    return float((a - b).abs().max() / b.abs().max())


def build(kind, n_train, comm=None):
    from gpr_calculator_amd.gaussianprocess import GP
    from gpr_calculator_amd.SO3 import SO3
    from gpr_calculator_amd.kernels.Dot_mb import Dot_mb
    from gpr_calculator_amd.kernels.RBF_mb import RBF_mb
    from gpr_calculator_amd.synthetic import fcc_cell, lj_energy_forces
    kern = Dot_mb(para=[2.0, 2.0], zeta=2.0) if kind == "Dot" else RBF_mb(para=[2.5, 1.3], zeta=2.0)
    gp = GP(kernel=kern, descriptor=SO3(nmax=3, lmax=4, rcut=5.0), noise_e=0.002, noise_f=0.1, log_file="/tmp/gpr_test.log")
    if comm is not None:
        gp.comm, gp.rank, gp.size = comm, comm.rank, comm.size
    rows, left = [], n_train
    for s in range(math.ceil(n_train / 32)):
        a = fcc_cell(seed=s, species=(46, 78) if s % 3 == 0 else (46,))
        E, F = lj_energy_forces(a)
        take = min(32, left)
        rows.append((a, E, F, True, list(range(take))))
        left -= take
    gp.set_train_pts(gp.points_from_structures(rows))
    return gp


@pytest.mark.parametrize("size", [2, 3])
def test_block_cyclic_cholesky_on_the_device(size):
    from gpr_calculator_amd import device as dev
    from gpr_calculator_amd.distributed import DeviceOps, potrf_distributed
    torch.manual_seed(1)
    for n in (300, 769, 1516, 3000):
        B = torch.randn(n, n // 2 + 5, dtype=torch.float64, device=dev.device())
        A = B @ B.T + torch.eye(n, dtype=torch.float64, device=dev.device()) * (0.02 * n + 1.0)
        ref = A.clone()
        info, dinv_ref = dev.potrf_lower_(ref, want_dinv=True)
        outs = run_ranks(size, lambda comm: potrf_distributed(A.clone(), comm, DeviceOps()))
        for (L, dinv, info_d) in outs:
            assert int(info_d.item()) == 0
            assert rel(L, ref) < 1e-12, (n, size, rel(L, ref))
            assert float((dinv - dinv_ref).abs().max()) < 1e-10 * float(dinv_ref.abs().max())
            assert float(torch.triu(L, 1).abs().max()) == 0.0


@pytest.mark.parametrize("kind,size", [("RBF", 2), ("RBF", 3), ("Dot", 2)])
def test_sharded_covariance_assembly_matches_single_rank(kind, size):
    gp = build(kind, 150)
    gp.kernel.use_feature_map = False                        # Dot: the row-pair kernels (the feature route never forms K)
    K1 = gp.kernel.k_total_dev(gp.train_x)
    Kg = gp.kernel.k_total_with_grad_dev(gp.train_x)

    def rank_fn(comm):
        return gp.kernel.k_total_dev(gp.train_x, comm=comm), gp.kernel.k_total_with_grad_dev(gp.train_x, comm=comm)
    for (K, grad) in run_ranks(size, rank_fn):
        assert K.shape == K1.shape and rel(K, K1) < 1e-13
        assert bool(torch.equal(K, K.T))                     # symmetric by construction (mirrored)
        assert rel(grad[0], Kg[0]) < 1e-13 and rel(grad[2], Kg[2]) < 1e-13 * max(1.0, 1.0)


def test_sharded_assembly_at_cfg3_size():
    """N_train = 2000 (N = 6063), RBF, four ranks: the gathered slabs are the single-rank matrix"""
    gp = build("RBF", 2000)
    K1 = gp.kernel.k_total_dev(gp.train_x)
    outs = run_ranks(4, lambda comm: gp.kernel.k_total_dev(gp.train_x, comm=comm))
    for K in outs:
        assert K.shape == (6063, 6063) and rel(K, K1) < 1e-13
    assert bool(torch.equal(outs[0], outs[3]))


@pytest.mark.parametrize("kind", ["RBF", "Dot"])
def test_fit_and_lml_over_ranks_match_single_rank(kind):
    from gpr_calculator_amd.synthetic import fcc_cell
    single = build(kind, 120)
    single.fit(show=False, opt=False)
    theta = np.array([1.7, 1.1])
    lml1, g1 = single.log_marginal_likelihood(theta.copy(), eval_gradient=True)
    single.kernel.update([2.0, 2.0] if kind == "Dot" else [2.5, 1.3])
    test = fcc_cell(seed=77)
    E1, F1, _, s1, fs1 = single.predict_structure(test, stress=False, return_std=True)

    def rank_fn(comm):
        gp = build(kind, 120, comm)
        gp.dist_chol_min_n = 0                               # force the distributed factorisation on the dense route
        gp.fit(show=False, opt=False)
        lml, g = gp.log_marginal_likelihood(theta.copy(), eval_gradient=True)
        gp.kernel.update([2.0, 2.0] if kind == "Dot" else [2.5, 1.3])
        gp.device_single = False                             # several ranks: the tuple route of predict_structure
        E, F, _, s, fs = gp.predict_structure(test, stress=False, return_std=True)
        return lml, g, E, F, s, fs, gp.alpha_.copy()
    for (lml, g, E, F, s, fs, alpha) in run_ranks(2, rank_fn):
        assert abs(lml - lml1) < 1e-9 * abs(lml1) and np.abs(g - g1).max() < 1e-7 * np.abs(g1).max()
        assert abs(E - E1) < 1e-7 and np.abs(F - F1).max() < 1e-7 and abs(s - s1) < 1e-7 and np.abs(fs - fs1).max() < 1e-7
        assert np.abs(alpha - single.alpha_).max() < 1e-7 * np.abs(single.alpha_).max()


def test_cfg4_sharded_assembly_and_block_cyclic_cholesky_at_full_size():
    """BASELINE.json configs[3]: N_train = 8000 (N = 24 250) sharded 4-way.  Four thread ranks on the one device run the
    production schedule with the device arithmetic: (i) the block-cyclic factorisation -- owner factors, tile broadcast,
    panel solve, panel all-gather into a reused landing zone, device-resident row permutations, trailing update, factor
    gather -- against the single-GPU factor; (ii) the point-sharded covariance assembly -- every rank's row slabs
    computed in place in its copy of K, exchanged in place, mirrored -- against the single-rank matrix.  What is not
    covered here is the transport itself (RCCL over xGMI needs four GPUs; the gloo tests drive the same calls on CPU)."""
    from gpr_calculator_amd import device as dev
    from gpr_calculator_amd.distributed import DeviceOps, potrf_distributed
    n, size = 24250, 4
    d = dev.device()
    torch.manual_seed(0)
    B = torch.randn(n, 2048, dtype=torch.float64, device=d)
    A = B @ B.T
    A.diagonal().add_(0.05 * n)
    del B
    ref = A.clone()
    info, dinv_ref = dev.potrf_lower_(ref, want_dinv=True)
    assert int(info.item()) == 0
    outs = run_ranks(size, lambda comm: potrf_distributed(A, comm, DeviceOps()))
    for (L, dinv, info_d) in outs:
        assert int(info_d.item()) == 0
        assert rel(L, ref) < 1e-12
        assert float((dinv - dinv_ref).abs().max()) < 1e-10 * float(dinv_ref.abs().max())
    del outs, ref, A, dinv_ref
    torch.cuda.empty_cache()
    gp = build("RBF", 8000)
    assert len(gp.y_train) == n
    K1 = gp.kernel.k_total_dev(gp.train_x)
    errs = run_ranks(size, lambda comm: (rel(gp.kernel.k_total_dev(gp.train_x, comm=comm), K1)))
    assert max(errs) < 1e-13, errs
    del K1
    torch.cuda.empty_cache()
