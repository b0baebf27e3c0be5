"""GPU, BASELINE.json sizes (cfg3: N_train = 2000 force environments + 63 energies, N = 6063): size-independent
properties of the hot path, plus oracle parity on randomly drawn sub-blocks of the full matrices."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def model():
    import bench
    from gpr_calculator_amd.parallel import Comm
    gp = bench.build_model("RBF", 2000, Comm())
    gp.fit(show=False, opt=False)
    return gp


def test_factor_reconstructs_K_and_K_is_symmetric(model):
    import torch
    from gpr_calculator_amd import device as dev
    gp = model
    K = gp.kernel.k_total_dev(gp.train_x)
    N = K.shape[0]
    assert N == 6063
    scale = K.abs().max().item()
    assert (K - K.T).abs().max().item() / scale < 1e-13
    assert (torch.diagonal(K) > 0).all()
    NE = len(gp.train_x["energy"][-1])
    dev.add_diag_(K, NE, gp.noise_e ** 2, gp.noise_f ** 2)
    L = gp._L_dev
    assert torch.equal(torch.triu(L, 1), torch.zeros_like(L))
    rec = dev.gemm(L, L, transb=True)
    assert (rec - K).abs().max().item() / scale < 1e-13
    # alpha solves (K + noise) alpha = y
    res = dev.gemm(K, gp._alpha_dev) - dev.to_dev(gp.y_train)
    assert res.abs().max().item() < 1e-8 * max(1.0, np.abs(gp.y_train).max())
    # L^-1 L = I
    LiL = dev.trmm_lower(gp._Linv(), L)
    assert (LiL - torch.eye(N, dtype=torch.float64, device=L.device)).abs().max().item() < 1e-9


def test_random_subblocks_match_oracle(model, port_natives):
    from oracle import gp_numpy as og
    gp = model
    rng = np.random.default_rng(0)
    okern = og.OracleRBF(para=tuple(gp.kernel.parameters()), zeta=gp.kernel.zeta, natives=port_natives)
    K = gp.kernel.k_total(gp.train_x)
    X, dX, ELE, cnt = gp.train_x["force"]
    eX, eELE, ecnt = gp.train_x["energy"]
    off = np.concatenate(([0], np.cumsum(cnt)))
    eoff = np.concatenate(([0], np.cumsum(ecnt)))
    NE = len(ecnt)
    fi, fj = rng.choice(len(cnt), 5, replace=False), rng.choice(len(cnt), 6, replace=False)
    ei = rng.choice(NE, 3, replace=False)
    pick_f = lambda ids: (np.concatenate([X[off[i]:off[i + 1]] for i in ids]), np.concatenate([dX[off[i]:off[i + 1]] for i in ids]),
                          np.concatenate([ELE[off[i]:off[i + 1]] for i in ids]), [cnt[i] for i in ids])
    pick_e = lambda ids: (np.concatenate([eX[eoff[i]:eoff[i + 1]] for i in ids]),
                          np.concatenate([eELE[eoff[i]:eoff[i + 1]] for i in ids]), [ecnt[i] for i in ids])
    sub = okern.k_total({"energy": pick_e(ei), "force": pick_f(fi)}, {"energy": pick_e(ei), "force": pick_f(fj)})
    rows = np.concatenate((ei, np.concatenate([NE + 3 * i + np.arange(3) for i in fi])))
    cols = np.concatenate((ei, np.concatenate([NE + 3 * j + np.arange(3) for j in fj])))
    got = K[np.ix_(rows, cols)]
    assert np.abs(got - sub).max() / np.abs(sub).max() < 1e-12


def test_prediction_properties(model):
    from gpr_calculator_amd.synthetic import fcc_cell, neb_images, lj_energy_forces
    gp = model
    # training structures are reproduced to within the noise scale
    s0 = fcc_cell(seed=0)
    E0, F0 = lj_energy_forces(s0)
    E, F, _, E_std, F_std = gp.predict_structure(s0, stress=False, return_std=True)
    assert abs(E - E0) / len(s0) < 5 * gp.noise_e
    assert np.abs(F - F0).mean() < 0.5 * np.abs(F0).mean()          # fixed, un-optimised hyper-parameters
    # variances: non-negative, below the prior, small on training data and larger on unseen data
    prior = gp.kernel.diag(gp._structure_data(s0, gp.descriptor.calculate(s0))[0])
    assert E_std >= 0 and (F_std >= 0).all() and E_std ** 2 <= prior[0] and (F_std.reshape(-1) ** 2 <= prior[1:] + 1e-12).all()
    batch = neb_images(6, seed=300)
    Es, Fs, Es_std, Fs_std = gp.predict_structures(batch, return_std=True)
    far = fcc_cell(seed=900, sigma=0.45)                              # far from anything in the training set
    assert gp.predict_structure(far, stress=False, return_std=True)[4].mean() > 2 * F_std.mean()
    # batched path == one-by-one path
    for k in (0, 5):
        gp.device_single = False            # tuple route (host descriptor, packed sets) as the cross-check
        e1, f1, _, es1, fs1 = gp.predict_structure(batch[k], stress=False, return_std=True)
        gp.device_single = True
        assert abs(Es[k] - e1) < 1e-8 and np.abs(Fs[k] - f1).max() < 1e-8 and np.abs(Fs_std[k] - fs1).max() < 1e-7
    # linearity of the mean in the labels: doubling y doubles alpha and the prediction
    import torch
    from gpr_calculator_amd import device as dev
    a0 = gp._alpha_dev.clone()
    y2 = dev.to_dev(2.0 * gp.y_train).clone()
    dev.trsm_lower_(gp._L_dev, y2, dinv=gp._dinv_dev)
    dev.trsm_lower_(gp._L_dev, y2, trans=True, dinv=gp._dinv_dev)
    assert (y2 - 2 * a0).abs().max().item() <= 1e-12 * a0.abs().max().item()


def test_sharded_assembly_is_exact_at_full_size(model):
    import torch
    from gpr_calculator_amd import device as dev
    gp = model
    full = gp.kernel.k_total_dev(gp.train_x)
    acc = torch.zeros_like(full)
    try:
        for r in range(4):
            dev.set_shard(4, r)
            acc += gp.kernel.k_total_dev(gp.train_x)
    finally:
        dev.set_shard(1, 0)
    assert torch.equal(acc, full)


def test_dot_weight_space_fit_matches_dense_route_at_full_size():
    """cfg3 Dot model: the weight-space fit (Woodbury on the 466-wide feature Gram, N x N never formed) against the
    dense route (K + noise -> L -> alpha, then w and M from L) on a batch prediction; north-star tolerances"""
    import bench
    from gpr_calculator_amd.parallel import Comm
    from gpr_calculator_amd.synthetic import neb_images
    gp = bench.build_model("Dot", 2000, Comm())
    batch = neb_images(8, seed=500)
    gp.weight_space_fit = True
    gp.fit(show=False, opt=False)
    assert gp._dense_pending and gp._dense["L"] is None          # nothing N x N was factored
    Ew, Fw, Ews, Fws = gp.predict_structures(batch, return_std=True)
    assert gp._dense_pending
    gp.weight_space_fit = False
    gp.fit(show=False, opt=False)
    assert not gp._dense_pending and gp._dense["L"] is not None
    Ed, Fd, Eds, Fds = gp.predict_structures(batch, return_std=True)
    n = len(batch[0])
    for k in range(len(batch)):
        assert abs(Ew[k] - Ed[k]) < 1e-6 and np.abs(Fw[k] - Fd[k]).max() < 1e-5      # total energies, north-star bars
        assert abs(Ews[k] - Eds[k]) < 1e-6 and np.abs(Fws[k] - Fds[k]).max() < 1e-5
    # the lazy mirrors of a weight-space fit are the dense ones
    gp.weight_space_fit = True
    gp.fit(show=False, opt=False)
    a_lazy = gp.alpha_
    assert not gp._dense_pending and a_lazy.shape == (6063, 1)


def test_symmetries_of_the_native_prediction_at_full_size():
    """cfg3 Dot model, native batch call: size-independent properties of the whole device path (neighbour search ->
    descriptor -> features -> tail).  Batch order only permutes the results (bit for bit); a rigid translation (atoms
    re-wrapped into other periodic images) and a rigid rotation of cell and atoms leave E and sigma unchanged and
    rotate the forces."""
    import bench
    from gpr_calculator_amd.parallel import Comm
    from gpr_calculator_amd.neighbors import SimpleAtoms
    from gpr_calculator_amd.synthetic import neb_images
    gp = bench.build_model("Dot", 2000, Comm())
    gp.fit(show=False, opt=False)
    batch = neb_images(12, seed=700)
    E, F, Es, Fs = gp.predict_structures(batch, return_std=True)
    perm = np.random.default_rng(0).permutation(len(batch))
    Ep, Fp, Esp, Fsp = gp.predict_structures([batch[k] for k in perm], return_std=True)
    for a, k in enumerate(perm):
        assert Ep[a] == E[k] and Esp[a] == Es[k] and np.array_equal(Fp[a], F[k]) and np.array_equal(Fsp[a], Fs[k])
    # translation by a vector that moves atoms across cell boundaries
    s = batch[3]
    t = SimpleAtoms(s.numbers, s.positions + np.array([3.3, -7.9, 11.2]), s.cell, s.pbc)
    # rotation about an oblique axis (Rodrigues), applied to cell rows and positions alike
    ax = np.array([1.0, 2.0, -0.5]); ax /= np.linalg.norm(ax); th = 0.83
    Kx = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
    R = np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * Kx @ Kx
    r = SimpleAtoms(s.numbers, s.positions @ R.T, s.cell @ R.T, s.pbc)
    (Et, Er), (Ft, Fr), (Est, Esr), (Fst, Fsr) = gp.predict_structures([t, r], return_std=True)
    n = len(s)
    assert abs(Et - E[3]) / n < 1e-9 and np.abs(Ft - F[3]).max() < 1e-8 and abs(Est - Es[3]) < 1e-8 and np.abs(Fst - Fs[3]).max() < 1e-7
    assert abs(Er - E[3]) / n < 1e-9 and np.abs(Fr - F[3] @ R.T).max() < 1e-8 and abs(Esr - Es[3]) < 1e-8
    # the force variances are those of the rotated components: compare the rotation-invariant trace per atom
    assert np.abs((Fsr ** 2).sum(1) - (Fs[3] ** 2).sum(1)).max() < 1e-6 * max(1.0, (Fs[3] ** 2).sum(1).max())


def test_translation_and_rotation_invariance_rbf(model):
    """the same physics through the pair-kernel route (RBF, K_trans + triangular product)"""
    from gpr_calculator_amd.neighbors import SimpleAtoms
    from gpr_calculator_amd.synthetic import neb_images
    gp = model
    s = neb_images(2, seed=710)[1]
    ax = np.array([-0.3, 1.0, 0.7]); ax /= np.linalg.norm(ax); th = 2.1
    Kx = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
    R = np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * Kx @ Kx
    t = SimpleAtoms(s.numbers, s.positions + np.array([-5.1, 2.2, 9.7]), s.cell, s.pbc)
    r = SimpleAtoms(s.numbers, s.positions @ R.T, s.cell @ R.T, s.pbc)
    E, F, Es, Fs = gp.predict_structures([s, t, r], return_std=True)
    n = len(s)
    assert abs(E[1] - E[0]) / n < 1e-9 and np.abs(F[1] - F[0]).max() < 1e-8 and abs(Es[1] - Es[0]) < 1e-7
    assert abs(E[2] - E[0]) / n < 1e-9 and np.abs(F[2] - F[0] @ R.T).max() < 1e-8 and abs(Es[2] - Es[0]) < 1e-7


def test_cfg4_size_weight_space_model():
    """BASELINE config 4 sizes on one GPU (N_train = 8000 force environments + 250 energies, N = 24 250), Dot kernel:
    the weight-space fit never forms the 4.7 GB covariance; the model reproduces training structures within the noise
    scale, is more certain about them than about unseen ones, and its lazily factored dense mirrors (L, alpha) solve
    the same system"""
    import torch
    import bench
    from gpr_calculator_amd import device as dev
    from gpr_calculator_amd.parallel import Comm
    from gpr_calculator_amd.synthetic import fcc_cell, lj_energy_forces, neb_images
    gp = bench.build_model("Dot", 8000, Comm())
    gp.fit(show=False, opt=False)
    assert gp._dense_pending and gp._dense["L"] is None
    assert len(gp.y_train) == 24250
    s0 = fcc_cell(seed=0)
    E0, F0 = lj_energy_forces(s0)
    far = fcc_cell(seed=901, sigma=0.45)
    E, F, Es, Fs = gp.predict_structures([s0, far], return_std=True)
    assert abs(E[0] - E0) / len(s0) < 5 * gp.noise_e
    assert np.isfinite(F[0]).all() and (Fs[0] >= 0).all() and Fs[1].mean() > 2 * Fs[0].mean()
    # dense mirrors on demand: (K + noise) alpha = y with K = sigma^2 Psi Psi^T applied matrix-free
    alpha = gp._alpha_dev
    assert not gp._dense_pending and alpha.shape[0] == 24250
    species = gp._weight_space()[0]
    Psi = gp.kernel._psi(gp.train_x, species)
    NE = len(gp.train_x["energy"][-1])
    noise = torch.full((24250, 1), gp.noise_f ** 2, dtype=torch.float64, device=alpha.device)
    noise[:NE] = gp.noise_e ** 2
    lhs = gp.kernel.sigma ** 2 * dev.gemm(Psi, dev.gemm(Psi, alpha, transa=True)) + noise * alpha
    y = dev.to_dev(gp.y_train)
    assert (lhs - y).abs().max().item() < 1e-7 * max(1.0, y.abs().max().item())
