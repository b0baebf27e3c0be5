"""GPU parity of GP.fit / predict / predict_structure(s) / log-marginal-likelihood against the oracle
(numpy/scipy restatement of gaussianprocess.py on top of the plain-C natives) on seeded synthetic
structures.  Tolerances from BASELINE.json north_star: energies 1e-6 eV, forces 1e-5 eV/A, and the
same for the predictive standard deviations."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
E_TOL, F_TOL = 1e-6, 1e-5


def oracle_descriptor():
    from oracle.so3_numpy import so3_calculate
    from gpr_calculator_amd.neighbors import symbol_of

    class D:
        def calculate(self, atoms, use_mpi=False):
            out = so3_calculate(atoms.numbers, atoms.positions, atoms.cell, atoms.pbc, 3, 4, 5.0, 2.0)
            out["elements"] = [symbol_of(z) for z in atoms.numbers]
            return out
    return D()


def training_structures(n, species=(46,), natoms_reps=1):
    from gpr_calculator_amd.synthetic import fcc_cell, lj_energy_forces
    out = []
    for s in range(n):
        atoms = fcc_cell(seed=s, reps=natoms_reps, species=species, sigma=0.08)
        E, F = lj_energy_forces(atoms)
        out.append((atoms, E, F))
    return out


def oracle_train(kernel_o, strucs, force_ids):
    """training dictionary through the oracle descriptor, reference packing"""
    from oracle.gp_numpy import list_to_tuple
    from gpr_calculator_amd.utilities import species_numbers
    des = oracle_descriptor()
    energy, force, yE, yF = [], [], [], []
    for (atoms, E, F) in strucs:
        d = des.calculate(atoms)
        ele = species_numbers(d["elements"])
        energy.append((d["x"], ele)); yE.append(E / len(atoms))
        for a in force_ids:
            ids = np.argwhere(d["seq"][:, 1] == a).flatten()
            ci = d["seq"][ids, 0]
            force.append((d["x"][ci], d["dxdr"][ids], ele[ci])); yF.append(F[a])
    train_x = {"energy": list_to_tuple(energy, mode="energy"), "force": list_to_tuple(force)}
    y = np.concatenate((np.array(yE), np.array(yF).reshape(-1))).reshape(-1, 1)
    return train_x, y, len(energy)


@pytest.mark.parametrize("kind", ["RBF", "Dot"])
def test_fit_predict_structure_matches_oracle(kind, port_natives):
    from gpr_calculator_amd.gaussianprocess import GP
    from gpr_calculator_amd.SO3 import SO3
    from gpr_calculator_amd.kernels.RBF_mb import RBF_mb
    from gpr_calculator_amd.kernels.Dot_mb import Dot_mb
    from gpr_calculator_amd.synthetic import fcc_cell
    from oracle import gp_numpy as og
    strucs = training_structures(4, species=(46, 78))          # 4-atom cells, two species
    force_ids = [0, 2, 3]
    noise_e, noise_f = 0.002, 0.1
    if kind == "RBF":
        kern, okern = RBF_mb(para=[2.5, 1.3], zeta=2.0), og.OracleRBF(para=(2.5, 1.3), zeta=2.0, natives=port_natives)
    else:
        kern, okern = Dot_mb(para=[2.0, 2.0], zeta=2.0), og.OracleDot(para=(2.0, 2.0), zeta=2.0, natives=port_natives)
    gp = GP(kernel=kern, descriptor=SO3(nmax=3, lmax=4, rcut=5.0), noise_e=noise_e, noise_f=noise_f, log_file="/tmp/gpr_test.log")
    rows = [(a, E, F, True, force_ids) for (a, E, F) in strucs]
    gp.fit(gp.points_from_structures(rows), show=False, opt=False)

    train_x, y, NE = oracle_train(okern, strucs, force_ids)
    assert np.abs(gp.y_train - y).max() < 1e-12
    K = okern.k_total(train_x)
    L, alpha, K_inv = og.fit_core(K, y, NE, noise_e, noise_f)
    assert np.abs(gp.L_ - L).max() / np.abs(L).max() < 1e-10
    assert np.abs(gp.alpha_ - alpha).max() / np.abs(alpha).max() < 1e-7     # cond(K) amplifies rounding
    assert np.abs(gp._K_inv - K_inv).max() / np.abs(K_inv).max() < 1e-7

    # predict a new structure with one fixed atom, E/F/sigma: tuple route (host descriptor, packed sets) first
    gp.device_single = False
    new = fcc_cell(seed=77, reps=1, species=(46, 78), sigma=0.08)
    new.fixed = np.array([1])
    E, F, S, E_std, F_std = gp.predict_structure(new, stress=False, return_std=True, f_tol=1e-8)
    gp.device_single = True            # ... and the device-resident single-structure route gives the same numbers
    Ed, Fd, _, Ed_std, Fd_std = gp.predict_structure(new, stress=False, return_std=True, f_tol=1e-8)
    assert abs(Ed - E) < 1e-9 and np.abs(Fd - F).max() < 1e-9 and abs(Ed_std - E_std) < 1e-8 and np.abs(Fd_std - F_std).max() < 1e-8
    gp.device_single = False
    des = oracle_descriptor()
    d = des.calculate(new)
    from gpr_calculator_amd.utilities import species_numbers
    ele = species_numbers(d["elements"])
    free = [0, 2, 3]
    data = {"energy": og.list_to_tuple([(d["x"], ele)], mode="energy"), "force": []}
    for a in free:
        ids = np.argwhere(d["seq"][:, 1] == a).flatten()
        ci = d["seq"][ids, 0]
        data["force"].append((d["x"][ci], d["dxdr"][ids], ele[ci]))
    K_trans = okern.k_total(data, train_x, 1e-8) if kind == "RBF" else okern.k_total(data, train_x)
    mean, std = og.predict_core(K_trans, alpha, K_inv, okern.diag(data))
    assert abs(E - mean[0] * len(new)) < E_TOL
    assert np.abs(F[free] - mean[1:].reshape(-1, 3)).max() < F_TOL and np.all(F[1] == 0)
    assert abs(E_std - std[0]) < E_TOL
    assert np.abs(F_std[free] - std[1:].reshape(-1, 3)).max() < F_TOL and S is None

    if kind == "Dot":       # the fit above ran in weight space; the dense route gives the same predictions
        gp.weight_space_fit = False
        gp.fit(show=False, opt=False)
        Ef, Ff, _, Ef_std, Ff_std = gp.predict_structure(new, stress=False, return_std=True, f_tol=1e-8)
        assert abs(Ef - E) < E_TOL and np.abs(Ff - F).max() < F_TOL
        assert abs(Ef_std - E_std) < E_TOL and np.abs(Ff_std - F_std).max() < F_TOL
    if kind == "Dot":       # weight-space mean (Psi* w) and training-point mean (K_trans alpha) are the same number
        gp.weight_space_mean = False
        E2, F2, _ = gp.predict_structure(new, stress=False, return_std=False, f_tol=1e-8)
        gp.weight_space_mean = True
        assert abs(E2 - E) < 1e-8 and np.abs(F2 - F).max() < 1e-8

    # batched predict of several structures equals the one-by-one path
    batch = [fcc_cell(seed=80 + k, reps=1, species=(46, 78), sigma=0.08) for k in range(3)]
    batch[1].fixed = np.array([0, 3])
    Es, Fs, Es_std, Fs_std = gp.predict_structures(batch, return_std=True, f_tol=1e-8)
    for k, s in enumerate(batch):
        e1, f1, _, es1, fs1 = gp.predict_structure(s, stress=False, return_std=True, f_tol=1e-8)
        assert abs(Es[k] - e1) < 1e-9 and np.abs(Fs[k] - f1).max() < 1e-9
        assert abs(Es_std[k] - es1) < 1e-8 and np.abs(Fs_std[k] - fs1).max() < 1e-8

    # validate_data on the training set = K alpha of the oracle
    Et, Ep, Ft, Fp = gp.validate_data()
    pred_o = (K @ alpha)[:, 0]
    assert np.abs(Ep - pred_o[:NE]).max() < E_TOL and np.abs(Fp - pred_o[NE:]).max() < F_TOL
    assert np.abs(Et - y[:NE, 0]).max() == 0 and np.abs(Ft - y[NE:, 0]).max() == 0


def test_log_marginal_likelihood_and_gradient_match_oracle(port_natives):
    from gpr_calculator_amd.gaussianprocess import GP
    from gpr_calculator_amd.SO3 import SO3
    from gpr_calculator_amd.kernels.RBF_mb import RBF_mb
    from oracle import gp_numpy as og
    strucs = training_structures(3, species=(46,))
    force_ids = [0, 1, 2, 3]
    gp = GP(kernel=RBF_mb(para=[1.0, 0.5], zeta=2.0), descriptor=SO3(nmax=3, lmax=4, rcut=5.0), noise_e=0.002,
            noise_f=0.1, log_file="/tmp/gpr_test.log")
    gp.set_train_pts(gp.points_from_structures([(a, E, F, True, force_ids) for (a, E, F) in strucs]))
    okern = og.OracleRBF(para=(1.0, 0.5), zeta=2.0, natives=port_natives)
    train_x, y, NE = oracle_train(okern, strucs, force_ids)
    for params in ([1.0, 0.5], [2.2, 1.7]):
        lml, grad = gp.log_marginal_likelihood(np.array(params), eval_gradient=True)
        lml_o, grad_o = og.log_marginal_likelihood(okern, train_x, y, NE, 0.002, 0.1, np.array(params))
        assert abs(lml - lml_o) < 1e-7 * max(1.0, abs(lml_o))
        assert np.abs(grad - grad_o).max() < 1e-6 * max(1.0, np.abs(grad_o).max())
    # and the optimiser runs end to end on the device-evaluated objective
    gp.fit(show=False, opt=True, maxiter=3)
    assert gp.alpha_ is not None and gp.N_queue == 0 and gp.fits == 1


def test_add_structure_queue_semantics():
    from gpr_calculator_amd.gaussianprocess import GP
    from gpr_calculator_amd.SO3 import SO3
    from gpr_calculator_amd.kernels.RBF_mb import RBF_mb
    strucs = training_structures(3, species=(46,))
    gp = GP(kernel=RBF_mb(para=[1.0, 0.5], zeta=2.0), descriptor=SO3(nmax=3, lmax=4, rcut=5.0), noise_e=0.002,
            noise_f=0.1, log_file="/tmp/gpr_test.log")
    # untrained model: every atom is a candidate, de-duplicated by cosine similarity (A.17)
    pts, n_pts, _ = gp.add_structure(strucs[0])
    assert n_pts >= 2 and gp.N_energy == 1 and gp.N_queue == n_pts
    gp.fit(show=False, opt=False)
    assert gp.N_queue == 0
    n_before = gp.alpha_.shape[0]
    gp.add_structure(strucs[1])
    # queued points are invisible to prediction until the next fit (A.16)
    tx = gp.get_train_x()
    assert len(tx["energy"][-1]) == 1 and gp.alpha_.shape[0] == n_before
    E, F, _ = gp.predict_structure(strucs[2][0], stress=False)
    assert np.isfinite(E) and np.isfinite(F).all()
    gp.fit(show=False, opt=False)
    assert gp.alpha_.shape[0] > n_before


def test_dot_direct_features_match_packed_features_and_diag():
    """gpr_dot2_features_direct (prediction path: no packed sets) against gpr_dot2_features on packed sets and
    the pair-sum diag kernels, on a two-species batch with fixed atoms"""
    import torch
    from gpr_calculator_amd.gaussianprocess import GP
    from gpr_calculator_amd.SO3 import SO3
    from gpr_calculator_amd.kernels.Dot_mb import Dot_mb
    from gpr_calculator_amd.synthetic import fcc_cell
    kern = Dot_mb(para=[1.7, 0.9], zeta=2.0)
    gp = GP(kernel=kern, descriptor=SO3(nmax=3, lmax=4, rcut=5.0), log_file="/tmp/gpr_test.log")
    batch = [fcc_cell(seed=s, species=(46, 78) if s % 2 else (46,)) for s in range(4)]
    batch[2].fixed = np.array([1, 5, 6])
    X, frees = gp._batch_data_dev(batch)
    species = (46, 78)
    P_direct, diag_direct = kern.psi_diag_dev(X, species, want_diag=True)
    P_packed = kern._psi(X, species)
    diag_packed = kern.diag_dev(X)
    scale = float(P_packed.abs().max())
    assert float((P_direct - P_packed).abs().max()) < 1e-13 * scale
    assert float((diag_direct - diag_packed).abs().max()) < 1e-12 * float(diag_packed.abs().max())
    assert frees[0] is None and len(frees[2]) == len(batch[2]) - 3


def test_predict_structure_with_stress_matches_oracle(port_natives):
    """predict_structure(stress=True): SO3 stress rows (rdxdr) on the GPU -> 9-wide force rows -> K_trans1 alpha
    (gaussianprocess.py:846-918, RBF_mb.py:206-229) against the oracle descriptor + the plain-C stress natives"""
    from gpr_calculator_amd.gaussianprocess import GP
    from gpr_calculator_amd.SO3 import SO3
    from gpr_calculator_amd.kernels.RBF_mb import RBF_mb
    from gpr_calculator_amd.synthetic import fcc_cell
    from gpr_calculator_amd.utilities import species_numbers
    from gpr_calculator_amd.neighbors import symbol_of
    from oracle import gp_numpy as og
    from oracle.so3_numpy import so3_calculate
    strucs = training_structures(3, species=(46, 78))
    force_ids = [0, 1, 3]
    gp = GP(kernel=RBF_mb(para=[2.5, 1.3], zeta=2.0), descriptor=SO3(nmax=3, lmax=4, rcut=5.0, stress=True),
            noise_e=0.002, noise_f=0.1, log_file="/tmp/gpr_test.log")
    gp.fit(gp.points_from_structures([(a, E, F, True, force_ids) for (a, E, F) in strucs]), show=False, opt=False)
    new = fcc_cell(seed=91, reps=1, species=(46, 78), sigma=0.08)
    E, F, S = gp.predict_structure(new, stress=True, return_std=False, f_tol=1e-10)
    assert S.shape == (len(new), 6)

    okern = og.OracleRBF(para=(2.5, 1.3), zeta=2.0, natives=port_natives)
    train_x, y, NE = oracle_train(okern, strucs, force_ids)
    K = okern.k_total(train_x)
    _, alpha, _ = og.fit_core(K, y, NE, 0.002, 0.1)
    d = so3_calculate(new.numbers, new.positions, new.cell, new.pbc, 3, 4, 5.0, 2.0, stress=True)
    ele = species_numbers([symbol_of(z) for z in new.numbers])
    energy = og.list_to_tuple([(d["x"], ele)], mode="energy")
    force9 = []
    for a in range(len(new)):                                   # every atom carries stress rows (gaussianprocess.py:852-870)
        ids = np.argwhere(d["seq"][:, 1] == a).flatten()
        ci = d["seq"][ids, 0]
        rd = d["rdxdr"][ids].reshape(len(ids), d["x"].shape[1], 9)[:, :, [0, 4, 8, 1, 2, 5]]
        force9.append((d["x"][ci], np.concatenate((d["dxdr"][ids], rd), axis=2), ele[ci]))
    W = okern.w
    f9 = og.list_to_tuple(force9, stress=True)
    C_fe, C_se = W.rbf_kef(train_x["energy"], f9, 2.5, 1.3, 2.0, stress=True, transpose=True)
    C_ff, C_sf = W.rbf_kff(f9, train_x["force"], 2.5, 1.3, 2.0, stress=True, tol=1e-10)
    S_ref = (np.hstack((C_se, C_sf)) @ alpha).reshape(len(new), 6)
    F_ref = (np.hstack((C_fe, C_ff)) @ alpha).reshape(len(new), 3)
    assert np.abs(F - F_ref).max() < F_TOL
    assert np.abs(S - S_ref).max() < 1e-5 * max(1.0, np.abs(S_ref).max())


def test_dot_weight_space_lml_matches_dense_route_and_oracle(port_natives):
    """log marginal likelihood and gradient of the Dot kernel (zeta = 2) through the F x F weight-space system
    against the dense N x N route on the device and against the oracle (fixed noise), plus the noise column"""
    from gpr_calculator_amd.gaussianprocess import GP
    from gpr_calculator_amd.SO3 import SO3
    from gpr_calculator_amd.kernels.Dot_mb import Dot_mb
    from oracle import gp_numpy as og
    strucs = training_structures(3, species=(46, 78))
    force_ids = [0, 1, 3]
    gp = GP(kernel=Dot_mb(para=[2.0, 1.5], zeta=2.0), descriptor=SO3(nmax=3, lmax=4, rcut=5.0), noise_e=0.002,
            noise_f=0.1, log_file="/tmp/gpr_test.log")
    gp.set_train_pts(gp.points_from_structures([(a, E, F, True, force_ids) for (a, E, F) in strucs]))
    okern = og.OracleDot(para=(2.0, 1.5), zeta=2.0, natives=port_natives)
    train_x, y, NE = oracle_train(okern, strucs, force_ids)
    for params in ([2.0, 1.5], [0.9, 2.4]):
        gp.weight_space_fit = True
        lml_w, grad_w = gp.log_marginal_likelihood(np.array(params), eval_gradient=True)
        assert abs(gp.log_marginal_likelihood(np.array(params)) - lml_w) < 1e-9 * max(1.0, abs(lml_w))
        gp.weight_space_fit = False
        lml_d, grad_d = gp.log_marginal_likelihood(np.array(params), eval_gradient=True)
        lml_o, grad_o = og.log_marginal_likelihood(okern, train_x, y, NE, 0.002, 0.1, np.array(params))
        assert abs(lml_w - lml_d) < 1e-8 * max(1.0, abs(lml_d)) and abs(lml_w - lml_o) < 1e-7 * max(1.0, abs(lml_o))
        assert np.abs(grad_w - grad_d).max() < 1e-6 * max(1.0, np.abs(grad_d).max())
        assert np.abs(grad_w - grad_o).max() < 1e-6 * max(1.0, np.abs(grad_o).max())
    # noise as a third parameter (noise_bounds set): same objective and gradient on both routes
    gp.noise_bounds = [1e-3, 1e-1]
    p3 = np.array([1.3, 0.8, 0.004])
    gp.weight_space_fit = True
    lml_w, grad_w = gp.log_marginal_likelihood(p3, eval_gradient=True)
    gp.weight_space_fit = False
    lml_d, grad_d = gp.log_marginal_likelihood(p3, eval_gradient=True)
    assert len(grad_w) == 3 and abs(lml_w - lml_d) < 1e-8 * max(1.0, abs(lml_d))
    assert np.abs(grad_w - grad_d).max() < 1e-6 * max(1.0, np.abs(grad_d).max())
    gp.weight_space_fit = True
    gp.fit(show=False, opt=True, maxiter=3)                  # L-BFGS-B on the weight-space objective
    assert gp._fitted and gp._dense_pending


def test_validation_on_the_training_set_needs_no_second_covariance_assembly():
    """validate_data() with no arguments right after a fit -- what the calculator does after every refit -- takes
    K alpha = y - D alpha instead of predicting at the training points.  Same numbers as the predicting route (forced by
    making the shortcut's precondition fail), after a plain fit, an optimised one and an appended one; with points in the
    queue the shortcut must step aside."""
    from gpr_calculator_amd.gaussianprocess import GP
    from gpr_calculator_amd.SO3 import SO3
    from gpr_calculator_amd.kernels.RBF_mb import RBF_mb
    strucs = training_structures(5, species=(46, 78))
    gp = GP(kernel=RBF_mb(para=[1.5, 0.8], zeta=2.0), descriptor=SO3(nmax=3, lmax=4, rcut=5.0), noise_e=0.002,
            noise_f=0.1, log_file="/tmp/gpr_test.log")
    for st in strucs[:3]:
        gp.add_structure(st)

    def both():
        assert gp._train_predictions() is not None
        quick = gp.validate_data(show=True)
        err_quick = dict(gp.error)
        saved, gp._dense_params = gp._dense_params, None          # precondition fails -> the predicting route
        assert gp._train_predictions() is None
        full = gp.validate_data(show=True)
        gp._dense_params = saved
        for a, b in zip(quick, full):
            assert np.abs(np.asarray(a) - np.asarray(b)).max() < 1e-9 * max(1.0, np.abs(np.asarray(b)).max())
        for k in err_quick:
            assert abs(err_quick[k] - gp.error[k]) < 1e-8
    gp.fit(show=False, opt=False)
    both()
    gp.fit(show=False, opt=True, maxiter=3)
    both()
    gp.add_structure(strucs[3])
    assert gp.N_queue > 0 and gp._train_predictions() is None    # queued points: the factor does not cover the training set
    gp.fit(show=False, opt=False)                                 # (appended to the factor, rows permuted behind the scenes)
    both()


@pytest.mark.parametrize("kind", ["RBF", "Dot"])
def test_add_structure_judged_on_the_device_queues_what_the_host_route_queues(kind):
    """A fitted model judges a labelled structure in one device pass and brings only the picked force points to the host
    (GP._judge_on_device); the route through convert_train_data / validate_data builds every atom's force point on the host
    first.  Same picked atoms, same queued arrays, same error report -- with fixed atoms present (both routes ignore them,
    utilities.py:97-129) and two species."""
    from gpr_calculator_amd.gaussianprocess import GP
    from gpr_calculator_amd.SO3 import SO3
    from gpr_calculator_amd.kernels.Dot_mb import Dot_mb
    from gpr_calculator_amd.kernels.RBF_mb import RBF_mb
    strucs = training_structures(6, species=(46, 78))
    strucs[4][0].fixed = np.array([0, 2])
    out = {}
    for route in (True, False):
        kern = RBF_mb(para=[1.5, 0.8], zeta=2.0) if kind == "RBF" else Dot_mb(para=[2.0, 2.0], zeta=2.0)
        gp = GP(kernel=kern, descriptor=SO3(nmax=3, lmax=4, rcut=5.0), noise_e=0.002, noise_f=0.05, log_file="/tmp/gpr_test.log")
        gp.device_judge = route
        for st in strucs[:3]:
            gp.add_structure((st[0], st[1], st[2].copy()))
        gp.fit(show=False, opt=False)
        res = [gp.add_structure((st[0], st[1], st[2].copy()), N_max=3 + k) for k, st in enumerate(strucs[3:])]
        out[route] = (gp, res)
    (ga, ra), (gb, rb) = out[True], out[False]
    for (pa, na, ea), (pb, nb, eb) in zip(ra, rb):
        assert na == nb and pa["db"][0][4] == pb["db"][0][4] and len(pa["db"][0][4]) >= 1          # the same atoms picked
        for a, b in zip(ea, eb):
            assert np.abs(np.asarray(a) - np.asarray(b)).max() < 1e-9
        for fa, fb in zip(pa["force"], pb["force"]):
            for u, v in zip(fa, fb):
                assert np.asarray(u).shape == np.asarray(v).shape and np.abs(np.asarray(u) - np.asarray(v)).max() < 1e-12
    for key in ("energy", "force"):
        for u, v in zip(ga.train_x[key], gb.train_x[key]):
            assert np.abs(np.asarray(u, dtype=np.float64) - np.asarray(v, dtype=np.float64)).max() < 1e-12
    assert np.abs(ga.y_train - gb.y_train).max() == 0.0 and ga.N_queue == gb.N_queue > 0
