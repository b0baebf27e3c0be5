"""GPU parity of the GP layer against fixtures produced by the reference's OWN Python + C++
(tests/golden/gp_*.npz, see tests/golden/make_golden_gp.py): wrappers, kernel objects, fit, predict_structure,
log-marginal likelihood, add_structure / queue / refit.  Tolerances of BASELINE.json's north_star: energies 1e-6 eV,
forces 1e-5 eV/A, the same for the standard deviations; blocks 1e-12 relative."""
import numpy as np
import pytest

from gp_golden import (MODEL_CASES, atoms, explicit_inverse_noise, load, predicted_data, rel, split_std, stable_std,
                       train_x, training_rows)

pytestmark = pytest.mark.gpu
E_TOL, F_TOL = 1e-6, 1e-5


def check_std(E_std, F_std, z, k, L_key="L", Kt_key=None, tag="test"):
    """predictive standard deviations against the reference: (1) at the north-star tolerance against the reference's
    own L / K_trans / diag evaluated through a triangular solve, (2) against the std the reference recorded, within the
    rounding noise of its explicit-inverse formula (gp_golden.explicit_inverse_noise)"""
    n, fixed = len(z[f"test{k}_numbers"]), z[f"test{k}_fixed"]
    diag = z[f"test{k}_diag"]
    Es, Fs = split_std(stable_std(z[L_key], z[Kt_key or f"test{k}_K_trans"], diag), n, fixed)
    assert abs(E_std - Es) < E_TOL and np.abs(F_std - Fs).max() < F_TOL, (tag, k)
    dvar = explicit_inverse_noise(z[L_key], diag)
    ref_E, ref_F = float(z[f"{tag}{k}_E_std" if tag == "test" else f"{tag}_E_std"]), z[f"{tag}{k}_F_std" if tag == "test" else f"{tag}_F_std"]
    assert abs(E_std ** 2 - ref_E ** 2) < max(dvar, 2 * E_TOL * ref_E), (tag, k, E_std, ref_E, dvar)
    assert np.abs(F_std ** 2 - ref_F ** 2).max() < max(dvar, 2 * F_TOL * np.abs(ref_F).max())


def make_gp(z, stress=False):
    from gpr_calculator_amd.gaussianprocess import GP
    from gpr_calculator_amd.SO3 import SO3
    from gpr_calculator_amd.kernels.RBF_mb import RBF_mb
    from gpr_calculator_amd.kernels.Dot_mb import Dot_mb
    cls = RBF_mb if str(z["kind"]) == "RBF" else Dot_mb
    kern = cls(para=[float(v) for v in z["para"]], zeta=float(z["zeta"]))
    des = SO3(nmax=int(z["nmax"]), lmax=int(z["lmax"]), rcut=float(z["rcut"]), alpha=float(z["des_alpha"]), stress=stress)
    return GP(kernel=kern, descriptor=des, noise_e=float(z["noise_e"]), noise_f=float(z["noise_f"]),
              log_file="/tmp/gpr_test.log")


def test_wrappers_match_reference():
    """kee_C / kef_C / kff_C (rbf_kernel.py:7-337, dot_kernel.py:9-270): value, hyper-gradient, stress, transpose"""
    from gpr_calculator_amd.kernels import rbf_kernel as r, dot_kernel as d
    z = load("wrappers")
    c1, c2 = z["counts1"].tolist(), z["counts2"].tolist()
    XE1, XE2 = (z["x1"], z["ele1"], c1), (z["x2"], z["ele2"], c2)
    XF1, XF2 = (z["x1"], z["dx1dr"], z["ele1"], c1), (z["x2"], z["dx2dr"], z["ele2"], c2)
    XF1s = (z["x1"], z["dx1dr9"], z["ele1"], c1)
    s, l, s0 = float(z["sigma"]), float(z["l"]), float(z["sigma0"])
    for zeta in (2.0, 3.0):
        t = f"z{int(zeta)}"
        got = {
            f"rbf_kee_{t}": r.kee_C(XE1, XE2, s, l, zeta),
            f"rbf_kee_grad_{t}": np.stack(r.kee_C(XE1, XE2, s, l, zeta, grad=True)),
            f"rbf_kef_{t}": r.kef_C(XE1, XF2, s, l, zeta),
            f"rbf_kef_grad_{t}": np.stack(r.kef_C(XE1, XF2, s, l, zeta, grad=True)),
            f"rbf_kef_T_{t}": r.kef_C(XE1, XF2, s, l, zeta, transpose=True),
            f"rbf_kff_{t}": r.kff_C(XF1, XF2, s, l, zeta, tol=1e-12),
            f"rbf_kff_grad_{t}": np.stack(r.kff_C(XF1, XF2, s, l, zeta, grad=True)),
            f"dot_kee_{t}": d.kee_C(XE1, XE2, s, s0, zeta),
            f"dot_kee_grad_{t}": np.stack(d.kee_C(XE1, XE2, s, s0, zeta, grad=True)),
            f"dot_kef_{t}": d.kef_C(XE1, XF2, s, s0, zeta),
            f"dot_kef_grad_{t}": np.stack(d.kef_C(XE1, XF2, s, s0, zeta, grad=True)),
            f"dot_kef_T_{t}": d.kef_C(XE1, XF2, s, s0, zeta, transpose=True),
            f"dot_kff_{t}": d.kff_C(XF1, XF2, s, s0, zeta),
            f"dot_kff_grad_{t}": np.stack(d.kff_C(XF1, XF2, s, s0, zeta, grad=True)),
        }
        a, b = r.kef_C(XE2, XF1s, s, l, zeta, stress=True, transpose=True)
        got[f"rbf_kef_stress_T_{t}"], got[f"rbf_kef_stress_T1_{t}"] = a, b
        a, b = r.kff_C(XF1s, XF2, s, l, zeta, stress=True, tol=1e-12)
        got[f"rbf_kff_stress_{t}"], got[f"rbf_kff_stress1_{t}"] = a, b
        a, b = d.kef_C(XE2, XF1s, s, s0, zeta, stress=True, transpose=True)
        got[f"dot_kef_stress_T_{t}"], got[f"dot_kef_stress_T1_{t}"] = a, b
        a, b = d.kff_C(XF1s, XF2, s, s0, zeta, stress=True)
        got[f"dot_kff_stress_{t}"], got[f"dot_kff_stress1_{t}"] = a, b
        for k, v in got.items():
            assert v.shape == z[k].shape, (k, v.shape, z[k].shape)
            assert np.abs(v - z[k]).max() <= 1e-12 * max(np.abs(z[k]).max(), 1e-300) + 0.0, (k, rel(v, z[k]))


@pytest.mark.parametrize("name", MODEL_CASES)
def test_kernel_objects_match_reference(name):
    """k_total / k_total_with_grad / k_total_with_stress / diag of RBF_mb and Dot_mb on the reference's packed data"""
    from gpr_calculator_amd.utilities import tuple_to_list
    z = load(name)
    kern, tx = make_gp(z).kernel, train_x(z)
    for fm in ((True, False) if str(z["kind"]) == "Dot" else (None,)):
        if fm is not None:
            kern.use_feature_map = fm                        # Dot: feature-map route and row-pair route
        assert rel(kern.k_total(tx), z["K"]) < 1e-12
        K, dK = kern.k_total_with_grad(tx)
        assert rel(K, z["K_with_grad"]) < 1e-12
        for j in range(2):
            assert np.abs(dK[:, :, j] - z["dK"][:, :, j]).max() <= 1e-12 * max(1.0, np.abs(z["dK"][:, :, j]).max())
        for k in range(int(z["n_tests"])):
            data = predicted_data(z, k)
            assert rel(kern.k_total(data, tx, 1e-12), z[f"test{k}_K_trans"]) < 1e-12
            lists = {"energy": data["energy"], "force": tuple_to_list(data["force"])}
            assert rel(kern.diag(lists), z[f"test{k}_diag"]) < 1e-12
            assert rel(kern.diag(data), z[f"test{k}_diag"]) < 1e-12
            if f"test{k}_stress_K_trans" in z.files:
                Kt, Kt1 = kern.k_total_with_stress(predicted_data(z, k, stress=True), tx, 1e-12)
                assert rel(Kt, z[f"test{k}_stress_K_trans"]) < 1e-12
                assert rel(Kt1, z[f"test{k}_stress_K_trans1"]) < 1e-12
    assert rel(kern.diag({"energy": tx["energy"], "force": tuple_to_list(tx["force"])}), z["diag_train"]) < 1e-12


@pytest.mark.parametrize("name", MODEL_CASES)
def test_fit_and_predict_structure_match_reference(name):
    """structures -> device SO3 -> packing -> K -> L, alpha, K^-1 -> predict_structure E / F / std, end to end"""
    z = load(name)
    gp = make_gp(z)
    gp.fit(gp.points_from_structures(training_rows(z)), show=False, opt=False)
    assert np.abs(gp.y_train - z["y_train"]).max() < 1e-12
    tx = train_x(z)
    for key in ("energy", "force"):                              # packing (a16) and descriptor (a15) vs the reference's
        assert list(gp.train_x[key][-1]) == list(tx[key][-1])
        assert rel(gp.train_x[key][0], tx[key][0]) < 1e-10
    assert rel(gp.train_x["force"][1], tx["force"][1]) < 1e-10
    assert rel(gp.L_, z["L"]) < 1e-8
    assert rel(gp.alpha_, z["alpha"]) < 1e-5                     # cond(K + noise) amplifies the 1e-11 of the descriptor
    assert rel(gp._K_inv, z["K_inv"]) < 1e-5
    routes = [dict(device_single=True), dict(device_single=False)]
    if str(z["kind"]) == "Dot":
        routes.append(dict(device_single=False, weight_space_fit=False))
    for route in routes:
        for k, v in route.items():
            setattr(gp, k, v)
        if "weight_space_fit" in route:
            gp.fit(show=False, opt=False)                        # dense N x N factorisation instead of weight space
        for k in range(int(z["n_tests"])):
            t = atoms(z, f"test{k}", fixed=z[f"test{k}_fixed"])
            E, F, S, E_std, F_std = gp.predict_structure(t, stress=False, return_std=True, f_tol=1e-12)
            assert abs(E - float(z[f"test{k}_E"])) < E_TOL, (route, k)
            assert np.abs(F - z[f"test{k}_F"]).max() < F_TOL
            check_std(E_std, F_std, z, k)
            assert S is None and np.all(F[z[f"test{k}_fixed"]] == 0) and np.all(F_std[z[f"test{k}_fixed"]] == 0)
            E2, F2, _ = gp.predict_structure(t, stress=False, return_std=False, f_tol=1e-12)
            assert abs(E2 - E) < 1e-9 and np.abs(F2 - F).max() < 1e-9
    # batch call = one-by-one
    ts = [atoms(z, f"test{k}", fixed=z[f"test{k}_fixed"]) for k in range(int(z["n_tests"]))]
    Es, Fs, Es_std, Fs_std = gp.predict_structures(ts, return_std=True, f_tol=1e-12)
    for k in range(len(ts)):
        assert abs(Es[k] - float(z[f"test{k}_E"])) < E_TOL and np.abs(Fs[k] - z[f"test{k}_F"]).max() < F_TOL
        check_std(Es_std[k], Fs_std[k], z, k)
    # validate_data on the training set
    E, Ep, F, Fp = gp.validate_data()
    assert np.abs(E - z["val_E"]).max() < 1e-12 and np.abs(F - z["val_F"]).max() < 1e-12
    assert np.abs(Ep - z["val_E_pred"]).max() < E_TOL and np.abs(Fp - z["val_F_pred"]).max() < F_TOL


@pytest.mark.parametrize("name", [n for n in MODEL_CASES if n.endswith("pdpt")])
def test_predict_structure_stress_matches_reference(name):
    """S = K_trans1 alpha (gaussianprocess.py:890-893) with K_trans1 of the reference's k_total_with_stress and the
    reference's alpha; forces of the stress route equal the plain route"""
    z = load(name)
    gp = make_gp(z, stress=True)
    gp.fit(gp.points_from_structures(training_rows(z)), show=False, opt=False)
    for k in range(int(z["n_tests"])):
        t = atoms(z, f"test{k}")
        E, F, S = gp.predict_structure(t, stress=True, return_std=False, f_tol=1e-12)
        S_ref = (z[f"test{k}_stress_K_trans1"] @ z["alpha"]).reshape(len(t), 6)
        pred = (z[f"test{k}_stress_K_trans"] @ z["alpha"])[:, 0]
        assert np.abs(S - S_ref).max() < 1e-5 * max(1.0, np.abs(S_ref).max())
        assert abs(E - pred[0] * len(t)) < E_TOL and np.abs(F - pred[1:].reshape(-1, 3)).max() < F_TOL


@pytest.mark.parametrize("name", MODEL_CASES)
def test_log_marginal_likelihood_matches_reference(name):
    z = load(name)
    gp = make_gp(z)
    gp.set_train_pts(gp.points_from_structures(training_rows(z)))
    routes = (True, False) if str(z["kind"]) == "Dot" else (False,)
    for ws in routes:
        gp.weight_space_fit = ws
        for k in range(2):
            lml, g = gp.log_marginal_likelihood(z[f"lml{k}_theta"].copy(), eval_gradient=True)
            ref, gref = float(z[f"lml{k}"]), z[f"lml{k}_grad"]
            assert abs(lml - ref) < 1e-7 * max(1.0, abs(ref)), (ws, k)
            assert np.abs(g - gref).max() < 1e-6 * max(1.0, np.abs(gref).max()), (ws, k, g, gref)
            lml0 = gp.log_marginal_likelihood(z[f"lml{k}_theta"].copy())
            assert abs(lml0 - float(z[f"lml{k}_nograd"])) < 1e-7 * max(1.0, abs(ref))
        gp.noise_bounds = [1e-4, 1e-1]
        gp.f_coef = float(z["f_coef"])
        lml, g = gp.log_marginal_likelihood(z["lml_noise_theta"].copy(), eval_gradient=True)
        gp.noise_bounds = None
        assert len(g) == 3 and abs(lml - float(z["lml_noise"])) < 1e-7 * max(1.0, abs(float(z["lml_noise"])))
        assert np.abs(g - z["lml_noise_grad"]).max() < 1e-6 * max(1.0, np.abs(z["lml_noise_grad"]).max())


@pytest.mark.parametrize("name", MODEL_CASES)
def test_add_structure_queue_and_refit_match_reference(name):
    """selection policy of add_structure on a trained model (gaussianprocess.py:921-1002), queue counters, queued
    points invisible until the refit (:553-577), the refit itself, and -- where recorded -- fit(opt=True)"""
    z = load(name)
    gp = make_gp(z)
    gp.fit(gp.points_from_structures(training_rows(z)), show=False, opt=False)
    new = atoms(z, "add")
    pts, n_pts, errors = gp.add_structure((new, float(z["add_E"]), z["add_F"].copy()))
    assert list(pts["db"][0][4]) == z["add_force_in"].tolist() and n_pts == int(z["add_N_pts"])
    assert abs(errors[0] - float(z["add_err_E"])) < 1e-12 and abs(errors[1] - float(z["add_err_E_pred"])) < E_TOL
    dvar = explicit_inverse_noise(z["L"], z["test0_diag"])          # the recorded std carries the explicit-inverse noise
    assert abs(errors[2] ** 2 - float(z["add_err_E_std"]) ** 2) < max(dvar, 2 * E_TOL * float(z["add_err_E_std"]))
    assert np.abs(errors[4] - z["add_err_F_pred"]).max() < F_TOL
    assert np.abs(errors[5] ** 2 - z["add_err_F_std"] ** 2).max() < max(dvar, 2 * F_TOL * np.abs(z["add_err_F_std"]).max())
    assert [gp.N_energy, gp.N_forces, gp.N_energy_queue, gp.N_forces_queue, gp.N_queue] == z["add_counters"].tolist()
    t = atoms(z, "test0", fixed=z["test0_fixed"])
    E, F, _, E_std, F_std = gp.predict_structure(t, stress=False, return_std=True, f_tol=1e-12)
    assert abs(E - float(z["queued_E"])) < E_TOL and abs(E - float(z["test0_E"])) < E_TOL
    check_std(E_std, F_std, z, 0)                                    # queued points change nothing
    gp.fit(show=False, opt=False)
    assert np.abs(gp.y_train - z["refit_y_train"]).max() < 1e-12
    assert rel(gp.alpha_, z["refit_alpha"]) < 1e-5
    E, F, _, E_std, F_std = gp.predict_structure(t, stress=False, return_std=True, f_tol=1e-12)
    assert abs(E - float(z["refit_E"])) < E_TOL and np.abs(F - z["refit_F"]).max() < F_TOL
    check_std(E_std, F_std, z, 0, L_key="refit_L", Kt_key="refit_K_trans", tag="refit")
    if "opt_params" in z.files:
        gp.kernel.update([float(v) for v in z["para"]])
        gp.fit(show=False, opt=True, maxiter=int(z["opt_iters"]))
        assert np.abs(np.array(gp.kernel.parameters()) - z["opt_params"]).max() < 1e-4 * np.abs(z["opt_params"]).max()
        E, F, _, E_std, F_std = gp.predict_structure(t, stress=False, return_std=True, f_tol=1e-12)
        assert abs(E - float(z["opt_E"])) < 1e-4 and np.abs(F - z["opt_F"]).max() < 1e-4
