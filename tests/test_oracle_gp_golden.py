"""CPU: oracle/gp_numpy.py (the numpy/scipy restatement of the reference's Python above the natives) against
fixtures the reference's own Python produced (tests/golden/make_golden_gp.py): wrapper normalisations, k_total /
k_total_with_grad / k_total_with_stress / diag, fit, predict, log-marginal likelihood + gradient.  1e-12."""
import numpy as np
import pytest

from gp_golden import MODEL_CASES, load, train_x, predicted_data, rel, energy_tuple, force_tuple
from oracle import gp_numpy as og

TOL = 1e-12


def oracle_kernel(z, natives):
    cls = og.OracleRBF if str(z["kind"]) == "RBF" else og.OracleDot
    return cls(para=tuple(z["para"]), zeta=float(z["zeta"]), natives=natives)


def test_fixtures_present():
    assert len(MODEL_CASES) >= 7


def test_wrappers_match_reference(port_natives):
    z = load("wrappers")
    W = og.Wrappers(port_natives)
    c1, c2 = z["counts1"].tolist(), z["counts2"].tolist()
    XE1, XE2 = (z["x1"], z["ele1"], c1), (z["x2"], z["ele2"], c2)
    XF1, XF2 = (z["x1"], z["dx1dr"], z["ele1"], c1), (z["x2"], z["dx2dr"], z["ele2"], c2)
    XF1s = (z["x1"], z["dx1dr9"], z["ele1"], c1)
    s, l, s0 = float(z["sigma"]), float(z["l"]), float(z["sigma0"])
    for zeta in (2.0, 3.0):
        t = f"z{int(zeta)}"
        got = {
            f"rbf_kee_{t}": W.rbf_kee(XE1, XE2, s, l, zeta),
            f"rbf_kee_grad_{t}": np.stack(W.rbf_kee(XE1, XE2, s, l, zeta, grad=True)),
            f"rbf_kef_{t}": W.rbf_kef(XE1, XF2, s, l, zeta),
            f"rbf_kef_grad_{t}": np.stack(W.rbf_kef(XE1, XF2, s, l, zeta, grad=True)),
            f"rbf_kef_T_{t}": W.rbf_kef(XE1, XF2, s, l, zeta, transpose=True),
            f"rbf_kff_{t}": W.rbf_kff(XF1, XF2, s, l, zeta, tol=1e-12),
            f"rbf_kff_grad_{t}": np.stack(W.rbf_kff(XF1, XF2, s, l, zeta, grad=True)),
            f"dot_kee_{t}": W.dot_kee(XE1, XE2, s, s0, zeta),
            f"dot_kee_grad_{t}": np.stack(W.dot_kee(XE1, XE2, s, s0, zeta, grad=True)),
            f"dot_kef_{t}": W.dot_kef(XE1, XF2, s, s0, zeta),
            f"dot_kef_grad_{t}": np.stack(W.dot_kef(XE1, XF2, s, s0, zeta, grad=True)),
            f"dot_kef_T_{t}": W.dot_kef(XE1, XF2, s, s0, zeta, transpose=True),
            f"dot_kff_{t}": W.dot_kff(XF1, XF2, s, s0, zeta),
            f"dot_kff_grad_{t}": np.stack(W.dot_kff(XF1, XF2, s, s0, zeta, grad=True)),
        }
        a, b = W.rbf_kef(XE2, XF1s, s, l, zeta, stress=True, transpose=True)
        got[f"rbf_kef_stress_T_{t}"], got[f"rbf_kef_stress_T1_{t}"] = a, b
        a, b = W.rbf_kff(XF1s, XF2, s, l, zeta, stress=True, tol=1e-12)
        got[f"rbf_kff_stress_{t}"], got[f"rbf_kff_stress1_{t}"] = a, b
        a, b = W.dot_kef(XE2, XF1s, s, s0, zeta, stress=True, transpose=True)
        got[f"dot_kef_stress_T_{t}"], got[f"dot_kef_stress_T1_{t}"] = a, b
        a, b = W.dot_kff(XF1s, XF2, s, s0, zeta, stress=True)
        got[f"dot_kff_stress_{t}"], got[f"dot_kff_stress1_{t}"] = a, b
        for k, v in got.items():
            assert v.shape == z[k].shape, k
            assert rel(v, z[k]) < TOL, (k, rel(v, z[k]))


@pytest.mark.parametrize("name", MODEL_CASES)
def test_kernel_objects_match_reference(name, port_natives):
    z = load(name)
    kern, tx = oracle_kernel(z, port_natives), train_x(z)
    assert rel(kern.k_total(tx), z["K"]) < TOL
    K, dK = kern.k_total_with_grad(tx)
    assert rel(K, z["K_with_grad"]) < TOL
    for j in range(2):
        assert np.abs(dK[:, :, j] - z["dK"][:, :, j]).max() <= TOL * max(1.0, np.abs(z["dK"][:, :, j]).max())
    assert rel(kern.diag({"energy": tx["energy"], "force": og.tuple_to_list(tx["force"])}), z["diag_train"]) < TOL
    for k in range(int(z["n_tests"])):
        data = predicted_data(z, k)
        assert rel(kern.k_total(data, tx, 1e-12), z[f"test{k}_K_trans"]) < TOL
        assert rel(kern.diag({"energy": data["energy"], "force": og.tuple_to_list(data["force"])}), z[f"test{k}_diag"]) < TOL
        if f"test{k}_stress_K_trans" in z.files:
            sdata = predicted_data(z, k, stress=True)
            sdata["force"] = og.tuple_to_list(sdata["force"])
            Kt, Kt1 = kern.k_total_with_stress(sdata, tx, 1e-12)
            assert rel(Kt, z[f"test{k}_stress_K_trans"]) < TOL and rel(Kt1, z[f"test{k}_stress_K_trans1"]) < TOL


@pytest.mark.parametrize("name", MODEL_CASES)
def test_fit_predict_lml_match_reference(name, port_natives):
    z = load(name)
    kern, tx = oracle_kernel(z, port_natives), train_x(z)
    NE, ne, nf, y = len(tx["energy"][-1]), float(z["noise_e"]), float(z["noise_f"]), z["y_train"]
    L, alpha, K_inv = og.fit_core(kern.k_total(tx), y, NE, ne, nf)
    # same scipy calls as the reference on K that agrees to 1e-12: the factor agrees to rounding x cond
    assert rel(L, z["L"]) < 1e-11 and rel(alpha, z["alpha"]) < 1e-7 and rel(K_inv, z["K_inv"]) < 1e-7
    for k in range(int(z["n_tests"])):
        data = predicted_data(z, k)
        diag = kern.diag({"energy": data["energy"], "force": og.tuple_to_list(data["force"])})
        mean, std = og.predict_core(kern.k_total(data, tx, 1e-12), alpha, K_inv, diag)
        n = len(z[f"test{k}_numbers"])
        free = [i for i in range(n) if i not in set(z[f"test{k}_fixed"].tolist())]
        assert abs(mean[0] * n - float(z[f"test{k}_E"])) < 1e-9
        assert np.abs(mean[1:].reshape(-1, 3) - z[f"test{k}_F"][free]).max() < 1e-9
        # var = diag - k K^-1 k through the explicit inverse cancels ~cond(K) * eps: two evaluations of the
        # reference's own formula on inputs equal to 1e-12 differ by up to ~1e-7 in std (Dot, 32-atom cells)
        assert abs(std[0] - float(z[f"test{k}_E_std"])) < 1e-6
        assert np.abs(std[1:].reshape(-1, 3) - z[f"test{k}_F_std"][free]).max() < 1e-6
    for k in range(2):
        lml, g = og.log_marginal_likelihood(kern, tx, y, NE, ne, nf, z[f"lml{k}_theta"])
        assert abs(lml - float(z[f"lml{k}"])) < 1e-9 * max(1.0, abs(float(z[f"lml{k}"])))
        assert np.abs(g - z[f"lml{k}_grad"]).max() < 1e-7 * max(1.0, np.abs(z[f"lml{k}_grad"]).max())
        lml0 = og.log_marginal_likelihood(kern, tx, y, NE, ne, nf, z[f"lml{k}_theta"], eval_gradient=False)
        assert abs(lml0 - float(z[f"lml{k}_nograd"])) < 1e-9 * max(1.0, abs(lml0))
    lml, g = og.log_marginal_likelihood(kern, tx, y, NE, ne, nf, z["lml_noise_theta"], noise_param=True,
                                        f_coef=float(z["f_coef"]))
    assert len(g) == 3 and abs(lml - float(z["lml_noise"])) < 1e-9 * max(1.0, abs(lml))
    assert np.abs(g - z["lml_noise_grad"]).max() < 1e-7 * max(1.0, np.abs(z["lml_noise_grad"]).max())
