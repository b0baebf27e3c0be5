"""Helpers shared by the CPU and GPU tests that read tests/golden/gp_*.npz (reference-produced GP-level
fixtures, see tests/golden/make_golden_gp.py)."""
import glob
import os
import numpy as np

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
MODEL_CASES = sorted(os.path.basename(p)[3:-4] for p in glob.glob(os.path.join(GOLDEN, "gp_*.npz"))
                     if not os.path.basename(p).startswith(("gp_wrappers", "gp_cfg")))


def load(name):
    return np.load(os.path.join(GOLDEN, f"gp_{name}.npz"))


def energy_tuple(z, key):
    return (z[key + "_X"], z[key + "_ELE"], z[key + "_counts"].tolist())


def force_tuple(z, key):
    return (z[key + "_X"], z[key + "_dXdR"], z[key + "_ELE"], z[key + "_counts"].tolist())


def train_x(z):
    return {"energy": energy_tuple(z, "trainx_energy"), "force": force_tuple(z, "trainx_force")}


def predicted_data(z, k, stress=False):
    key = f"test{k}_stress_data" if stress else f"test{k}_data"
    return {"energy": energy_tuple(z, key + "_energy"), "force": force_tuple(z, key + "_force")}


def rel(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-300))


def atoms(z, key, fixed=()):
    from gpr_calculator_amd.neighbors import SimpleAtoms
    return SimpleAtoms(z[key + "_numbers"], z[key + "_positions"], z[key + "_cell"], z[key + "_pbc"], fixed=fixed)


def training_rows(z):
    """(atoms, E, F, energy_in, force_ids) rows of the case's training set"""
    return [(atoms(z, f"train{s}"), float(z[f"train{s}_E"]), z[f"train{s}_F"], True, z[f"train{s}_force_ids"].tolist())
            for s in range(int(z["n_train"]))]


def stable_std(L, K_trans, diag):
    """sqrt(max(0, diag - |L^-1 k|^2)): the reference's predictive variance (gaussianprocess.py:903-907) evaluated
    from the reference's OWN L, K_trans and diag through a triangular solve instead of its explicit K^-1"""
    from scipy.linalg import solve_triangular
    v = solve_triangular(np.asarray(L), np.asarray(K_trans).T, lower=True)
    return np.sqrt(np.clip(np.asarray(diag) - (v * v).sum(axis=0), 0.0, None))


def explicit_inverse_noise(L, diag):
    """rounding noise of var = diag - k K^-1 k through an explicit inverse: ~ cond(K + noise) * eps * diag.  The
    reference's recorded std carries it (checked against long-double arithmetic when the fixtures were made: its
    E_std of the one-species 4-atom cases is off by 2e-5 while the triangular form agrees to 1e-11)."""
    return 20 * 1.1e-16 * np.linalg.cond(np.asarray(L)) ** 2 * float(np.abs(diag).max())


def split_std(std_flat, n, fixed):
    free = [i for i in range(n) if i not in set(int(f) for f in fixed)]
    F_std = np.zeros((n, 3))
    F_std[free] = std_flat[1:].reshape(-1, 3)
    return std_flat[0], F_std
