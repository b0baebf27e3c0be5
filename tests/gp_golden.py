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
