"""Seeded synthetic inputs of the benchmark and the parity tests:
32-atom fcc Pd cells with Gaussian displacements, a smooth Lennard-Jones labelling potential, and
NEB-like image batches.  Data generation only -- nothing here is part of the model."""
import numpy as np

from .neighbors import SimpleAtoms, neighbor_pairs


def fcc_cell(seed=0, a=3.89, reps=2, sigma=0.10, species=(46,)):
    """2x2x2 conventional fcc cells = 32 atoms, fully periodic, displaced by N(0, sigma)"""
    rng = np.random.default_rng(seed)
    base = np.array([[0, 0, 0], [0.5, 0.5, 0], [0.5, 0, 0.5], [0, 0.5, 0.5]])
    pos = np.array([(b + [i, j, k]) * a for i in range(reps) for j in range(reps) for k in range(reps) for b in base])
    pos = pos + rng.normal(0.0, sigma, pos.shape)
    numbers = rng.choice(np.asarray(species), size=len(pos)) if len(species) > 1 else np.full(len(pos), species[0])
    return SimpleAtoms(numbers, pos, np.eye(3) * a * reps, True)


def lj_energy_forces(atoms, eps=0.5, sig=2.5, rc=5.0):
    """truncated-and-shifted Lennard-Jones labels (E, F[n,3]) -- consistent derivatives for the GP"""
    ci, cj, vec = neighbor_pairs(atoms, rc)
    r = np.linalg.norm(vec, axis=1)
    sr6 = (sig / r) ** 6
    e_rc = 4 * eps * ((sig / rc) ** 12 - (sig / rc) ** 6)
    e_pair = 4 * eps * (sr6 * sr6 - sr6) - e_rc
    dedr = 4 * eps * (-12 * sr6 * sr6 + 6 * sr6) / r
    F = np.zeros((len(atoms), 3))
    np.add.at(F, ci, (dedr / r)[:, None] * vec)          # force on centre i from neighbour j
    return 0.5 * e_pair.sum(), F


def neb_images(n_images=64, seed=100, jitter=0.02):
    """linear interpolation between two displaced 32-atom cells plus a small jitter"""
    a, b = fcc_cell(seed), fcc_cell(seed + 1)
    rng = np.random.default_rng(seed + 2)
    out = []
    for k in range(n_images):
        t = k / max(n_images - 1, 1)
        pos = (1 - t) * a.positions + t * b.positions + rng.normal(0.0, jitter, a.positions.shape)
        out.append(SimpleAtoms(a.numbers, pos, a.cell, a.pbc))
    return out
