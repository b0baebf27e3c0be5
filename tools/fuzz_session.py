"""Randomised cross-check of the prediction routes (Dot, zeta = 2): the session call (neighbour search, descriptor,
feature rows and weight-space tail on the device behind one C call) against the kernel-object route of the same model
(pair kernel K_trans, dense Cholesky factor, triangular products) -- two independent implementations of the same
numbers.  Models with one to three species, structures of 4 to 64 atoms (odd counts, strained cells, fixed atoms), batches
of 1 to 100 mixed structures: every tail form, both force-feature forms, one to three feature blocks.
usage: python tools/fuzz_session.py [cases=40] [seed=0]"""
import sys, os, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np
from gpr_calculator_amd.gaussianprocess import GP
from gpr_calculator_amd.SO3 import SO3
from gpr_calculator_amd.kernels.Dot_mb import Dot_mb
from gpr_calculator_amd.neighbors import SimpleAtoms
from gpr_calculator_amd.synthetic import fcc_cell, lj_energy_forces

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(seed0)


def random_structure(species, seed):
    r = np.random.default_rng(seed)
    reps = int(r.choice([1, 2, 2, 2]))
    at = fcc_cell(seed=seed, a=float(r.uniform(3.7, 4.1)), reps=reps, sigma=float(r.uniform(0.03, 0.12)), species=species)
    keep = np.ones(len(at), bool)
    if len(at) > 8 and r.random() < 0.5:                      # odd atom counts
        keep[r.choice(len(at), size=int(r.integers(1, 6)), replace=False)] = False
    cell = np.array(at.cell) @ (np.eye(3) + r.uniform(-0.04, 0.04, (3, 3)))      # strained, slightly triclinic
    pos = np.array(at.positions)[keep] @ np.linalg.solve(np.array(at.cell), cell)
    fixed = np.sort(r.choice(keep.sum(), size=int(r.integers(0, 3)), replace=False)) if r.random() < 0.3 else np.array([], int)
    return SimpleAtoms(np.asarray(at.numbers)[keep], pos, cell, True, fixed=fixed)


worst = {"E": 0.0, "F": 0.0, "Es": 0.0, "Fs": 0.0}
t0 = time.time()
for c in range(cases):
    species = [(46,), (46, 8), (12, 8, 46)][c % 3]
    gp = GP(kernel=Dot_mb(para=[float(rng.uniform(1.0, 3.0)), float(rng.uniform(0.5, 2.5))], zeta=2.0),
            descriptor=SO3(nmax=3, lmax=4, rcut=5.0, alpha=2.0), noise_e=0.002, noise_f=0.1,
            log_file=os.path.join("/tmp", f"gpr_fuzz_{os.getpid()}.log"))
    rows = []
    for s in range(int(rng.integers(4, 9))):
        at = random_structure(species, 1000 * c + s)
        E, F = lj_energy_forces(at)
        rows.append((at, E, F, True, list(rng.choice(len(at), size=min(len(at), int(rng.integers(2, 9))), replace=False))))
    gp.set_train_pts(gp.points_from_structures(rows))
    gp.fit(show=False, opt=False)
    nb = int(rng.choice([1, 1, 2, 5, 9, 17, 40, 70, 100]))
    batch = [random_structure(species, 1000 * c + 100 + k) for k in range(nb)]
    a = gp.predict_structures(batch, return_std=True)
    gp.session_batch = False; gp.native_batch = False; gp.weight_space_mean = False
    b = gp.predict_structures(batch, return_std=True)
    scale_f = max(1.0, max(np.abs(np.asarray(f)).max() for f in b[1]))
    for k in range(nb):
        n = len(batch[k])
        worst["E"] = max(worst["E"], abs(a[0][k] - b[0][k]) / max(1.0, abs(b[0][k])))
        worst["F"] = max(worst["F"], np.abs(np.asarray(a[1][k]) - np.asarray(b[1][k])).max() / scale_f)
        worst["Es"] = max(worst["Es"], abs(a[2][k] - b[2][k]))
        worst["Fs"] = max(worst["Fs"], np.abs(np.asarray(a[3][k]) - np.asarray(b[3][k])).max())
    print(f"case {c:3d}: species {species}, {len(rows)} training structures, batch {nb:2d} ({sum(len(s) for s in batch)} atoms): "
          f"worst so far E {worst['E']:.1e} F {worst['F']:.1e} E_std {worst['Es']:.1e} F_std {worst['Fs']:.1e}", flush=True)
ok = worst["E"] < 1e-7 and worst["F"] < 1e-7 and worst["Es"] < 1e-5 and worst["Fs"] < 1e-5
print(f"{cases} cases in {time.time() - t0:.0f} s: {'ALL OK' if ok else 'FAILED'}")
sys.exit(0 if ok else 1)
