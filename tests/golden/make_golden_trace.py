"""A scripted on-the-fly run through the reference's OWN ASE calculator class (gpr_calc/calculator.py, class GPR,
imported unmodified through refshim.py) on top of its own GP: which calls the surrogate answers, which go to the
base calculator, what is queued, when the model refits -- recorded call by call.  Build container only.

    python tests/golden/make_golden_trace.py            14 steps on 4-atom Pd/Pt cells, refit every 3 queued points
    python tests/golden/make_golden_trace.py pd4        BASELINE.json configs[4] in shape: the Pd4 / MgO(100) images of the
                                                        reference's shipped database (tests/golden/pd4_subset.npz, 220
                                                        atoms), freq = 1 -- the model is refitted (opt=True, maxiter=10)
                                                        after EVERY structure the base calculator labels

Base calculator: the seeded Lennard-Jones labelling potential of gpr_calculator_amd/synthetic.py (EMT and VASP, the base
calculators of the reference's examples, need ASE / a DFT code; any deterministic potential exercises the same logic).
Output (data only): tests/golden/gpr_trace.npz, tests/golden/gpr_trace_pd4.npz
"""
import contextlib
import io
import os
import sys
import time
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import refshim                                                     # noqa: E402
from refshim import CalcAtoms                                      # noqa: E402
from gpr_calculator_amd.synthetic import fcc_cell, lj_energy_forces  # noqa: E402


EPS = 0.05          # a soft Lennard-Jones: forces of a few tenths of an eV/A, so that both branches of the decision occur


class LJBase:
    def get_potential_energy(self, atoms):
        return lj_energy_forces(atoms, eps=EPS)[0]

    def get_forces(self, atoms):
        return lj_energy_forces(atoms, eps=EPS)[1].copy()


def script():
    """(seed, displacement sigma, fixed atoms) of the structures an optimiser would ask for, in order"""
    return [(0, 0.05, ()), (1, 0.05, ()), (20, 0.05, ()), (21, 0.12, (0,)), (22, 0.05, ()), (23, 0.20, ()),
            (24, 0.05, (1, 2)), (25, 0.16, ()), (26, 0.05, ()), (27, 0.25, ()), (28, 0.08, ()), (29, 0.05, ()),
            (30, 0.22, (3,)), (31, 0.05, ())]


def make_atoms(seed, sigma, fixed):
    a = fcc_cell(seed=seed, reps=1, species=(46, 78), sigma=sigma)
    return CalcAtoms(a.numbers, a.positions, a.cell, a.pbc, fixed=fixed)


def main():
    warnings.simplefilter("ignore")
    ns = refshim.import_reference()
    des = ns.SO3.SO3(nmax=3, lmax=4, rcut=5.0, alpha=2.0, derivative=True, stress=False)
    kern = ns.RBF_mb.RBF_mb(para=[2.5, 1.3], zeta=2.0)
    gp = ns.gaussianprocess.GP(kernel=kern, descriptor=des, noise_e=0.002, noise_f=0.1, log_file="/tmp/ref_trace.log")
    for s in range(3):                                              # bootstrap as train_images does: add_structure + fit
        a = make_atoms(s, 0.05, ())
        E, F = lj_energy_forces(a, eps=EPS)
        gp.add_structure((a, E, F))
    gp.fit(show=False, opt=False)
    calc = ns.calculator.GPR(base=LJBase(), ff=gp, save=False, freq=3)
    p = dict(freq=3, lj_eps=EPS, para0=np.array([2.5, 1.3]), noise_e=0.002, noise_f=0.1, n_boot=3)
    steps = script()
    p["script"] = np.array([(s, sig, len(f)) for (s, sig, f) in steps], dtype=float)
    for k, (seed, sigma, fixed) in enumerate(steps):
        atoms = make_atoms(seed, sigma, fixed)
        p[f"fixed{k}"] = np.asarray(fixed, int)
        before = (gp.use_base, gp.use_surrogate, gp.fits)
        out = io.StringIO()
        with contextlib.redirect_stdout(out):
            calc.calculate(atoms)
        line = [l for l in out.getvalue().splitlines() if l.startswith("From ")][0]
        p[f"line{k}"] = line
        p[f"used_base{k}"] = gp.use_base - before[0]
        p[f"refit{k}"] = gp.fits - before[2]
        p[f"E{k}"], p[f"F{k}"] = calc.results["energy"], calc.results["forces"]
        p[f"var_e{k}"], p[f"var_f{k}"] = calc.results["var_e"], calc.results["var_f"]
        p[f"counters{k}"] = np.array([gp.N_energy, gp.N_forces, gp.N_energy_queue, gp.N_forces_queue, gp.N_queue])
        p[f"para{k}"] = np.array(kern.parameters())
        print(k, line, "| base" if p[f"used_base{k}"] else "| surrogate", "| refit" if p[f"refit{k}"] else "",
              p[f"counters{k}"].tolist(), np.round(p[f"para{k}"], 4))
    p["n_steps"] = len(steps)
    p["totals"] = np.array([gp.use_base, gp.use_surrogate, gp.fits])
    path = os.path.join(HERE, "gpr_trace.npz")
    np.savez_compressed(path, **p)
    print(path, os.path.getsize(path) // 1024, "KiB; base/surrogate/fits =", p["totals"].tolist())


# ---------------------------------------------------------------------------------------------- Pd4 / MgO, freq = 1
PD4_EPS, PD4_SIG = 0.3, 2.0         # Mg-O neighbours (2.1 A) sit near the minimum; displaced atoms feel forces of order 1 eV/A
PD4_FREE = (0, 108, 216, 217, 218, 219)     # what moves: the cluster and one Mg / O pair under it (cf. force_in of the db)


class LJBasePd4:
    def get_potential_energy(self, atoms):
        return lj_energy_forces(atoms, eps=PD4_EPS, sig=PD4_SIG)[0]

    def get_forces(self, atoms):
        return lj_energy_forces(atoms, eps=PD4_EPS, sig=PD4_SIG)[1].copy()


def pd4_script():
    """(image of the db subset, seed, displacement sigma of the free atoms): a band evaluated twice with growing noise"""
    return [(1, 11, 0.00), (2, 12, 0.00), (3, 13, 0.05), (4, 14, 0.03), (5, 15, 0.00), (1, 16, 0.10), (2, 17, 0.20),
            (3, 18, 0.02), (4, 19, 0.30), (5, 20, 0.05), (3, 21, 0.15), (2, 22, 0.01)]


def pd4_atoms(Z, image, seed, sigma):
    pos = Z[f"positions_{image}"].copy()
    free = np.array(PD4_FREE)
    if sigma > 0.0:
        pos[free] += np.random.default_rng(seed).normal(0.0, sigma, (len(free), 3))
    fixed = [i for i in range(len(pos)) if i not in set(PD4_FREE)]
    return CalcAtoms(Z[f"numbers_{image}"], pos, Z[f"cell_{image}"], Z[f"pbc_{image}"], fixed=fixed)


def main_pd4():
    warnings.simplefilter("ignore")
    Z = np.load(os.path.join(HERE, "pd4_subset.npz"))
    ns = refshim.import_reference()
    des = ns.SO3.SO3(nmax=3, lmax=4, rcut=5.0, alpha=2.0, derivative=True, stress=False)
    para0 = [5.0, 1.5]
    kern = ns.RBF_mb.RBF_mb(para=list(para0), zeta=2.0)
    noise_e, noise_f = 0.05 / 220, 0.05            # examples/Pd4/gp_neb.py: noise_e / len(image), noise_f
    gp = ns.gaussianprocess.GP(kernel=kern, descriptor=des, noise_e=noise_e, noise_f=noise_f, log_file="/tmp/ref_trace_pd4.log")
    for image in (0, 6):                            # the two end points, as GP.set_GPR labels them
        a = pd4_atoms(Z, image, 0, 0.0)
        E, F = lj_energy_forces(a, eps=PD4_EPS, sig=PD4_SIG)
        F[a.fixed] = 0.0
        gp.add_structure((a, E, F))
    gp.fit(show=False, opt=False)
    calc = ns.calculator.GPR(base=LJBasePd4(), ff=gp, save=False, freq=1)
    p = dict(freq=1, lj_eps=PD4_EPS, lj_sig=PD4_SIG, para0=np.array(para0), noise_e=noise_e, noise_f=noise_f,
             boot_images=np.array([0, 6]), free=np.array(PD4_FREE))
    steps = pd4_script()
    p["script"] = np.array(steps, dtype=float)
    for k, (image, seed, sigma) in enumerate(steps):
        atoms = pd4_atoms(Z, image, seed, sigma)
        before = (gp.use_base, gp.use_surrogate, gp.fits)
        out = io.StringIO()
        t_step = time.perf_counter()
        with contextlib.redirect_stdout(out):
            calc.calculate(atoms)
        p[f"ref_seconds{k}"] = time.perf_counter() - t_step      # the reference's CPU path, one process, this container
        line = [l for l in out.getvalue().splitlines() if l.startswith("From ")][0]
        p[f"line{k}"] = line
        p[f"used_base{k}"] = gp.use_base - before[0]
        p[f"refit{k}"] = gp.fits - before[2]
        p[f"E{k}"], p[f"F{k}"] = calc.results["energy"], calc.results["forces"]
        p[f"var_e{k}"], p[f"var_f{k}"] = calc.results["var_e"], calc.results["var_f"]
        p[f"counters{k}"] = np.array([gp.N_energy, gp.N_forces, gp.N_energy_queue, gp.N_forces_queue, gp.N_queue])
        p[f"para{k}"] = np.array(kern.parameters())
        print(k, line, "| base" if p[f"used_base{k}"] else "| surrogate", "| refit" if p[f"refit{k}"] else "",
              p[f"counters{k}"].tolist(), np.round(p[f"para{k}"], 4), flush=True)
    p["n_steps"] = len(steps)
    p["totals"] = np.array([gp.use_base, gp.use_surrogate, gp.fits])
    p["ref_cores"] = 1
    path = os.path.join(HERE, "gpr_trace_pd4.npz")
    np.savez_compressed(path, **p)
    print(path, os.path.getsize(path) // 1024, "KiB; base/surrogate/fits =", p["totals"].tolist())


if __name__ == "__main__":
    main_pd4() if sys.argv[1:] == ["pd4"] else main()
