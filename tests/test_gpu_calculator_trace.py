"""GPU: the ASE-calculator layer (GPR.calculate: surrogate-or-base decision, queueing, refit trigger) replayed against
a call-by-call trace of the reference's OWN calculator class on the same scripted run (tests/golden/gpr_trace.npz,
produced by tests/golden/make_golden_trace.py from gpr_calc/calculator.py:48-155 + gaussianprocess.py)."""
import contextlib
import io
import os
import re

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
Z = np.load(os.path.join(os.path.dirname(__file__), "golden", "gpr_trace.npz"))


def numbers_of(line):
    return [float(x) for x in re.findall(r"-?\d+\.\d+", line)]


def test_calculator_replays_the_reference_trace():
    from gpr_calculator_amd.calculator import GPR
    from gpr_calculator_amd.gaussianprocess import GP
    from gpr_calculator_amd.SO3 import SO3
    from gpr_calculator_amd.kernels.RBF_mb import RBF_mb
    from gpr_calculator_amd.neighbors import SimpleAtoms
    from gpr_calculator_amd.synthetic import fcc_cell, lj_energy_forces
    eps = float(Z["lj_eps"])

    class Base:
        calls = 0

        def get_potential_energy(self, atoms):
            Base.calls += 1
            return lj_energy_forces(atoms, eps=eps)[0]

        def get_forces(self, atoms):
            return lj_energy_forces(atoms, eps=eps)[1].copy()

    class Atoms(SimpleAtoms):
        calc = None

        def get_potential_energy(self):
            return self.calc.get_potential_energy(self)

        def get_forces(self):
            return self.calc.get_forces(self)

        def copy(self):
            return Atoms(self.numbers.copy(), self.positions.copy(), self.cell.copy(), self.pbc.copy(), self.fixed.copy())

    def make(seed, sigma, fixed):
        a = fcc_cell(seed=seed, reps=1, species=(46, 78), sigma=sigma)
        return Atoms(a.numbers, a.positions, a.cell, a.pbc, fixed=fixed)

    gp = GP(kernel=RBF_mb(para=[float(v) for v in Z["para0"]], zeta=2.0), descriptor=SO3(nmax=3, lmax=4, rcut=5.0),
            noise_e=float(Z["noise_e"]), noise_f=float(Z["noise_f"]), log_file="/tmp/gpr_test.log")
    for s in range(int(Z["n_boot"])):
        a = make(s, 0.05, ())
        E, F = lj_energy_forces(a, eps=eps)
        gp.add_structure((a, E, F))
    gp.fit(show=False, opt=False)
    calc = GPR(base=Base(), ff=gp, save=False, freq=int(Z["freq"]))
    refits = 0
    for k in range(int(Z["n_steps"])):
        seed, sigma, _ = Z["script"][k]
        atoms = make(int(seed), float(sigma), Z[f"fixed{k}"].tolist())
        before = (gp.use_base, gp.fits)
        out = io.StringIO()
        with contextlib.redirect_stdout(out):
            calc.calculate(atoms)
        line = [l for l in out.getvalue().splitlines() if l.startswith("From ")][0]
        ref_line = str(Z[f"line{k}"])
        # same branch, same counters, same refit points
        assert gp.use_base - before[0] == int(Z[f"used_base{k}"]), (k, line, ref_line)
        assert gp.fits - before[1] == int(Z[f"refit{k}"]), k
        assert [gp.N_energy, gp.N_forces, gp.N_energy_queue, gp.N_forces_queue, gp.N_queue] == Z[f"counters{k}"].tolist(), k
        # numbers: north-star tolerances while the hyper-parameters are the scripted ones; after a refit with
        # opt=True both sides have run L-BFGS-B (ftol 1e-2) on their own objective evaluations, compared at 1e-3
        tol_e, tol_f = (1e-6, 1e-5) if refits == 0 else (2e-3, 2e-3)
        assert abs(calc.results["energy"] - float(Z[f"E{k}"])) < tol_e, (k, calc.results["energy"], float(Z[f"E{k}"]))
        assert np.abs(calc.results["forces"] - Z[f"F{k}"]).max() < tol_f
        assert abs(calc.results["var_e"] - float(Z[f"var_e{k}"])) < max(tol_e, 1e-5)
        assert np.abs(calc.results["var_f"] - Z[f"var_f{k}"]).max() < max(tol_f, 1e-4)
        # the console line the reference's analysis scripts parse: same prefix, same fields to the printed precision
        assert line.split("E:")[0] == ref_line.split("E:")[0]
        got, want = numbers_of(line), numbers_of(ref_line)
        assert len(got) == len(want) == 6 and np.abs(np.array(got) - np.array(want)).max() <= (0.0011 if refits == 0 else 0.01), (line, ref_line)
        refits += int(Z[f"refit{k}"])
        if int(Z[f"refit{k}"]):
            assert np.abs(np.array(gp.kernel.parameters()) - Z[f"para{k}"]).max() < 2e-3 * np.abs(Z[f"para{k}"]).max()
    assert [gp.use_base, gp.use_surrogate, gp.fits] == Z["totals"].tolist() and Base.calls == gp.use_base


def test_pd4_band_with_a_refit_after_every_labelled_structure():
    """BASELINE.json configs[4] in shape: the Pd4 / MgO(100) images of the reference's shipped database (220 atoms, six of
    them moving), GPR(freq=1) -- every structure the base calculator labels triggers fit(opt=True, maxiter=10) -- replayed
    against the call-by-call trace of the reference's own calculator on the same script
    (tests/golden/gpr_trace_pd4.npz from make_golden_trace.py pd4).  Branch taken, queue counters and refit points must
    be the reference's at every step; numbers at the north-star tolerance until the first optimised refit, then at the
    resolution L-BFGS-B with ftol = 1e-2 leaves to two implementations of the same objective."""
    from gpr_calculator_amd.calculator import GPR
    from gpr_calculator_amd.gaussianprocess import GP
    from gpr_calculator_amd.SO3 import SO3
    from gpr_calculator_amd.kernels.RBF_mb import RBF_mb
    from gpr_calculator_amd.neighbors import SimpleAtoms
    from gpr_calculator_amd.synthetic import lj_energy_forces
    T = np.load(os.path.join(os.path.dirname(__file__), "golden", "gpr_trace_pd4.npz"))
    P = np.load(os.path.join(os.path.dirname(__file__), "golden", "pd4_subset.npz"))
    eps, sig, free = float(T["lj_eps"]), float(T["lj_sig"]), T["free"]

    class Base:
        calls = 0

        def get_potential_energy(self, atoms):
            Base.calls += 1
            return lj_energy_forces(atoms, eps=eps, sig=sig)[0]

        def get_forces(self, atoms):
            return lj_energy_forces(atoms, eps=eps, sig=sig)[1].copy()

    class Atoms(SimpleAtoms):
        calc = None

        def get_potential_energy(self):
            return self.calc.get_potential_energy(self)

        def get_forces(self):
            return self.calc.get_forces(self)

        def copy(self):
            return Atoms(self.numbers.copy(), self.positions.copy(), self.cell.copy(), self.pbc.copy(), self.fixed.copy())

    def make(image, seed, sigma):
        pos = P[f"positions_{image}"].copy()
        if sigma > 0.0:
            pos[free] += np.random.default_rng(seed).normal(0.0, sigma, (len(free), 3))
        fixed = [i for i in range(len(pos)) if i not in set(free.tolist())]
        return Atoms(P[f"numbers_{image}"], pos, P[f"cell_{image}"], P[f"pbc_{image}"], fixed=fixed)

    gp = GP(kernel=RBF_mb(para=[float(v) for v in T["para0"]], zeta=2.0), descriptor=SO3(nmax=3, lmax=4, rcut=5.0),
            noise_e=float(T["noise_e"]), noise_f=float(T["noise_f"]), log_file="/tmp/gpr_test.log")
    for image in T["boot_images"].tolist():
        a = make(image, 0, 0.0)
        E, F = lj_energy_forces(a, eps=eps, sig=sig)
        F[a.fixed] = 0.0
        gp.add_structure((a, E, F))
    gp.fit(show=False, opt=False)
    calc = GPR(base=Base(), ff=gp, save=False, freq=int(T["freq"]))
    refits, report = 0, []
    for k in range(int(T["n_steps"])):
        image, seed, sigma = T["script"][k]
        atoms = make(int(image), int(seed), float(sigma))
        before = (gp.use_base, gp.fits)
        out = io.StringIO()
        with contextlib.redirect_stdout(out):
            calc.calculate(atoms)
        line = [l for l in out.getvalue().splitlines() if l.startswith("From ")][0]
        ref_line = str(T[f"line{k}"])
        report.append((k, line, ref_line, np.round(gp.kernel.parameters(), 4), np.round(T[f"para{k}"], 4)))
        assert gp.use_base - before[0] == int(T[f"used_base{k}"]), report
        assert gp.fits - before[1] == int(T[f"refit{k}"]), report
        assert [gp.N_energy, gp.N_forces, gp.N_energy_queue, gp.N_forces_queue, gp.N_queue] == T[f"counters{k}"].tolist(), report
        if int(T[f"used_base{k}"]):
            # labelled by the base calculator: E and F ARE the base potential's, on both sides
            assert abs(calc.results["energy"] - float(T[f"E{k}"])) < 1e-9 * abs(float(T[f"E{k}"])), report
            assert np.abs(calc.results["forces"] - T[f"F{k}"]).max() < 1e-9 * max(1.0, np.abs(T[f"F{k}"]).max())
        else:
            scale = max(1.0, np.abs(T[f"F{k}"]).max())
            tol_e, tol_f = (1e-6, 1e-5) if refits == 0 else (5e-3 * scale, 5e-3 * scale)
            assert abs(calc.results["energy"] - float(T[f"E{k}"])) < tol_e * (1 if refits == 0 else 220 ** 0.5), report
            assert np.abs(calc.results["forces"] - T[f"F{k}"]).max() < tol_f, report
        assert line.split("E:")[0] == ref_line.split("E:")[0]
        refits += int(T[f"refit{k}"])
    assert [gp.use_base, gp.use_surrogate, gp.fits] == T["totals"].tolist() and Base.calls == gp.use_base
    assert refits >= 6 and gp.use_surrogate >= 3          # the script does exercise both branches and retrain-every-add
