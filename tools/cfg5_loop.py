"""BASELINE.json configs[4] in shape, wall clock: the Pd4 / MgO(100) band of tests/golden/gpr_trace_pd4.npz (220 atoms,
GPR(freq=1): a refit with opt=True, maxiter=10 after every structure the base calculator labels) through this build's
calculator on the GPU, step by step, next to the seconds the reference's own calculator took for the same steps on one
host core of the build container (recorded in the fixture by tests/golden/make_golden_trace.py pd4).
usage: python tools/cfg5_loop.py"""
import contextlib, io, os, sys, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import numpy as np, torch
from gpr_calculator_amd.calculator import GPR
from gpr_calculator_amd.gaussianprocess import GP
from gpr_calculator_amd.SO3 import SO3
from gpr_calculator_amd.kernels.RBF_mb import RBF_mb
from gpr_calculator_amd.neighbors import SimpleAtoms
from gpr_calculator_amd.synthetic import lj_energy_forces

T = np.load(os.path.join(ROOT, "tests", "golden", "gpr_trace_pd4.npz"))
P = np.load(os.path.join(ROOT, "tests", "golden", "pd4_subset.npz"))
eps, sig, free = float(T["lj_eps"]), float(T["lj_sig"]), T["free"]


class Base:
    """the base calculator's stand-in (a soft LJ in numpy on the host).  Like an ASE calculator it keeps the result of the
    last geometry, so energy and forces of one structure cost one evaluation; its seconds are kept apart from the GPR side."""
    seconds, _key, _res = 0.0, None, None

    def _eval(self, atoms):
        key = atoms.positions.tobytes()
        if key != Base._key:
            t0 = time.perf_counter()
            Base._res, Base._key = lj_energy_forces(atoms, eps=eps, sig=sig), key
            Base.seconds += time.perf_counter() - t0
        return Base._res

    def get_potential_energy(self, atoms): return self._eval(atoms)[0]
    def get_forces(self, atoms): return self._eval(atoms)[1].copy()


class Atoms(SimpleAtoms):
    calc = None
    def get_potential_energy(self): return self.calc.get_potential_energy(self)
    def get_forces(self): return self.calc.get_forces(self)
    def copy(self): return Atoms(self.numbers.copy(), self.positions.copy(), self.cell.copy(), self.pbc.copy(), self.fixed.copy())


def make(image, seed, sigma):
    pos = P[f"positions_{image}"].copy()
    if sigma > 0.0:
        pos[free] += np.random.default_rng(seed).normal(0.0, sigma, (len(free), 3))
    fixed = [i for i in range(len(pos)) if i not in set(free.tolist())]
    return Atoms(P[f"numbers_{image}"], pos, P[f"cell_{image}"], P[f"pbc_{image}"], fixed=fixed)


def run():
    gp = GP(kernel=RBF_mb(para=[float(v) for v in T["para0"]], zeta=2.0), descriptor=SO3(nmax=3, lmax=4, rcut=5.0),
            noise_e=float(T["noise_e"]), noise_f=float(T["noise_f"]), log_file="/tmp/gpr_cfg5.log")
    for image in T["boot_images"].tolist():
        a = make(image, 0, 0.0)
        E, F = lj_energy_forces(a, eps=eps, sig=sig)
        F[a.fixed] = 0.0
        gp.add_structure((a, E, F))
    gp.fit(show=False, opt=False)
    calc = GPR(base=Base(), ff=gp, save=False, freq=int(T["freq"]))
    secs, base_secs = [], []
    for k in range(int(T["n_steps"])):
        image, seed, sigma = T["script"][k]
        atoms = make(int(image), int(seed), float(sigma))
        b0 = Base.seconds
        torch.cuda.synchronize(); t0 = time.perf_counter()
        with contextlib.redirect_stdout(io.StringIO()):
            calc.calculate(atoms)
        torch.cuda.synchronize(); secs.append(time.perf_counter() - t0)
        base_secs.append(Base.seconds - b0)
    run.base_secs = base_secs
    return gp, secs


run()                                        # first pass: library, tables, allocator warm
gp, secs = run()
base_secs = run.base_secs
ref = [float(T[f"ref_seconds{k}"]) for k in range(int(T["n_steps"]))] if "ref_seconds0" in T.files else None
print(f"cfg5-shaped loop: {len(secs)} calculator steps on 220-atom Pd4/MgO cells, base/surrogate/fits = {gp.use_base}/{gp.use_surrogate}/{gp.fits}")
print("  (\"GPR side\" = the step without the base calculator's own time: predict, add_structure, fit(opt=True, maxiter=10), validate)")
for k, s in enumerate(secs):
    kind = "base + refit" if int(T[f"refit{k}"]) else ("base" if int(T[f"used_base{k}"]) else "surrogate")
    print(f"  step {k:2d} {kind:13s} this build {s * 1e3:9.2f} ms = base calculator {base_secs[k] * 1e3:7.2f} + GPR side {(s - base_secs[k]) * 1e3:7.2f}"
          + (f"   reference (1 core) {ref[k]:8.2f} s   x{ref[k] / s:8.0f}" if ref else ""))
print(f"total: this build {sum(secs):.3f} s (base calculator {sum(base_secs):.3f} s, GPR side {sum(secs) - sum(base_secs):.3f} s)"
      + (f"; reference CPU path {sum(ref):.1f} s on {int(T['ref_cores'])} core -> x{sum(ref) / sum(secs):.0f}" if ref else ""))
