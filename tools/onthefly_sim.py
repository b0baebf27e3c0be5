"""cfg5-shaped plumbing run without ASE: an on-the-fly loop through the GPR calculator (calculator.py) on synthetic
32-atom cells with Lennard-Jones labels standing in for the ab-initio base.  A band of `nimg` images is relaxed by
steepest descent on the calculator's forces for `nsteps` steps; the calculator decides surrogate vs base per call,
queues points and refits with fit(opt=True, maxiter=10) exactly like the reference's loop.  Prints the base /
surrogate / fit counters and the wall time split.  usage: python tools/onthefly_sim.py [Dot|RBF] [nsteps] [nimg]"""
import sys, os, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np, torch
from gpr_calculator_amd.calculator import GPR
from gpr_calculator_amd.gaussianprocess import GP
from gpr_calculator_amd.SO3 import SO3
from gpr_calculator_amd.kernels.RBF_mb import RBF_mb
from gpr_calculator_amd.kernels.Dot_mb import Dot_mb
from gpr_calculator_amd.synthetic import fcc_cell, lj_energy_forces, neb_images

kind = sys.argv[1] if len(sys.argv) > 1 else "Dot"
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
nimg = int(sys.argv[3]) if len(sys.argv) > 3 else 5


class Base:
    name, calls, seconds = "lj", 0, 0.0


base = Base()
Cell = type(fcc_cell(0))


class Atoms(Cell):
    def get_potential_energy(self):
        t0 = time.perf_counter(); base.calls += 1
        e = lj_energy_forces(self)[0]
        base.seconds += time.perf_counter() - t0
        return e

    def get_forces(self):
        return lj_energy_forces(self)[1]

    def copy(self):
        return Atoms(self.numbers.copy(), self.positions.copy(), self.cell.copy(), self.pbc.copy(), self.fixed.copy())


def wrap(s):
    return Atoms(s.numbers, s.positions.copy(), s.cell, s.pbc)


kern = Dot_mb(para=[2.0, 2.0], zeta=2.0) if kind == "Dot" else RBF_mb(para=[1.0, 0.5], zeta=2.0)
gp = GP(kernel=kern, descriptor=SO3(nmax=3, lmax=4, rcut=5.0), noise_e=0.002, noise_f=0.1, log_file="/tmp/gpr_otf.log")
images = [wrap(s) for s in neb_images(nimg, seed=7)]
for a in (images[0], images[-1]):                      # the two end points are labelled up front (set_GPR does this)
    E, F = lj_energy_forces(a)
    gp.add_structure((a.copy(), E, F))
t_fit0 = time.perf_counter()
gp.fit(show=False, opt=True, maxiter=10)
torch.cuda.synchronize()
print(f"initial fit ({gp.N_energy} energies, {gp.N_forces} force points): {(time.perf_counter() - t_fit0) * 1e3:.1f} ms")
calc = GPR(base=base, ff=gp, save=False, freq=10, return_std=True)
calc.verbose = False
import gpr_calculator_amd.gaussianprocess as gpm
t_calls, t0 = [], time.perf_counter()
fit_time = [0.0]
_fit = gp.fit
def timed_fit(*a, **k):
    t = time.perf_counter(); r = _fit(*a, **k); torch.cuda.synchronize(); fit_time[0] += time.perf_counter() - t; return r
gp.fit = timed_fit
for step in range(nsteps):
    for a in images[1:-1]:
        t1 = time.perf_counter()
        calc.calculate(a)
        t_calls.append(time.perf_counter() - t1)
        a.positions = a.positions + 0.02 * np.clip(calc.results["forces"], -1.0, 1.0)      # steepest descent step
torch.cuda.synchronize()
wall = time.perf_counter() - t0
print(f"{kind}: {nsteps} steps x {nimg - 2} moving images = {len(t_calls)} force calls in {wall:.2f} s "
      f"(base/surrogate/fit calls {gp.use_base}/{gp.use_surrogate}/{gp.fits}; N = {gp.N_energy} + 3 x {gp.N_forces})")
print(f"  refits {fit_time[0]:.2f} s, base labels {base.seconds:.2f} s, everything else {wall - fit_time[0] - base.seconds:.2f} s; "
      f"median force call {np.median(t_calls) * 1e3:.2f} ms")
