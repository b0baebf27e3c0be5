"""BASELINE.json config 1 shape, end to end without ASE: get_images -> GP.set_GPR -> GPR calculator -> neb_calc on the
reference's own end points (12 Al + 1 Au, atoms 0-7 fixed; tests/golden/emt_endpoints.npz), 5 images, climbing image,
on-the-fly learning live.  A soft Lennard-Jones stands in for EMT (ASE is not installed).  Prints the base / surrogate /
fit counters and the wall clock.  usage: python tools/neb_sim.py [RBF|Dot] [steps] [etol]"""
import os, sys, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np
from gpr_calculator_amd.NEB import Image, get_images, neb_calc
from gpr_calculator_amd.calculator import GPR
from gpr_calculator_amd.gaussianprocess import GP
from gpr_calculator_amd.synthetic import lj_energy_forces

kind = sys.argv[1] if len(sys.argv) > 1 else "RBF"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
etol = float(sys.argv[3]) if len(sys.argv) > 3 else 0.05
Z = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "emt_endpoints.npz"))


class SoftLJ:
    name, calls, seconds = "lj", 0, 0.0

    def get_potential_energy(self, atoms):
        t0 = time.perf_counter(); SoftLJ.calls += 1
        e = lj_energy_forces(atoms, eps=0.05)[0]
        SoftLJ.seconds += time.perf_counter() - t0
        return e

    def get_forces(self, atoms):
        return lj_energy_forces(atoms, eps=0.05)[1].copy()


end = lambda t: Image(Z[f"{t}_numbers"], Z[f"{t}_positions"], Z[f"{t}_cell"], Z[f"{t}_pbc"], fixed=Z[f"{t}_fixed"])
images = get_images(end("initial"), end("final"), num_images=5)
t0 = time.perf_counter()
gp = GP.set_GPR(images, SoftLJ(), kernel=kind, noise_e=etol / len(images[0]), noise_f=etol)
t1 = time.perf_counter()
calc = GPR(base=SoftLJ(), ff=gp, save=False)
neb = neb_calc(images, calc, fmax=0.05, steps=steps, climb=True)
t2 = time.perf_counter()
print(gp)
print(f"bootstrap {t1 - t0:.2f} s; band {neb.nsteps} steps in {t2 - t1:.2f} s ({(t2 - t1) / max(gp.use_base + gp.use_surrogate, 1) * 1e3:.2f} ms "
      f"per force call), converged {neb.converged}; base/surrogate/fits {gp.use_base}/{gp.use_surrogate}/{gp.fits}; "
      f"base potential {SoftLJ.seconds:.2f} s; energies {np.round(neb.energies, 4).tolist()}")
