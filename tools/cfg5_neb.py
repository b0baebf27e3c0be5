"""BASELINE.json configs[4] as a band: examples/Pd4/gp_neb.py in shape -- the Pd4 / MgO(100) cell of the reference's shipped
database (220 atoms: 108 Mg, 108 O, 4 Pd; the 112 atoms that differ between its images move, the other 108 are held by
FixAtoms), 7 images between two of its structures (IDPP interpolation with minimum-image displacements), GP.set_GPR on
the band, one GPR calculator per image with the shared model (image 5 alone may refit, freq = 10), neb_calc with FIRE,
fmax = noise_f, reference energies of the final band from the base calculator -- through this build's NEB.py / calculator.py /
gaussianprocess.py on the GPU.  ASE, VASP and EMT parameters for Mg are not available here: a soft Lennard-Jones (the base of
tests/golden/gpr_trace_pd4.npz) stands in for the base calculator and the built-in band + FIRE for ase.mep.NEB, so the run
shows the loop, its counters and its wall clock, not a barrier.
usage: python tools/cfg5_neb.py [steps] [passes]      (defaults 50 2, as the reference script)"""
import contextlib, io, os, sys, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import numpy as np

P = np.load(os.path.join(ROOT, "tests", "golden", "pd4_subset.npz"))
T = np.load(os.path.join(ROOT, "tests", "golden", "gpr_trace_pd4.npz"))
EPS, SIG = float(T["lj_eps"]), float(T["lj_sig"])


class SoftLJ:
    """base calculator stand-in; like an ASE calculator it keeps the result of the last geometry"""
    name = "lj"

    def __init__(self):
        self.calls, self.seconds, self._key, self._res = 0, 0.0, None, None

    def _eval(self, atoms):
        from gpr_calculator_amd.synthetic import lj_energy_forces
        key = atoms.positions.tobytes()
        if key != self._key:
            t0 = time.perf_counter()
            self._res, self._key = lj_energy_forces(atoms, eps=EPS, sig=SIG), key
            self.calls += 1
            self.seconds += time.perf_counter() - t0
        return self._res

    def get_potential_energy(self, atoms):
        return self._eval(atoms)[0]

    def get_forces(self, atoms):
        return self._eval(atoms)[1].copy()


def endpoints():
    from gpr_calculator_amd.NEB import Image
    p0, p5 = P["positions_0"], P["positions_5"]
    moving = np.flatnonzero(np.abs(p5 - p0).max(axis=1) > 1e-6)
    fixed = np.setdiff1d(np.arange(len(p0)), moving)
    make = lambda pos: Image(P["numbers_0"], pos.copy(), P["cell_0"], P["pbc_0"], fixed=fixed)
    return make(p0), make(p5), moving, fixed


def run(steps=50, passes=2, num_images=7, freq=10, quiet=True, log=None):
    import torch
    from gpr_calculator_amd.NEB import get_images, neb_calc
    from gpr_calculator_amd.calculator import GPR
    from gpr_calculator_amd.gaussianprocess import GP
    say = (lambda *a: None) if log is None else (lambda *a: print(*a, file=log))
    initial, final, moving, fixed = endpoints()
    t0 = time.perf_counter()
    images = get_images(initial, final, num_images, IDPP=True, mic=True, apply_constraint=True)
    t_images = time.perf_counter() - t0
    fixed0 = [im.positions[fixed].copy() for im in images]
    base = SoftLJ()
    noise_f = 0.05
    noise_e = max(0.0002, 0.05 / len(images[0]))                      # examples/Pd4/gp_neb.py:11-12, 28
    sink = io.StringIO()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    with contextlib.redirect_stdout(sink if quiet else sys.stdout):
        gp = GP.set_GPR(images, base, noise_e=noise_e, noise_f=noise_f, json_file=None)
    torch.cuda.synchronize(); t_boot = time.perf_counter() - t0
    say(f"band: {num_images} images of {len(images[0])} atoms ({len(moving)} moving, {len(fixed)} fixed), IDPP + minimum image: {t_images * 1e3:.1f} ms")
    say(f"set_GPR: {gp.N_energy} energies + {gp.N_forces} force points (N = {len(gp.y_train)}), base calls {base.calls}, "
        f"fit(opt=True): {t_boot * 1e3:.1f} ms, of which base calculator {base.seconds * 1e3:.1f} ms; kernel {gp.kernel}")
    for i, image in enumerate(images):
        image.calc = GPR(base=base, ff=gp, freq=freq, tag="Pd4", save=False)
        image.calc.verbose = False
        image.calc.update_gpr = (i == len(images) - 2)                # gp_neb.py:33
    out = []
    for p in range(passes):
        b0, s0, f0, bs0 = gp.use_base, gp.use_surrogate, gp.fits, base.seconds
        torch.cuda.synchronize(); t0 = time.perf_counter()
        with contextlib.redirect_stdout(sink if quiet else sys.stdout):
            neb, refs = neb_calc(images, steps=steps, algo='FIRE', fmax=noise_f, climb=False, use_ref=True)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        calls = (gp.use_base - b0) + (gp.use_surrogate - s0)
        say(f"pass {p}: {neb.nsteps} band steps, converged {neb.converged}; force calls {calls} = base {gp.use_base - b0} + surrogate "
            f"{gp.use_surrogate - s0}, refits {gp.fits - f0}; wall {dt:.3f} s, of which base calculator {base.seconds - bs0:.3f} s "
            f"-> {1e3 * (dt - (base.seconds - bs0)) / max(calls, 1):.2f} ms per force call on the GPR side; N = {len(gp.y_train)}")
        say("        image      surrogate E      base E        |diff|")
        for i, (e, r) in enumerate(zip(neb.energies, refs)):
            say(f"        {i:5d} {e:16.6f} {r:13.6f} {abs(e - r):13.6f}")
        out.append((neb, refs, dt))
        if neb.converged:
            break
    say(f"totals: base / surrogate / fits = {gp.use_base}/{gp.use_surrogate}/{gp.fits}; base calculator calls {base.calls}")
    return gp, base, images, fixed, fixed0, out


if __name__ == "__main__":
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    passes = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    run(steps=3, passes=1)                       # library, tables, allocator warm
    run(steps=steps, passes=passes, log=sys.stdout)
