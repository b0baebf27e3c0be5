"""NEB helpers with the interface of the reference's gpr_calc/NEB.py (:10-138): `get_images` builds the initial band
between two end points, `neb_calc` relaxes it with the given calculator -- the GPR calculator learns on the fly while
the band moves (calculator.py), image 1 alone being allowed to trigger refits (NEB.py:43-47).

Host orchestration only; the arithmetic of every force call is the GPU path behind `GPR.calculate`.  With ASE
installed both functions hand over to ase.mep.NEB and ase.optimize exactly as the reference does.  Without ASE (this
build's containers) the band is a `Band` of `Image`s: linear interpolation (optionally minimum-image), improved-tangent
nudged elastic band forces with spring constant k and an optional climbing image (Henkelman & Jonsson, J. Chem. Phys.
113, 9978 (2000); Henkelman, Uberuaga & Jonsson, ibid. 9901 -- what ase.mep.NEB implements), relaxed by FIRE.  IDPP
interpolation (`get_images(..., IDPP=True)`, which the reference delegates to ase.mep.NEB.interpolate(method='idpp')) is
provided the same way: the image-dependent pair potential of Smidstrup et al. (J. Chem. Phys. 140, 214106 (2014)) with
linearly interpolated pair distances as targets, the linearly interpolated band relaxed on it to fmax = 0.1 in at most 100
steps -- ASE's procedure and defaults, with the built-in FIRE in place of ASE's MDMin.
"""
import os
from copy import copy

import numpy as np

from .neighbors import SimpleAtoms, fixed_indices

try:                                            # pragma: no cover - depends on the environment
    from ase.mep import NEB as _AseNEB
    HAVE_ASE = True
except Exception:
    _AseNEB = None
    HAVE_ASE = False


class Image(SimpleAtoms):
    """what the band needs of an ase.Atoms: a calculator slot, energies / forces through it, copies"""
    calc = None

    def get_potential_energy(self):
        return self.calc.get_potential_energy(self)

    def get_forces(self):
        return self.calc.get_forces(self)

    def copy(self):
        return Image(self.numbers.copy(), self.positions.copy(), self.cell.copy(), self.pbc.copy(), self.fixed.copy())

    @classmethod
    def of(cls, atoms):
        if isinstance(atoms, cls):
            return atoms
        return cls(np.asarray(atoms.numbers), np.asarray(atoms.positions), np.asarray(atoms.cell), atoms.pbc,
                   fixed=fixed_indices(atoms))


def _minimum_image(d, cell, pbc):
    """displacements d [n, 3] folded to the nearest periodic image along the periodic axes (ase.geometry.find_mic
    for cells whose shortest image is among the nearest neighbours of the reduced coordinates)"""
    if not np.any(pbc) or abs(np.linalg.det(cell)) < 1e-12:
        return d
    s = np.linalg.solve(cell.T, d.T).T
    s -= np.where(pbc, np.round(s), 0.0)
    return s @ cell


def assign_calculators(images, calculator):
    """every image gets its own copy of the calculator; with the GPR calculator only image 1 may trigger a refit of the
    shared model (NEB.py:39-47)"""
    for i, image in enumerate(images):
        image.calc = copy(calculator)
        if getattr(calculator, "name", None) == 'gpr':
            image.calc.update_gpr = (i == 1)


def get_images(init, final, num_images=5, vaccum=0.0, traj=None, IDPP=False, mic=False, apply_constraint=False):
    """Initial band of `num_images` images (end points included) between two structures (NEB.py:92-138).
    init / final: structure files (ASE needed for anything but .traj) or atoms objects."""
    if traj is not None and os.path.exists(traj):
        if HAVE_ASE:
            from ase.io import read
            return read(traj, index=':')[-num_images:]
        from .asedb import read_traj
        return [Image.of(f["atoms"]) for f in read_traj(traj)[-num_images:]]

    def load(x):
        if not isinstance(x, (str, os.PathLike)):
            return x
        if HAVE_ASE:
            from ase.io import read
            return read(x)
        if str(x).endswith(".traj"):
            from .asedb import read_traj
            return Image.of(read_traj(x)[-1]["atoms"])
        raise ImportError(f"reading {x} needs ASE (only .traj files are read without it)")
    initial, final = load(init), load(final)
    if HAVE_ASE and not isinstance(initial, SimpleAtoms):          # the reference's own path
        if initial.pbc[-1] and vaccum > 0:
            for atoms in (initial, final):
                atoms.cell[2, 2] += vaccum
                atoms.center()
                atoms.pbc = [True, True, True]
        images = [initial] + [initial.copy() for _ in range(num_images - 2)] + [final]
        neb = _AseNEB(images, parallel=False)
        if IDPP:
            neb.interpolate(method='idpp', mic=mic, apply_constraint=apply_constraint)
        else:
            neb.interpolate(apply_constraint=apply_constraint, mic=mic)
        return images
    initial, final = Image.of(initial), Image.of(final)
    if initial.pbc[-1] and vaccum > 0:
        for atoms in (initial, final):
            atoms.cell[2, 2] += vaccum
            atoms.positions += 0.5 * atoms.cell.sum(axis=0) - 0.5 * (atoms.positions.min(axis=0) + atoms.positions.max(axis=0))
            atoms.pbc = np.array([True, True, True])
    images = [initial] + [initial.copy() for _ in range(num_images - 2)] + [final]
    d = final.positions - initial.positions
    if mic:
        d = _minimum_image(d, initial.cell, initial.pbc)
    if not apply_constraint and len(initial.fixed):               # ase: the interpolation moves fixed atoms too unless asked
        pass
    for i in range(1, num_images - 1):
        step = d * (i / (num_images - 1.0))
        if apply_constraint and len(initial.fixed):
            step[initial.fixed] = 0.0
        images[i].positions = initial.positions + step
    if IDPP:
        idpp_interpolate(images, mic=mic)
    return images


def _all_distances(atoms, mic):
    """D[i, j] = r_j - r_i (minimum image if asked) and their lengths"""
    P = atoms.positions
    n = len(P)
    D = P[None, :, :] - P[:, None, :]
    if mic:
        D = _minimum_image(D.reshape(-1, 3), atoms.cell, atoms.pbc).reshape(n, n, 3)
    return D, np.sqrt((D ** 2).sum(axis=-1))


class PairDistancePotential:
    """Image-dependent pair potential: S = sum_{i<j} w(d_ij) (d_ij - target_ij)^2 with w = d^-4, and its forces
    -dS/dr (the objective ase.mep.neb.IDPP evaluates).  Calculator-shaped: one instance per image."""
    name = 'idpp'

    def __init__(self, target, mic=False):
        self.target, self.mic = target, mic

    def evaluate(self, atoms):
        D, d = _all_distances(atoms, self.mic)
        dd = d - self.target
        np.fill_diagonal(d, 1.0)                                   # no self-pairs: dd is zero there
        e = 0.5 * (dd ** 2 / d ** 4).sum()
        f = -2.0 * ((dd * (1.0 - 2.0 * dd / d) / d ** 5)[..., None] * D).sum(axis=0)
        return e, f

    def get_potential_energy(self, atoms):
        return self.evaluate(atoms)[0]

    def get_forces(self, atoms):
        return self.evaluate(atoms)[1]


def idpp_interpolate(images, mic=False, fmax=0.1, steps=100, k=0.1):
    """Refine an (already linearly interpolated) band in place on the image-dependent pair potential: the target pair
    distances of image i are those of the first image plus i / (n - 1) of the change to the last; the interior images
    relax under nudged-elastic-band forces of that objective (fixed atoms stay where they are).  The images' own
    calculators are put back afterwards.  Returns the band object (`.nsteps`, `.converged`)."""
    n = len(images)
    d1, d2 = _all_distances(images[0], mic)[1], _all_distances(images[-1], mic)[1]
    saved = [im.calc for im in images]
    for i, im in enumerate(images):
        im.calc = PairDistancePotential(d1 + (d2 - d1) * (i / (n - 1.0)), mic=mic)
    band = Band(images, k=k)
    try:
        _fire(band, fmax, steps)
    finally:
        for im, c in zip(images, saved):
            im.calc = c
    return band


class Band:
    """nudged elastic band over Images (what ase.mep.NEB(images, k=k, parallel=False) provides to an optimiser)"""

    def __init__(self, images, k=0.1, climb=False):
        self.images, self.k, self.climb = images, k, climb
        self.nimages, self.natoms = len(images), len(images[0])
        self.energies = np.zeros(self.nimages)
        self.nsteps, self.converged = 0, False
        self._ends_done = False

    def get_positions(self):
        return np.concatenate([im.positions for im in self.images[1:-1]])

    def set_positions(self, pos):
        n = self.natoms
        for k, im in enumerate(self.images[1:-1]):
            im.positions = pos[k * n:(k + 1) * n].copy()

    def get_forces(self):
        """projected forces on the interior images [(nimages - 2) natoms, 3]: true force perpendicular to the
        improved tangent + spring force along it; the climbing image gets the true force with its tangent component
        reversed.  Interior images are evaluated in order, one calculator call each, as ase.mep.NEB does serially."""
        im = self.images
        if not self._ends_done:
            self.energies[0], self.energies[-1] = im[0].get_potential_energy(), im[-1].get_potential_energy()
            self._ends_done = True
        F = []
        for i in range(1, self.nimages - 1):
            self.energies[i] = im[i].get_potential_energy()
            F.append(np.array(im[i].get_forces(), dtype=float))
        cell, pbc = im[0].cell, im[0].pbc
        disp = [_minimum_image(im[i + 1].positions - im[i].positions, cell, pbc) for i in range(self.nimages - 1)]
        imax = 1 + int(np.argmax(self.energies[1:-1]))
        out = []
        for i in range(1, self.nimages - 1):
            t1, t2 = disp[i - 1], disp[i]
            e0, e1, e2 = self.energies[i - 1], self.energies[i], self.energies[i + 1]
            if e2 > e1 > e0:
                tan = t2.copy()
            elif e0 > e1 > e2:
                tan = t1.copy()
            else:                                                  # at an extremum: energy-weighted mix
                dmax, dmin = max(abs(e2 - e1), abs(e0 - e1)), min(abs(e2 - e1), abs(e0 - e1))
                tan = t2 * dmax + t1 * dmin if e2 > e0 else t2 * dmin + t1 * dmax
            nt = np.linalg.norm(tan)
            tan = tan / nt if nt > 0 else tan
            f = F[i - 1]
            ft = np.vdot(f, tan)
            if self.climb and i == imax:
                f = f - 2.0 * ft * tan
            else:
                f = f - ft * tan + self.k * (np.linalg.norm(t2) - np.linalg.norm(t1)) * tan
            f[im[i].fixed] = 0.0
            out.append(f)
        return np.concatenate(out)


def _fire(band, fmax, steps, dt=0.1, dtmax=1.0, maxstep=0.2):
    """FIRE (Bitzek et al., PRL 97, 170201 (2006)) with ase.optimize.FIRE's default parameters"""
    v, a, nsteps_pos = None, 0.1, 0
    for it in range(steps + 1):
        f = band.get_forces()
        band.nsteps = it
        if np.sqrt((f ** 2).sum(axis=1).max()) < fmax:
            band.converged = True
            return
        if it == steps:
            return
        if v is None:
            v = np.zeros_like(f)
        else:
            if np.vdot(f, v) > 0.0:
                v = (1.0 - a) * v + a * f / np.sqrt(np.vdot(f, f)) * np.sqrt(np.vdot(v, v))
                if nsteps_pos > 5:
                    dt, a = min(dt * 1.1, dtmax), a * 0.99
                nsteps_pos += 1
            else:
                v, a, dt, nsteps_pos = v * 0.0, 0.1, dt * 0.5, 0
        v = v + dt * f
        dr = dt * v
        nrm = np.sqrt(np.vdot(dr, dr))
        if nrm > maxstep:
            dr = maxstep * dr / nrm
        band.set_positions(band.get_positions() + dr)


def neb_calc(images, calculator=None, algo='BFGS', fmax=0.05, steps=100, k=0.1, climb=False, traj=None, use_ref=False):
    """Relax the band (NEB.py:10-90) and return the NEB object with `.images`, `.energies`, `.nsteps`, `.converged`;
    with use_ref also the base-calculator energies of the interior images."""
    ase_images = HAVE_ASE and not isinstance(images[0], SimpleAtoms)
    if ase_images:
        from ase.io.trajectory import Trajectory
        from ase.optimize import BFGS, FIRE
        neb = _AseNEB(images, k=k, parallel=False)
        neb.climb = climb
    else:
        neb = Band(images, k=k, climb=climb)
    if calculator is not None:
        assign_calculators(images, calculator)
    if ase_images:
        if algo == 'BFGS':
            opt = BFGS(neb, trajectory=traj, append_trajectory=True)
        elif algo == 'FIRE':
            opt = FIRE(neb, trajectory=Trajectory(traj, 'a')) if traj is not None else FIRE(neb)
        else:
            raise ValueError('Invalid algorithm for NEB calculation')
        opt.run(fmax=fmax, steps=steps)
        neb.nsteps = opt.nsteps + 1
        neb.converged = opt.converged()
    else:
        if algo not in ('BFGS', 'FIRE'):
            raise ValueError('Invalid algorithm for NEB calculation')
        _fire(neb, fmax, steps)                                   # without ASE both names run the built-in FIRE
        neb.nsteps += 1
    for i, image in enumerate(images):
        if getattr(image.calc, "name", None) == 'gpr':
            if i in (0, len(images) - 1):                         # end points: the reference energy they were trained on
                neb.energies[i] = image.calc.parameters.ff.train_y['energy'][i] * len(image)
            else:
                image.calc.freeze()
                neb.energies[i] = image.get_potential_energy()
                image.calc.unfreeze()
        else:
            neb.energies[i] = image.get_potential_energy()
    if not use_ref:
        return neb
    ref_engs = []
    for i, image in enumerate(images):
        if i in (0, len(images) - 1):
            ref_engs.append(neb.energies[i])
        else:                                                     # one forced base-calculator call per interior image
            image.calc.results = image.calc.result = {}
            if hasattr(image, "arrays") and 'forces' in image.arrays:
                del image.arrays['forces']
            image.calc.force_base = True
            ref_engs.append(image.get_potential_energy())
            image.calc.force_base = False
    return neb, ref_engs
