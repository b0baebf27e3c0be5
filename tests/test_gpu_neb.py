"""GPU, BASELINE.json config 1 shape: the 5-image band of examples/emt-serial.py -- 12 Al + 1 Au, periodic in x / y,
atoms 0-7 fixed, end points from the reference's own trajectory files (tests/golden/emt_endpoints.npz) -- run through
get_images -> GP.set_GPR -> GPR calculator -> neb_calc (gpr_calculator_amd/NEB.py, the reference's NEB.py interface)
with the on-the-fly learning loop live.  ASE and its EMT potential are not installed: a soft Lennard-Jones stands in for
the base calculator and the built-in band / FIRE for ase.mep.NEB / ase.optimize, so this pins plumbing and bookkeeping,
not a reaction barrier."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
Z = np.load(os.path.join(os.path.dirname(__file__), "golden", "emt_endpoints.npz"))


class SoftLJ:
    name, calls = "lj", 0

    def get_potential_energy(self, atoms):
        from gpr_calculator_amd.synthetic import lj_energy_forces
        SoftLJ.calls += 1
        return lj_energy_forces(atoms, eps=0.05)[0]

    def get_forces(self, atoms):
        from gpr_calculator_amd.synthetic import lj_energy_forces
        return lj_energy_forces(atoms, eps=0.05)[1].copy()


def endpoint(tag):
    from gpr_calculator_amd.NEB import Image
    return Image(Z[f"{tag}_numbers"], Z[f"{tag}_positions"], Z[f"{tag}_cell"], Z[f"{tag}_pbc"], fixed=Z[f"{tag}_fixed"])


def test_endpoint_fixture_is_the_reference_trajectory():
    # known answers stored in the reference's files (README.md:81,85): EMT energies of the relaxed end points
    assert abs(float(Z["initial_energy"]) - 3.3147542909997236) < 1e-12
    assert abs(float(Z["final_energy"]) - 3.316117208116346) < 1e-12
    assert Z["initial_numbers"].tolist() == [13] * 12 + [79] and Z["initial_fixed"].tolist() == list(range(8))
    assert Z["initial_pbc"].tolist() == [True, True, False]


def test_get_images_and_calculator_assignment():
    from gpr_calculator_amd.NEB import assign_calculators, get_images
    from gpr_calculator_amd.calculator import GPR
    images = get_images(endpoint("initial"), endpoint("final"), num_images=5)
    assert len(images) == 5 and all(len(im) == 13 for im in images)
    d = images[-1].positions - images[0].positions
    for i, im in enumerate(images):
        assert np.allclose(im.positions, images[0].positions + d * i / 4.0) and im.fixed.tolist() == list(range(8))
    calc = GPR(base=SoftLJ(), ff=None, save=False)
    assign_calculators(images, calc)
    assert [im.calc.update_gpr for im in images] == [False, True, False, False, False]       # NEB.py:43-47
    assert len({id(im.calc) for im in images}) == 5


def test_neb_shaped_on_the_fly_run():
    from gpr_calculator_amd.NEB import get_images, neb_calc
    from gpr_calculator_amd.calculator import GPR
    from gpr_calculator_amd.gaussianprocess import GP
    images = get_images(endpoint("initial"), endpoint("final"), num_images=5)
    fixed0 = [im.positions[:8].copy() for im in images]
    base = SoftLJ()
    SoftLJ.calls = 0
    gp = GP.set_GPR(images, base, kernel='RBF', noise_e=0.05 / 13, noise_f=0.05)          # examples/emt-serial.py:21-23
    assert gp.N_energy == 5 and SoftLJ.calls == 5 and gp.fits == 1
    calc = GPR(base=base, ff=gp, save=False)
    calc.verbose = False
    neb = neb_calc(images, calc, fmax=0.05, steps=12, climb=True)
    n_force_calls = gp.use_base + gp.use_surrogate
    assert n_force_calls >= 3 * neb.nsteps and gp.use_surrogate > 0                       # three interior images per step
    assert SoftLJ.calls == 5 + gp.use_base
    assert np.isfinite(neb.energies).all() and len(neb.energies) == 5
    # end points report the energies they were trained on (NEB.py:66-67), interior images the frozen surrogate's
    assert abs(neb.energies[0] - gp.train_y['energy'][0] * 13) < 1e-12
    assert abs(neb.energies[-1] - gp.train_y['energy'][4] * 13) < 1e-12
    for im, f0 in zip(images, fixed0):
        assert np.array_equal(im.positions[:8], f0)                                        # FixAtoms respected
    # the surrogate's interior energies are close to the base potential's on the final band
    from gpr_calculator_amd.synthetic import lj_energy_forces
    for i in (1, 2, 3):
        assert abs(neb.energies[i] - lj_energy_forces(images[i], eps=0.05)[0]) < 0.2
    neb2, ref = neb_calc(images, calc, fmax=10.0, steps=0, use_ref=True)
    assert len(ref) == 5 and all(np.isfinite(ref))
