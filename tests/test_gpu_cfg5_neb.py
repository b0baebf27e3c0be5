"""GPU, BASELINE.json configs[4] as a band (tools/cfg5_neb.py = examples/Pd4/gp_neb.py in shape): seven images of the
220-atom Pd4 / MgO(100) cell, IDPP + minimum-image interpolation, GP.set_GPR on the band, one GPR calculator per image on the
shared model with image 5 as the only one allowed to refit, neb_calc(FIRE, use_ref=True) -- the on-the-fly loop live, driven
by the band optimiser instead of a script.  The base calculator is a soft Lennard-Jones and the band the built-in one
(no ASE here): bookkeeping, constraints and the surrogate's agreement with the base on the final band are what is pinned."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def test_pd4_band_relaxed_through_neb_calc_with_on_the_fly_learning():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import cfg5_neb
    gp, base, images, fixed, fixed0, out = cfg5_neb.run(steps=8, passes=1)
    neb, refs, _ = out[0]
    assert len(images) == 7 and all(len(im) == 220 for im in images) and len(fixed) == 108
    # set_GPR: every image labelled once by the base calculator, energies always queued, one fit
    assert gp.N_energy >= 7 and gp.fits >= 1
    # force calls: five interior images per band step; the final use_ref pass forces the base on each of them once
    calls = gp.use_base + gp.use_surrogate
    assert calls >= 5 * neb.nsteps and gp.use_surrogate > 0
    # 7 bootstrap labels + every base fall-back, nothing else (the stand-in keeps its last result like an ASE calculator, so
    # a forced reference call on a geometry it has just labelled costs no evaluation: at most one per interior image)
    assert 7 + gp.use_base - 5 <= base.calls <= 7 + gp.use_base
    assert gp.N_energy == 7 + gp.use_base                          # every fall-back queues its energy (forced ones included)
    for im, f0 in zip(images, fixed0):
        assert np.array_equal(im.positions[fixed], f0)             # FixAtoms respected by interpolation, band and FIRE
    assert np.isfinite(neb.energies).all() and np.isfinite(refs).all() and len(refs) == 7
    # end points report the energies they were trained on; interior images the frozen surrogate, within its own tolerance
    # of the base calculator on the final band (energy criterion of the calculator: E_std * n vs noise, here 220 atoms)
    assert abs(neb.energies[0] - refs[0]) < 1e-9 and abs(neb.energies[-1] - refs[-1]) < 1e-9
    # (the calculator's acceptance test looks at force uncertainties only -- ignore_E_std -- so a few steps into the run the
    # accepted energies sit within a per cent of the base's, not within the noise level)
    assert np.abs(np.asarray(neb.energies[1:-1]) - np.asarray(refs[1:-1])).max() < 0.01 * np.abs(np.asarray(refs)).max()
    assert [im.calc.update_gpr for im in images] == [False] * 5 + [True, False]
