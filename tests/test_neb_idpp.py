"""IDPP interpolation without ASE (gpr_calculator_amd/NEB.py: get_images(IDPP=True), idpp_interpolate) -- what the
reference gets from ase.mep.NEB.interpolate(method='idpp') (gpr_calc/NEB.py:130-134).  ASE is not in this image, so
the checks are properties of the method: the forces are the gradient of the objective, the end points and fixed atoms
do not move, the relaxed band lowers the objective and keeps atoms apart where the straight line drives them through
each other."""
import numpy as np
import pytest

from gpr_calculator_amd.NEB import (Band, Image, PairDistancePotential, _all_distances, get_images, idpp_interpolate)


def _cluster(seed=0, n=7, box=6.0, pbc=False):
    rng = np.random.default_rng(seed)
    pos = rng.uniform(0.5, box - 0.5, size=(n, 3))
    return Image(np.full(n, 14), pos, np.eye(3) * box, np.array([pbc] * 3), fixed=np.array([], dtype=int))


@pytest.mark.parametrize("mic", [False, True])
def test_forces_are_the_gradient(mic):
    at = _cluster(1, pbc=mic)
    target = _all_distances(_cluster(2, pbc=mic), mic)[1]
    pot = PairDistancePotential(target, mic=mic)
    e0, f = pot.evaluate(at)
    h = 1e-6
    for (i, c) in [(0, 0), (3, 1), (6, 2)]:
        p = at.copy(); p.positions[i, c] += h
        m = at.copy(); m.positions[i, c] -= h
        num = -(pot.evaluate(p)[0] - pot.evaluate(m)[0]) / (2 * h)
        assert abs(num - f[i, c]) < 1e-6 * max(1.0, abs(f[i, c]))
    assert e0 > 0 and np.abs(f.sum(axis=0)).max() < 1e-9 * np.abs(f).max()      # pair forces: no net force


def _pass_through():
    """atom 0 travels along x past atom 1 at a distance of 0.3: the straight line puts them on top of each other"""
    pos0 = np.array([[-2.0, 0.3, 0.0], [0.0, 0.0, 0.0], [0.0, 2.5, 0.0], [0.0, -2.5, 0.3], [2.5, 0.0, 2.5], [-2.5, 0.2, -2.5]])
    pos1 = pos0.copy(); pos1[0] = [2.0, 0.3, 0.0]
    cell, pbc = np.eye(3) * 20.0, np.array([False] * 3)
    mk = lambda p: Image(np.full(len(p), 46), p + 10.0, cell, pbc, fixed=np.array([1, 2], dtype=int))
    return mk(pos0), mk(pos1)


def _objective(images):
    n = len(images)
    d1, d2 = _all_distances(images[0], False)[1], _all_distances(images[-1], False)[1]
    return [PairDistancePotential(d1 + (d2 - d1) * i / (n - 1.0)).evaluate(im)[0] for i, im in enumerate(images)]


def test_band_goes_round_instead_of_through():
    a, b = _pass_through()
    lin = get_images(a, b, num_images=5, IDPP=False, apply_constraint=True)
    idp = get_images(a, b, num_images=5, IDPP=True, apply_constraint=True)
    assert all(im.calc is None for im in idp)                                    # the pair potential is gone again
    for k in (0, -1):
        assert np.array_equal(idp[k].positions, lin[k].positions)               # end points untouched
    for im_l, im_i in zip(lin, idp):
        assert np.array_equal(im_i.positions[[1, 2]], im_l.positions[[1, 2]])    # fixed atoms untouched
    d_lin = np.linalg.norm(lin[2].positions[0] - lin[2].positions[1])
    d_idp = np.linalg.norm(idp[2].positions[0] - idp[2].positions[1])
    assert d_lin == pytest.approx(0.3) and d_idp > 3 * d_lin                     # target there: 2.02 (stops at fmax = 0.1)
    s_lin, s_idp = _objective(lin), _objective(idp)
    assert sum(s_idp) < 0.05 * sum(s_lin)
    assert s_idp[0] == 0.0 and s_idp[-1] == 0.0


def test_converges_and_reports():
    a, b = _pass_through()
    images = get_images(a, b, num_images=7, IDPP=False)
    band = idpp_interpolate(images, fmax=0.1, steps=200)
    assert isinstance(band, Band) and band.converged and 0 < band.nsteps <= 200
    # already relaxed: a second call stops at once
    again = idpp_interpolate(images, fmax=0.1, steps=200)
    assert again.converged and again.nsteps == 0
