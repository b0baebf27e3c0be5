"""CPU: the product's vectorised neighbour search agrees with the oracle's brute-force image loop
(both restate ASE NeighborList(bothways, no self interaction, skin 0) semantics)."""
import numpy as np
import pytest
from gpr_calculator_amd.neighbors import SimpleAtoms, neighbor_pairs, NeighborListCompat, atomic_number, symbol_of
from oracle.so3_numpy import neighbor_pairs_bruteforce


def canon(c, j, v):
    key = np.lexsort((np.round(v[:, 2], 9), np.round(v[:, 1], 9), np.round(v[:, 0], 9), j, c))
    return c[key], j[key], v[key]


@pytest.mark.parametrize("pbc", [(True, True, True), (True, True, False), (False, False, False)])
def test_pairs_match_bruteforce(pbc):
    rng = np.random.default_rng(3)
    cell = np.array([[4.1, 0.0, 0.0], [1.2, 3.8, 0.0], [0.3, 0.9, 4.4]])
    pos = rng.uniform(-0.3, 1.3, (7, 3)) @ cell           # some atoms outside the cell
    atoms = SimpleAtoms([46] * 7, pos, cell, pbc)
    c1, j1, v1 = canon(*neighbor_pairs(atoms, 5.0))
    c2, j2, v2 = canon(*neighbor_pairs_bruteforce(pos, cell, pbc, 5.0))
    assert len(c1) == len(c2) and np.array_equal(c1, c2) and np.array_equal(j1, j2)
    assert np.abs(v1 - v2).max() < 1e-12
    assert (np.linalg.norm(v1, axis=1) < 5.0).all()


def test_compat_offsets_reproduce_vectors():
    rng = np.random.default_rng(4)
    cell = np.eye(3) * 3.9
    atoms = SimpleAtoms([46, 8, 12], rng.uniform(0, 3.9, (3, 3)), cell, True)
    nl = NeighborListCompat([2.5] * 3, self_interaction=False, bothways=True, skin=0.0)
    nl.update(atoms)
    c, j, v = neighbor_pairs(atoms, 5.0)
    for i in range(3):
        idx, off = nl.get_neighbors(i)
        vec = atoms.positions[idx] + off @ cell - atoms.positions[i]
        assert np.abs(np.sort(np.linalg.norm(vec, axis=1)) - np.sort(np.linalg.norm(v[c == i], axis=1))).max() < 1e-12


def test_periodic_table_lookup():
    assert atomic_number("Pd") == 46 and atomic_number("Au") == 79 and symbol_of(13) == "Al"


def test_native_batch_index_matches_numpy_index():
    """the C++ host routine behind SO3.calculate_batch_dev (csrc/nbr.hip) against the numpy bookkeeping"""
    from gpr_calculator_amd.SO3 import SO3
    rng = np.random.default_rng(5)
    cell = np.array([[4.1, 0.0, 0.0], [1.2, 3.8, 0.0], [0.3, 0.9, 4.4]])
    strucs = [SimpleAtoms(rng.choice([8, 12, 46], 6), rng.uniform(-0.2, 1.2, (6, 3)) @ cell, cell, pbc)
              for pbc in [(True, True, True), (True, False, True), (False, False, False)]]
    strucs.append(SimpleAtoms([46], [[0.1, 0.2, 0.3]], np.eye(3) * 3.0, True))     # one atom, only self images
    for weight_on in (False, True):
        des = SO3(nmax=3, lmax=4, rcut=5.0, weight_on=weight_on)
        nb = des.neighbor_index_batch(strucs)
        p0 = 0
        for k, s in enumerate(strucs):
            ref = des.neighbor_index(s)
            r0, r1 = nb["row0"][k], nb["row0"][k + 1]
            a0, a1 = nb["atom0"][k], nb["atom0"][k + 1]
            assert np.array_equal(nb["seq"][r0:r1], ref["seq"])
            P = len(ref["w"])
            assert np.abs(nb["vec"][p0:p0 + P] - ref["vec"]).max() < 1e-12 if P else True
            assert np.array_equal(nb["w"][p0:p0 + P], ref["w"])
            assert np.array_equal(nb["centre_pair0"][a0:a1 + 1] - p0, ref["centre_pair0"])
            assert np.array_equal(nb["centre_row0"][a0:a1 + 1] - r0, ref["centre_row0"])
            assert np.array_equal(nb["row_pair0"][r0:r1 + 1] - p0, ref["row_pair0"])
            assert np.array_equal(nb["self_row"][a0:a1] - r0, ref["self_row"])
            p0 += P


def test_cur_selection_matches_leverage_score_definition():
    """CUR (gaussianprocess.py:1165-1182): rows ranked by summed squared components in the eigenvectors with
    eigenvalue below the tolerance; as many rows as there are such eigenvectors"""
    from gpr_calculator_amd.gaussianprocess import CUR
    rng = np.random.default_rng(4)
    A = rng.normal(size=(14, 6))
    K = A @ A.T                                  # rank 6: eight near-null directions
    ids = CUR(K, 1e-8)
    lam, U = np.linalg.eigh(K)
    null = [k for k in range(len(lam)) if lam[k] < 1e-8]
    assert len(ids) == len(null) == 8
    score = np.array([sum(U[j, k] ** 2 for k in null) for j in range(len(lam))])
    assert set(int(i) for i in ids) == set(int(i) for i in np.argsort(-score)[:8])
    assert len(CUR(np.eye(5), 1e-10)) == 0


def test_lj_base_potential_energy_forces_stress():
    """LJ base potential (calculator.py:183-283): forces = -dE/dr and total stress = (1/V) dE/d(strain) by finite
    differences; same energies and forces as the label generator of the synthetic benchmark"""
    from gpr_calculator_amd.calculator import LJ
    from gpr_calculator_amd.synthetic import fcc_cell, lj_energy_forces
    from gpr_calculator_amd.neighbors import SimpleAtoms
    atoms = fcc_cell(seed=3, reps=2, sigma=0.08)
    lj = LJ({"epsilon": 0.5, "sigma": 2.5, "rc": 5.0})
    E, F, S = lj.calculate(atoms)
    E0, F0 = lj_energy_forces(atoms)
    assert abs(E - E0) < 1e-10 and np.abs(F - F0).max() < 1e-10 and S.shape == (len(atoms), 6)
    h = 1e-5
    for (i, c) in ((0, 0), (7, 2)):
        p, m = atoms.positions.copy(), atoms.positions.copy()
        p[i, c] += h; m[i, c] -= h
        Ep = lj.calculate(SimpleAtoms(atoms.numbers, p, atoms.cell, atoms.pbc))[0]
        Em = lj.calculate(SimpleAtoms(atoms.numbers, m, atoms.cell, atoms.pbc))[0]
        assert abs(F[i, c] + (Ep - Em) / (2 * h)) < 1e-6
    V = abs(np.linalg.det(atoms.cell))
    for (a, b, v) in ((0, 0, 0), (1, 2, 3)):                     # Voigt: xx -> 0, yz -> 3
        eps = np.zeros((3, 3)); eps[a, b] += h / 2; eps[b, a] += h / 2
        def strained(sign):
            D = np.eye(3) + sign * eps
            return lj.calculate(SimpleAtoms(atoms.numbers, atoms.positions @ D.T, atoms.cell @ D.T, atoms.pbc))[0]
        dE = (strained(+1) - strained(-1)) / (2 * h)
        assert abs(S[:, v].sum() - dE / V) < 1e-6 * max(1.0, abs(dE / V))
    cluster = SimpleAtoms([46, 46, 46], [[0, 0, 0], [2.7, 0, 0], [0, 2.8, 0.1]], np.eye(3) * 20.0, [False] * 3)
    assert lj.calculate(cluster)[2] is None and str(lj).startswith("LJ(") and LJ().save_dict()["rc"] == 5.0


def test_c_gather_matches_numpy_batch_arrays():
    """csrc/pygather.c against neighbors.batch_arrays: same concatenated arrays, fixed-atom detection, and None for
    structures that do not expose plain buffers (the caller then takes the numpy route)"""
    import subprocess, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    try:
        from gpr_calculator_amd import _pygather
    except ImportError:
        subprocess.check_call(["make", "-C", os.path.join(root, "gpr_calculator_amd", "csrc"), os.path.join(root, "gpr_calculator_amd", "_pygather.so")])
        from gpr_calculator_amd import _pygather
    from gpr_calculator_amd.neighbors import SimpleAtoms, batch_arrays
    rng = np.random.default_rng(5)
    strucs = []
    for k in range(5):
        n = 3 + 2 * k
        strucs.append(SimpleAtoms(numbers=rng.choice([8, 46, 78], n), positions=rng.normal(size=(n, 3)),
                                  cell=rng.normal(size=(3, 3)) + 5 * np.eye(3), pbc=[True, k % 2 == 0, False]))
    a0, num, pos, cel, pbc, fixed, cmask, addr = _pygather.gather(strucs)
    ref = batch_arrays(strucs)
    # element 7: the buffers' addresses (atom0, numbers, positions, cells, pbc) and the number of atoms, as the C-ABI takes them
    assert addr[5] == int(ref[0][-1])
    assert [np.frombuffer(b, np.uint8).ctypes.data for b in (a0, num, pos, cel, pbc)] == list(addr[:5])
    assert np.array_equal(np.frombuffer(a0, np.int32), ref[0]) and np.array_equal(np.frombuffer(num, np.int32), ref[1])
    assert np.array_equal(np.frombuffer(pos, np.float64).reshape(-1, 3), ref[2])
    assert np.array_equal(np.frombuffer(cel, np.float64).reshape(-1, 9), ref[3])
    assert np.array_equal(np.frombuffer(pbc, np.int32).reshape(-1, 3), ref[4]) and fixed is False and cmask is None
    strucs[2].fixed = np.array([1, int(np.diff(ref[0])[2]) - 1])
    strucs[4].fixed = np.array([0], dtype=np.int32)
    got = _pygather.gather(strucs)
    want = np.ones(int(ref[0][-1]), np.uint8)
    want[ref[0][2] + 1] = 0; want[ref[0][3] - 1] = 0; want[ref[0][4]] = 0
    assert got[5] is True and np.array_equal(np.frombuffer(got[6], np.uint8), want)
    strucs[1].fixed = np.array([99])              # out of range (or negative): the caller resolves (and reports) it
    assert _pygather.gather(strucs)[6] is None
    strucs[1].fixed = np.zeros(0, dtype=np.int64)
    assert _pygather.gather([]) is not None and len(_pygather.gather([])[1]) == 0

    class Listy:                                  # attributes that are not array buffers
        numbers, positions, cell, pbc = [1, 1], [[0, 0, 0], [1, 0, 0]], np.eye(3).tolist(), [False] * 3
    assert _pygather.gather([strucs[0], Listy()]) is None

    class NonContiguous:
        numbers = np.array([1, 1]); positions = np.zeros((3, 2)).T; cell = np.eye(3); pbc = np.zeros(3, bool)
    assert _pygather.gather([NonContiguous()]) is None
