"""CPU: which atoms of a labelled structure become force training points (GP._pick_force_points, the selection rule of
the reference's add_structure, gaussianprocess.py:974-992): checked against a plain walk through the rule as SURVEY.md
App. A.17 states it -- candidate when max F_std > tol or the FLAT force entry number `atom id` is off by more than 1.5 tol
(upstream's indexing), taken in atom order when the descriptor is new among those taken (cosine rule of new_pt), until
N_max are in -- including the corners: N_max reached exactly, N_max = 0, duplicates of atom 0, no candidates."""
import numpy as np

from gpr_calculator_amd.gaussianprocess import GP
from gpr_calculator_amd.utilities import new_pt


def walk(x, ele, F, F1, Fs, tol, N_max):
    taken, seen = [], []
    for a in range(len(Fs)):
        cand = Fs[a].max() > tol or abs(F[a] - F1[a]) > 1.5 * tol
        if cand and (not seen or new_pt((x[a], ele[a]), seen)):
            taken.append(a)
            seen.append((x[a], ele[a]))
        if len(taken) == N_max:
            break
    return taken


def test_selection_rule_on_random_structures():
    rng = np.random.default_rng(0)
    for _ in range(500):
        n, N_max = int(rng.integers(1, 14)), int(rng.integers(0, 6))
        x = rng.random((n, 6))
        x[rng.integers(0, n)] = x[0]                              # a duplicate of atom 0's descriptor somewhere
        ele = rng.choice([8, 46], n)
        F = rng.normal(size=3 * n)
        F1 = F + rng.normal(scale=0.2, size=3 * n)
        Fs = rng.random((n, 3)) * 0.3
        assert GP._pick_force_points(x, ele, F, F1, Fs, 0.12, N_max) == walk(x, ele, F, F1, Fs, 0.12, N_max)


def test_corners():
    x = np.eye(4)
    ele = np.array([46, 46, 46, 46])
    F = np.zeros(12)
    sure, unsure = np.zeros((4, 3)), np.ones((4, 3))
    assert GP._pick_force_points(x, ele, F, F, sure, 0.1, 20) == []                 # nothing uncertain, nothing off
    assert GP._pick_force_points(x, ele, F, F, unsure, 0.1, 20) == [0, 1, 2, 3]
    assert GP._pick_force_points(x, ele, F, F, unsure, 0.1, 2) == [0, 1]
    same = np.ones((4, 4))                                                          # one descriptor four times: first only
    assert GP._pick_force_points(same, ele, F, F, unsure, 0.1, 20) == [0]
    F1 = F.copy()
    F1[2] = 1.0                                                                     # flat entry 2 = fz of atom 0, read as "atom 2"
    assert GP._pick_force_points(x, ele, F, F1, sure, 0.1, 20) == [2]
