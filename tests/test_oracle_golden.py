"""CPU: the oracle's C restatement of the 13 natives reproduces the golden vectors that the
unmodified reference C++ produced (tests/golden/make_golden_natives.py)."""
import glob
import os
import numpy as np
import pytest
from oracle.natives import run_all_natives, NativeSet

GOLD = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "natives_*.npz")))


def _load(path):
    z = np.load(path)
    case = {k: z[k] for k in z.files if not k.startswith("out/")}
    out = {k[4:]: z[k] for k in z.files if k.startswith("out/")}
    return case, out


def test_fixtures_present():
    assert len(GOLD) >= 6


@pytest.mark.parametrize("path", GOLD, ids=[os.path.basename(p)[8:-4] for p in GOLD])
def test_port_matches_reference_golden(path, port_natives):
    case, gold = _load(path)
    got = run_all_natives(port_natives, case)
    assert set(got) == set(gold) and len(gold) == 15   # 13 natives, two with a second output
    for k in gold:
        scale = np.abs(gold[k]).max() + 1e-300
        assert np.abs(got[k] - gold[k]).max() / scale < 1e-13, k


@pytest.mark.skipif(not NativeSet.available("ref"), reason="oracle/_ref not built here")
def test_ref_build_matches_its_own_golden():
    ref = NativeSet("ref")
    case, gold = _load(GOLD[0])
    got = run_all_natives(ref, case)
    for k in gold:
        assert np.array_equal(got[k], gold[k]), k
