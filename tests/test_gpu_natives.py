"""GPU parity of the 13 legacy natives (L0 C-ABI of libgprhip.so) against the golden vectors of the
unmodified reference C++ and against the plain-C oracle on larger seeded inputs.
Tolerance: 1e-12 relative to the largest magnitude of each output block (FP64 end to end)."""
import glob
import os
import numpy as np
import pytest
from oracle.natives import NativeSet, run_all_natives, make_case

pytestmark = pytest.mark.gpu
GOLD = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "natives_*.npz")))
REL_TOL = 1e-12


@pytest.fixture(scope="module")
def hip_natives():
    from gpr_calculator_amd import _lib
    _lib.require_gpu()
    _lib.lib()
    return NativeSet("hip", libs=[_lib.LIB_PATH], prefix="")


def _compare(got, want):
    worst = {}
    for k in want:
        scale = np.abs(want[k]).max() + 1e-300
        worst[k] = np.abs(got[k] - want[k]).max() / scale
    bad = {k: v for k, v in worst.items() if not v < REL_TOL}
    assert not bad, bad


@pytest.mark.parametrize("path", GOLD, ids=[os.path.basename(p)[8:-4] for p in GOLD])
def test_natives_match_reference_golden(path, hip_natives):
    z = np.load(path)
    case = {k: z[k] for k in z.files if not k.startswith("out/")}
    gold = {k[4:]: z[k] for k in z.files if k.startswith("out/")}
    _compare(run_all_natives(hip_natives, case), gold)


@pytest.mark.parametrize("seed,species", [(11, (46,)), (12, (8, 12, 46))])
def test_natives_match_oracle_ragged(seed, species, hip_natives, port_natives):
    rng = np.random.default_rng(seed)
    counts1 = rng.integers(1, 60, size=23).tolist()
    counts2 = rng.integers(1, 70, size=37).tolist() + [0, 3]      # an empty point in set 2
    case = make_case(seed, 30, counts1, counts2, species=species)
    _compare(run_all_natives(hip_natives, case), run_all_natives(port_natives, case))
