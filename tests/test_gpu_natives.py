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


def test_unsupported_descriptor_length_returns_instead_of_aborting():
    """Descriptors longer than the widest operand plane of this build (d > 128; the reference loops over any d,
    rbf_kernel.cpp:341-473): the drop-in natives must say so and hand the caller's zeroed buffers back untouched (they
    are `void`), not kill the host process.  (d <= 128 runs: natives_d50_mixed, _d75_mixed, _d105_zeta3, _d128_subrange.)"""
    import ctypes as C
    from gpr_calculator_amd import _lib
    from oracle.natives import SIGS, make_case
    L = C.CDLL(_lib.LIB_PATH)
    c = make_case(5, 130, [3, 4], [2, 5, 3], species=(8, 46))
    from oracle.natives import expand_counts
    n1, n2, m1, m2 = len(c["x1"]), len(c["x2"]), 2, 3
    fn = L.rbf_kff_many
    fn.argtypes, fn.restype = SIGS["rbf_kff_many"], None
    out = np.zeros((m1, 3, m2, 3))
    p = lambda a, t: a.ctypes.data_as(t)
    D, I = C.POINTER(C.c_double), C.POINTER(C.c_int)
    x1, x2 = np.ascontiguousarray(c["x1"]), np.ascontiguousarray(c["x2"])
    A, B = np.ascontiguousarray(c["dx1dr"]), np.ascontiguousarray(c["dx2dr"])
    e1, e2, i1, i2 = c["ele1"].astype(np.int32), c["ele2"].astype(np.int32), expand_counts(c["counts1"]), expand_counts(c["counts2"])
    fn(n1, n2, 0, n2, 130, m2, 2.0, 6.25, 1.69, 1e-12, p(x1, D), p(A, D), p(e1, I), p(i1, I), p(x2, D), p(B, D), p(e2, I), p(i2, I), p(out, D))
    L.gpr_l0_last_status.restype = C.c_int
    assert L.gpr_l0_last_status() == -3 and not out.any()                  # GPR_ERR_UNSUPPORTED, nothing written
    # ... and the library is still usable afterwards
    c30 = make_case(5, 30, [3, 4], [2, 5, 3], species=(8, 46))
    got = run_all_natives(NativeSet("hip", libs=[_lib.LIB_PATH], prefix=""), c30)
    assert np.isfinite(got["rbf_kff_many"]).all() and got["rbf_kff_many"].any() and L.gpr_l0_last_status() == 0
