"""CPU: the oracle's numpy restatement of SO3 (and of the numpy diag helpers) reproduces the golden
vectors produced by the reference's own Python (tests/golden/make_golden_python.py)."""
import glob
import os
import numpy as np
import pytest
from oracle import so3_numpy, gp_numpy

HERE = os.path.dirname(__file__)
SO3_GOLD = sorted(glob.glob(os.path.join(HERE, "golden", "so3_*.npz")))


@pytest.mark.parametrize("path", SO3_GOLD, ids=[os.path.basename(p)[4:-4] for p in SO3_GOLD])
def test_so3_oracle_matches_reference_golden(path):
    z = np.load(path)
    out = so3_numpy.so3_calculate(z["numbers"], z["positions"], z["cell"], z["pbc"], int(z["nmax"]), int(z["lmax"]),
                                  float(z["rcut"]), float(z["alpha"]), stress="rdxdr" in z.files)
    assert np.array_equal(out["seq"], z["seq"])
    for key in ("x", "dxdr") + (("rdxdr",) if "rdxdr" in z.files else ()):
        scale = np.abs(z[key]).max()
        assert np.abs(out[key] - z[key]).max() / scale < 1e-12, key


def test_diag_helpers_match_reference_golden():
    z = np.load(os.path.join(HERE, "golden", "diag_helpers.npz"))
    x1, dx1, ele = z["x1"], z["dx1"], z["ele"]
    mask = gp_numpy.get_mask(ele, ele)
    rel = lambda a, b: np.abs(a - b).max() / np.abs(b).max()
    assert rel(gp_numpy.K_ee_RBF(x1, x1, 2.5 ** 2, 1.3 ** 2, 2.0, mask), z["K_ee_RBF"]) < 1e-13
    assert rel(gp_numpy.K_ee_RBF(x1, x1, 2.5 ** 2, 1.3 ** 2, 2.0, None), z["K_ee_RBF_nomask"]) < 1e-13
    assert rel(gp_numpy.K_ee_RBF(x1, x1, 1.7 ** 2, 0.9 ** 2, 3.0, mask), z["K_ee_RBF_z3"]) < 1e-13
    assert rel(gp_numpy.K_ee_dot(x1, x1, 4.0, 4.0, 2.0, mask), z["K_ee_dot"]) < 1e-13
    assert rel(gp_numpy.K_ff_dot(x1, x1, dx1, dx1, 4.0, 4.0, 2.0, mask), z["K_ff_dot"]) < 1e-12
    assert rel(gp_numpy.K_ff_dot(x1, x1, dx1, dx1, 1.7 ** 2, 1.5 ** 2, 3.0, mask), z["K_ff_dot_z3"]) < 1e-12
