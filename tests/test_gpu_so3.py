"""GPU parity of the SO3 descriptor kernel against the golden vectors of the reference's own
SO3.calculate and against the oracle on the synthetic 32-atom cells of the benchmark.
Tolerance: 1e-11 relative to the largest |x| (|dxdr|) of the structure."""
import glob
import os
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(__file__)
GOLD = sorted(p for p in glob.glob(os.path.join(HERE, "golden", "so3_*.npz")))
REL = 1e-11


def rel(a, b):
    return np.abs(a - b).max() / (np.abs(b).max() + 1e-300)


@pytest.mark.parametrize("path", GOLD, ids=[os.path.basename(p)[4:-4] for p in GOLD])
def test_so3_matches_reference_golden(path):
    from gpr_calculator_amd.SO3 import SO3
    from gpr_calculator_amd.neighbors import SimpleAtoms
    z = np.load(path)
    atoms = SimpleAtoms(z["numbers"], z["positions"], z["cell"], z["pbc"])
    des = SO3(nmax=int(z["nmax"]), lmax=int(z["lmax"]), rcut=float(z["rcut"]), alpha=float(z["alpha"]))
    out = des.calculate(atoms)
    assert np.array_equal(out["seq"], z["seq"])
    assert rel(out["x"], z["x"]) < REL and rel(out["dxdr"], z["dxdr"]) < REL
    assert out["rdxdr"] is None and len(out["elements"]) == len(z["numbers"])


def test_so3_batch_matches_oracle_on_bench_cells():
    from gpr_calculator_amd.SO3 import SO3
    from gpr_calculator_amd.synthetic import fcc_cell
    from oracle.so3_numpy import so3_calculate
    des = SO3(nmax=3, lmax=4, rcut=5.0, alpha=2.0)
    strucs = [fcc_cell(seed=s, species=(46, 78) if s % 2 else (46,)) for s in range(3)]
    out = des.calculate_batch_dev(strucs)
    x, dxdr = out["x"].cpu().numpy(), out["dxdr"].cpu().numpy()
    for k, s in enumerate(strucs):
        ref = so3_calculate(s.numbers, s.positions, s.cell, s.pbc, 3, 4, 5.0, 2.0)
        a0, a1, r0, r1 = out["atom0"][k], out["atom0"][k + 1], out["row0"][k], out["row0"][k + 1]
        assert np.array_equal(out["seq"][k], ref["seq"])
        assert rel(x[a0:a1], ref["x"]) < REL and rel(dxdr[r0:r1], ref["dxdr"]) < REL
