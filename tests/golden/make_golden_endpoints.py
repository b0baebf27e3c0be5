"""The NEB endpoints of BASELINE.json config 1 (examples/emt-serial.py: examples/database/initial.traj, final.traj --
12 Al + 1 Au, periodic in x and y, atoms 0-7 fixed) decoded from the reference's ASE trajectory files with the
standard library (gpr_calculator_amd.asedb.read_traj; ASE is not installed here), last frame of each, with the EMT
energy / forces stored in them.  Data only.

    python tests/golden/make_golden_endpoints.py        -> tests/golden/emt_endpoints.npz
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.abspath(os.path.join(HERE, "..", "..")))
from gpr_calculator_amd.asedb import read_traj      # noqa: E402

p = {}
for tag in ("initial", "final"):
    fr = read_traj(f"/root/reference/examples/database/{tag}.traj")[-1]
    a = fr["atoms"]
    p.update({f"{tag}_numbers": a.numbers, f"{tag}_positions": a.positions, f"{tag}_cell": a.cell, f"{tag}_pbc": a.pbc,
              f"{tag}_fixed": a.fixed, f"{tag}_energy": fr["energy"], f"{tag}_forces": fr["forces"]})
    print(tag, fr["energy"], a.fixed.tolist())
np.savez_compressed(os.path.join(HERE, "emt_endpoints.npz"), **p)
