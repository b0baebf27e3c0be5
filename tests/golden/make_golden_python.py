"""Generate golden vectors from the reference's importable numpy/scipy Python (build container only).

    python tests/golden/make_golden_python.py

What is imported from /root/reference (read-only) and how:
  * gpr_calc/kernels/base.py         -- plain file import (numpy only): K_ee_RBF, get_mask
  * gpr_calc/kernels/Dot_mb.py       -- its module-top `from .dot_kernel import ...` needs cffi, which is
                                        not installed; a placeholder module object is registered under that
                                        name so the two pure-numpy functions K_ee / K_ff can be imported
  * gpr_calc/SO3.py                  -- its module-top `from ase.neighborlist import NeighborList` needs ASE,
                                        which is not installed; gpr_calculator_amd.neighbors.NeighborListCompat
                                        (same three calls, same semantics) is registered under that name.
                                        Everything numerical (compute_dcs, W, g, phi, quadrature, Bessel, Ylm,
                                        power-spectrum assembly) is the reference's own code.
Outputs (data only): tests/golden/diag_helpers.npz, tests/golden/so3_*.npz
"""
import importlib.util
import os
import sys
import types
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
REF = "/root/reference"
sys.path.insert(0, ROOT)
from gpr_calculator_amd.neighbors import NeighborListCompat, SimpleAtoms  # noqa: E402


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def import_reference():
    pkg = types.ModuleType("gpr_calc"); pkg.__path__ = [os.path.join(REF, "gpr_calc")]
    kpk = types.ModuleType("gpr_calc.kernels"); kpk.__path__ = [os.path.join(REF, "gpr_calc", "kernels")]
    sys.modules["gpr_calc"], sys.modules["gpr_calc.kernels"] = pkg, kpk
    base = _load("gpr_calc.kernels.base", os.path.join(REF, "gpr_calc", "kernels", "base.py"))
    placeholder = types.ModuleType("gpr_calc.kernels.dot_kernel")
    placeholder.kee_C = placeholder.kff_C = placeholder.kef_C = None
    sys.modules["gpr_calc.kernels.dot_kernel"] = placeholder
    dot = _load("gpr_calc.kernels.Dot_mb", os.path.join(REF, "gpr_calc", "kernels", "Dot_mb.py"))
    ase = types.ModuleType("ase"); nl = types.ModuleType("ase.neighborlist")
    nl.NeighborList = NeighborListCompat; nl.PrimitiveNeighborList = None
    sys.modules["ase"], sys.modules["ase.neighborlist"] = ase, nl
    so3 = _load("gpr_calc.SO3", os.path.join(REF, "gpr_calc", "SO3.py"))
    return base, dot, so3


def structures():
    rng = np.random.default_rng(0)
    out = {}
    a = 3.89
    fcc = np.array([[0, 0, 0], [0.5, 0.5, 0], [0.5, 0, 0.5], [0, 0.5, 0.5]]) * a
    out["fcc4_pd"] = dict(numbers=[46] * 4, positions=fcc + rng.normal(0, 0.1, (4, 3)), cell=np.eye(3) * a,
                          pbc=[True] * 3, nmax=3, lmax=4, rcut=5.0, alpha=2.0, stress=True)
    # 12 Al in two fcc(100) layers + one Au adatom, vacuum along z, periodic in x, y only
    d = 4.05 / np.sqrt(2)
    pos, num = [], []
    for iz in range(2):
        for ix in range(3):
            for iy in range(2):
                pos.append([(ix + 0.5 * iz) * d, (iy + 0.5 * iz) * d, iz * 4.05 / 2]); num.append(13)
    pos.append([1.1 * d, 0.6 * d, 4.05 / 2 + 1.9]); num.append(79)
    out["slab_al12au"] = dict(numbers=num, positions=np.array(pos) + rng.normal(0, 0.05, (13, 3)),
                              cell=np.diag([3 * d, 2 * d, 14.0]), pbc=[True, True, False],
                              nmax=3, lmax=4, rcut=5.0, alpha=2.0, stress=False)
    clu = rng.uniform(0, 3.2, (6, 3))
    out["cluster_mixed"] = dict(numbers=[46, 8, 12, 8, 46, 12], positions=clu, cell=np.zeros((3, 3)),
                                pbc=[False] * 3, nmax=3, lmax=4, rcut=5.0, alpha=2.0, stress=False)
    out["cluster_n2l3"] = dict(numbers=[46, 8, 12, 8, 46, 12], positions=clu, cell=np.zeros((3, 3)),
                               pbc=[False] * 3, nmax=2, lmax=3, rcut=4.0, alpha=1.5, stress=False)
    # nmax = 4, lmax = 4: ncoef = 50 (beyond the 32-column operand plane of the usual configuration)
    out["cluster_n4l4"] = dict(numbers=[46, 8, 12, 8, 46, 12], positions=clu, cell=np.zeros((3, 3)),
                               pbc=[False] * 3, nmax=4, lmax=4, rcut=4.0, alpha=2.0, stress=False)
    out["fcc4_n4l4"] = dict(numbers=[46, 78, 46, 46], positions=fcc + np.random.default_rng(5).normal(0, 0.1, (4, 3)),
                            cell=np.eye(3) * a, pbc=[True] * 3, nmax=4, lmax=4, rcut=5.0, alpha=2.0, stress=False)
    # sheared periodic cell, two species
    cell = np.array([[4.2, 0, 0], [1.3, 3.9, 0], [0.4, 0.7, 4.4]])
    out["triclinic_mgo"] = dict(numbers=[12, 8, 12, 8], positions=rng.uniform(0, 1, (4, 3)) @ cell, cell=cell,
                                pbc=[True] * 3, nmax=3, lmax=4, rcut=4.5, alpha=2.0, stress=False)
    # nmax = 5 and 6 (ncoef = 60 and 63: what the 64-column operand plane leaves beyond nmax = 4)
    out["cluster_n5l3"] = dict(numbers=[46, 8, 12, 8, 46, 12], positions=clu, cell=np.zeros((3, 3)),
                               pbc=[False] * 3, nmax=5, lmax=3, rcut=4.0, alpha=2.0, stress=False)
    out["fcc4_n6l2"] = dict(numbers=[46, 78, 46, 46], positions=fcc + np.random.default_rng(7).normal(0, 0.1, (4, 3)),
                            cell=np.eye(3) * a, pbc=[True] * 3, nmax=6, lmax=2, rcut=5.0, alpha=2.0, stress=False)
    # beyond 64 components (round 4): SO3(4, 6) -> 70, SO3(5, 4) -> 75, SO3(6, 3) -> 84
    out["cluster_n4l6"] = dict(numbers=[46, 8, 12, 8, 46, 12], positions=clu, cell=np.zeros((3, 3)),
                               pbc=[False] * 3, nmax=4, lmax=6, rcut=4.0, alpha=2.0, stress=False)
    out["fcc4_n5l4"] = dict(numbers=[46, 78, 46, 46], positions=fcc + np.random.default_rng(9).normal(0, 0.1, (4, 3)),
                            cell=np.eye(3) * a, pbc=[True] * 3, nmax=5, lmax=4, rcut=5.0, alpha=2.0, stress=False)
    out["cluster_n6l3"] = dict(numbers=[46, 8, 12, 8, 46, 12], positions=clu, cell=np.zeros((3, 3)),
                               pbc=[False] * 3, nmax=6, lmax=3, rcut=4.0, alpha=2.0, stress=False)
    return out


def main():
    warnings.simplefilter("ignore")
    base, dot, so3 = import_reference()
    rng = np.random.default_rng(42)
    # ---- numpy helpers behind kernel.diag
    d = 30
    x1 = rng.random((9, d)); dx1 = rng.standard_normal((9, d, 3))
    ele = np.array([46, 8, 46, 46, 8, 12, 46, 8, 46], dtype=np.int64)
    x1[4] = 0.0                                                  # a zero row: not skipped by these helpers
    mask = base.get_mask(ele, ele)
    payload = dict(x1=x1, dx1=dx1, ele=ele, sigma=2.5, l=1.3, sigma0=2.0, zeta=2.0)
    payload["K_ee_RBF"] = base.K_ee_RBF(x1, x1, 2.5 ** 2, 1.3 ** 2, 2.0, mask=mask)
    payload["K_ee_RBF_nomask"] = base.K_ee_RBF(x1, x1, 2.5 ** 2, 1.3 ** 2, 2.0, mask=None)
    payload["K_ee_dot"] = dot.K_ee(x1, x1, 2.0 ** 2, 2.0 ** 2, 2.0, mask)
    payload["K_ff_dot"] = dot.K_ff(x1, x1, dx1, dx1, 2.0 ** 2, 2.0 ** 2, 2.0, mask)
    payload["K_ee_RBF_z3"] = base.K_ee_RBF(x1, x1, 1.7 ** 2, 0.9 ** 2, 3.0, mask=mask)
    payload["K_ff_dot_z3"] = dot.K_ff(x1, x1, dx1, dx1, 1.7 ** 2, 1.5 ** 2, 3.0, mask)
    if not sys.argv[1:]:
        np.savez_compressed(os.path.join(HERE, "diag_helpers.npz"), **payload)
        print("diag_helpers.npz", payload["K_ee_RBF"], payload["K_ee_dot"], payload["K_ff_dot"][0])
    # ---- SO3
    only = sys.argv[1:]                                          # fixture names: write just these (the others stay as committed)
    for name, s in structures().items():
        if only and name not in only:
            continue
        atoms = SimpleAtoms(s["numbers"], s["positions"], s["cell"], s["pbc"])
        des = so3.SO3(nmax=s["nmax"], lmax=s["lmax"], rcut=s["rcut"], alpha=s["alpha"], derivative=True,
                      stress=s["stress"])
        out = des.calculate(atoms)
        payload = dict(numbers=np.asarray(s["numbers"]), positions=atoms.positions, cell=atoms.cell,
                       pbc=np.asarray(s["pbc"]), nmax=s["nmax"], lmax=s["lmax"], rcut=s["rcut"], alpha=s["alpha"],
                       x=out["x"], dxdr=out["dxdr"], seq=out["seq"])
        if s["stress"]:
            payload["rdxdr"] = out["rdxdr"]
        path = os.path.join(HERE, f"so3_{name}.npz")
        np.savez_compressed(path, **payload)
        print(f"so3_{name}.npz: x{out['x'].shape} dxdr{out['dxdr'].shape} seq{out['seq'].shape} "
              f"{os.path.getsize(path)/1024:.0f} KiB  |x|max={np.abs(out['x']).max():.3e}")


if __name__ == "__main__":
    main()
