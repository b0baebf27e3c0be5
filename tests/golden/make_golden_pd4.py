"""Decode a subset of the reference's shipped training database examples/database/pd4-RBF.db (ASE sqlite
format, read with the standard library only -- ASE is not installed here) into tests/golden/pd4_subset.npz,
and attach golden descriptor values for its first structure computed by the reference's own SO3.calculate
(imported exactly as in make_golden_python.py).  Data only: structures, labels, hyper-parameters of
examples/database/pd4-RBF.json, and reference outputs.

    python tests/golden/make_golden_pd4.py
"""
import json
import os
import sqlite3
import struct
import sys
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.abspath(os.path.join(HERE, "..", "..")))
sys.path.insert(0, HERE)
from make_golden_python import import_reference          # noqa: E402
from gpr_calculator_amd.neighbors import SimpleAtoms       # noqa: E402

DB = "/root/reference/examples/database/pd4-RBF.db"
JS = "/root/reference/examples/database/pd4-RBF.json"
N_STRUCT = 8


def decode_data(blob):
    off = struct.unpack("<q", blob[:8])[0]
    meta = json.loads(blob[off:].decode())
    out = {}
    for k, v in meta.items():
        if isinstance(v, dict) and "__ndarray__" in v:
            shape, dtype, o = v["__ndarray__"]
            cnt = int(np.prod(shape))
            out[k] = np.frombuffer(blob, dtype=dtype, count=cnt, offset=o).reshape(shape).copy()
        else:
            out[k] = v
    return out


def main():
    warnings.simplefilter("ignore")
    db = sqlite3.connect(f"file:{DB}?mode=ro", uri=True)
    rows = db.execute("select numbers, positions, cell, pbc, data, natoms from systems order by id limit ?", (N_STRUCT,)).fetchall()
    payload = {"n_struct": len(rows)}
    for k, (numbers, positions, cell, pbc, data, natoms) in enumerate(rows):
        d = decode_data(data)
        payload[f"numbers_{k}"] = np.frombuffer(numbers, dtype=np.int32).copy()
        payload[f"positions_{k}"] = np.frombuffer(positions, dtype=np.float64).reshape(natoms, 3).copy()
        payload[f"cell_{k}"] = np.frombuffer(cell, dtype=np.float64).reshape(3, 3).copy()
        payload[f"pbc_{k}"] = np.array([(pbc >> i) & 1 for i in range(3)], dtype=bool)
        payload[f"energy_{k}"] = float(d["energy"])
        payload[f"force_{k}"] = d["force"]
        payload[f"energy_in_{k}"] = bool(d["energy_in"])
        payload[f"force_in_{k}"] = np.asarray(d["force_in"], dtype=np.int64)
    js = json.load(open(JS))
    payload.update(sigma=js["kernel"]["sigma"], l=js["kernel"]["l"], zeta=js["kernel"]["zeta"],
                   noise_e=js["noise"]["energy"], noise_f=js["noise"]["force"],
                   nmax=js["descriptor"]["nmax"], lmax=js["descriptor"]["lmax"], rcut=js["descriptor"]["rcut"],
                   alpha=js["descriptor"]["alpha"], json_energy_mae=js["error"]["energy_mae"],
                   json_forces_mae=js["error"]["forces_mae"])
    # golden descriptor of structure 0 from the reference's own SO3
    _, _, so3 = import_reference()
    atoms = SimpleAtoms(payload["numbers_0"], payload["positions_0"], payload["cell_0"], payload["pbc_0"])
    des = so3.SO3(nmax=3, lmax=4, rcut=5.0, alpha=2.0, derivative=True, stress=False)
    out = des.calculate(atoms)
    sel = np.arange(0, len(out["seq"]), 37)
    payload.update(ref_x=out["x"], ref_seq=out["seq"].astype(np.int16), ref_rows=sel, ref_dxdr_rows=out["dxdr"][sel])
    path = os.path.join(HERE, "pd4_subset.npz")
    np.savez_compressed(path, **payload)
    print(path, os.path.getsize(path) // 1024, "KiB; seq rows", len(out["seq"]), "force_in", [len(payload[f"force_in_{k}"]) for k in range(len(rows))])


if __name__ == "__main__":
    main()
