"""Training-database files in the ASE sqlite format (db version 9), read and written with the standard library.

The reference stores its training set with ase.db (gaussianprocess.py:689-724 writes, :726-753 reads): one row per
structure, the labels in the row's `data` record (`energy`, `force[n,3]`, `energy_in`, `force_in`) and
`dft_energy` / `dft_fmax` as key-value pairs.  This module speaks that file format directly -- the `systems` row
(numbers int32, positions / cell float64 as raw little-endian blobs, pbc as a 3-bit mask), the side tables ase.db
keeps for queries (`species`, `keys`, `number_key_values`, `text_key_values`, `information`) and the `data` record
(8-byte offset, raw arrays, compact JSON with `{"__ndarray__": [shape, dtype, offset]}` references) -- so that
models saved by the reference (e.g. examples/database/pd4-RBF.json + .db) load here and files written here open in
ase.db, whether or not ASE is installed.  Host-side I/O only; nothing here is on the prediction / fit path.
"""
import json
import os
import sqlite3
import struct
import time

import numpy as np

from .neighbors import SimpleAtoms, atoms_arrays

VERSION = 9
_YEAR = 31557600.0               # ase.db time unit: Julian years since 2000-01-01
_T2000 = 946681200.0

_SCHEMA = [
    """CREATE TABLE systems (
    id INTEGER PRIMARY KEY AUTOINCREMENT,
    unique_id TEXT UNIQUE, ctime REAL, mtime REAL, username TEXT,
    numbers BLOB, positions BLOB, cell BLOB, pbc INTEGER,
    initial_magmoms BLOB, initial_charges BLOB, masses BLOB, tags BLOB, momenta BLOB,
    constraints TEXT, calculator TEXT, calculator_parameters TEXT,
    energy REAL, free_energy REAL, forces BLOB, stress BLOB, dipole BLOB, magmoms BLOB, magmom REAL, charges BLOB,
    key_value_pairs TEXT, data BLOB,
    natoms INTEGER, fmax REAL, smax REAL, volume REAL, mass REAL, charge REAL)""",
    "CREATE TABLE species (Z INTEGER, n INTEGER, id INTEGER, FOREIGN KEY (id) REFERENCES systems(id))",
    "CREATE TABLE keys (key TEXT, id INTEGER, FOREIGN KEY (id) REFERENCES systems(id))",
    "CREATE TABLE text_key_values (key TEXT, value TEXT, id INTEGER, FOREIGN KEY (id) REFERENCES systems(id))",
    "CREATE TABLE number_key_values (key TEXT, value REAL, id INTEGER, FOREIGN KEY (id) REFERENCES systems(id))",
    "CREATE TABLE information (name TEXT, value TEXT)",
    "CREATE INDEX unique_id_index ON systems(unique_id)",
    "CREATE INDEX ctime_index ON systems(ctime)",
    "CREATE INDEX username_index ON systems(username)",
    "CREATE INDEX calculator_index ON systems(calculator)",
    "CREATE INDEX species_index ON species(Z)",
    "CREATE INDEX key_index ON keys(key)",
    "CREATE INDEX text_index ON text_key_values(key)",
    "CREATE INDEX number_index ON number_key_values(key)",
]


# ---------------------------------------------------------------------- the `data` record
def encode_record(obj):
    """dict of scalars / lists / numpy arrays -> bytes: [int64 offset of the JSON][arrays, 8-byte aligned][JSON]"""
    chunks, pos = [], 8

    def conv(v):
        nonlocal pos
        if isinstance(v, np.ndarray):
            a = np.ascontiguousarray(v)
            if a.dtype == bool or a.dtype == object:
                return conv(a.tolist())
            if a.dtype.byteorder == ">":
                a = a.astype(a.dtype.newbyteorder("<"))
            pad = (-pos) % 8
            if pad:
                chunks.append(b"\0" * pad)
                pos += pad
            ref = {"__ndarray__": [list(a.shape), a.dtype.name, pos]}
            chunks.append(a.tobytes())
            pos += a.nbytes
            return ref
        if isinstance(v, dict):
            return {str(k): conv(x) for k, x in v.items()}
        if isinstance(v, (list, tuple)):
            return [conv(x) for x in v]
        if isinstance(v, np.generic):
            return v.item()
        return v

    meta = conv(obj)
    body = b"".join(chunks)
    return struct.pack("<q", 8 + len(body)) + body + json.dumps(meta, separators=(",", ":")).encode()


def decode_record(blob):
    """inverse of encode_record; also accepts the plain-JSON text older ase.db versions stored"""
    if blob is None:
        return {}
    if isinstance(blob, str):
        return _plain_json(json.loads(blob))
    blob = bytes(blob)
    if blob[:1] in (b"{", b"["):
        return _plain_json(json.loads(blob.decode()))
    off = struct.unpack("<q", blob[:8])[0]

    def conv(v):
        if isinstance(v, dict):
            if "__ndarray__" in v and len(v) == 1:
                shape, dtype, o = v["__ndarray__"]
                count = int(np.prod(shape)) if len(shape) else 1
                return np.frombuffer(blob, dtype=np.dtype(dtype), count=count, offset=int(o)).reshape(shape).copy()
            return {k: conv(x) for k, x in v.items()}
        if isinstance(v, list):
            return [conv(x) for x in v]
        return v

    return conv(json.loads(blob[off:].decode()))


def _plain_json(v):
    if isinstance(v, dict):
        if "__ndarray__" in v and len(v) == 1:          # inline form: [shape, dtype, flat list]
            shape, dtype, flat = v["__ndarray__"]
            return np.array(flat, dtype=dtype).reshape(shape)
        return {k: _plain_json(x) for k, x in v.items()}
    if isinstance(v, list):
        return [_plain_json(x) for x in v]
    return v


# ---------------------------------------------------------------------- rows
def _fixed_from_constraints(text):
    """indices of a FixAtoms entry in the row's constraints JSON (gaussianprocess.py:823-832 honours only those)"""
    if not text:
        return ()
    for c in json.loads(text):
        if c.get("name") == "FixAtoms":
            kw = c.get("kwargs", {})
            idx = kw.get("indices", kw.get("mask", ()))
            if isinstance(idx, dict):
                idx = _plain_json(idx)
            idx = np.asarray(idx)
            return np.flatnonzero(idx) if idx.dtype == bool else idx.astype(np.int64)
    return ()


def _make_atoms(numbers, positions, cell, pbc, fixed):
    try:
        from ase import Atoms
        from ase.constraints import FixAtoms
    except ImportError:
        return SimpleAtoms(numbers, positions, cell, pbc, fixed=fixed)
    atoms = Atoms(numbers=numbers, positions=positions, cell=cell, pbc=pbc)
    if len(fixed):
        atoms.set_constraint(FixAtoms(indices=list(fixed)))
    return atoms


def read_db(filename, limit=None):
    """rows of an ASE sqlite file in id order: list of dicts {id, atoms, data, key_value_pairs}; `atoms` is an
    ase.Atoms when ASE is importable, else a SimpleAtoms"""
    if not os.path.exists(filename):
        raise FileNotFoundError(filename)
    con = sqlite3.connect(f"file:{filename}?mode=ro", uri=True)
    try:
        q = "SELECT id, numbers, positions, cell, pbc, constraints, key_value_pairs, data, natoms FROM systems ORDER BY id"
        if limit is not None:
            q += f" LIMIT {int(limit)}"
        out = []
        for rid, numbers, positions, cell, pbc, cons, kvp, data, natoms in con.execute(q):
            n = int(natoms)
            num = np.frombuffer(numbers, dtype="<i4").astype(np.int64) if numbers is not None else np.zeros(0, np.int64)
            pos = np.frombuffer(positions, dtype="<f8").reshape(n, 3).copy() if positions is not None else np.zeros((0, 3))
            cel = np.frombuffer(cell, dtype="<f8").reshape(3, 3).copy() if cell is not None else np.zeros((3, 3))
            pb = np.array([(int(pbc or 0) >> k) & 1 for k in range(3)], dtype=bool)
            out.append({"id": rid, "atoms": _make_atoms(num, pos, cel, pb, _fixed_from_constraints(cons)),
                        "data": decode_record(data), "key_value_pairs": json.loads(kvp) if kvp else {}})
        return out
    finally:
        con.close()


def write_db(filename, records, append=False):
    """records: iterable of (atoms, data dict, key_value_pairs dict).  Creates the file (or appends to it) with the
    tables and indices ase.db expects; constraints are not stored (the reference clears them before writing,
    gaussianprocess.py:722)."""
    if not append and os.path.exists(filename):
        os.remove(filename)
    fresh = not os.path.exists(filename)
    con = sqlite3.connect(filename)
    try:
        cur = con.cursor()
        if fresh:
            for stmt in _SCHEMA:
                cur.execute(stmt)
            cur.execute("INSERT INTO information VALUES ('version', ?)", (str(VERSION),))
        user = os.environ.get("USER") or os.environ.get("USERNAME") or "gpr"
        for atoms, data, kvp in records:
            numbers, pos, cell, pbc = atoms_arrays(atoms)
            now = (time.time() - _T2000) / _YEAR
            kvp = {k: (v.item() if isinstance(v, np.generic) else v) for k, v in (kvp or {}).items()}
            mask = int(sum(int(bool(b)) << k for k, b in enumerate(pbc)))
            cur.execute(
                "INSERT INTO systems (unique_id, ctime, mtime, username, numbers, positions, cell, pbc, key_value_pairs, data,"
                " natoms, volume, charge) VALUES (?,?,?,?,?,?,?,?,?,?,?,?,?)",
                (os.urandom(16).hex(), now, now, user, numbers.astype("<i4").tobytes(),
                 np.ascontiguousarray(pos, dtype="<f8").tobytes(), np.ascontiguousarray(cell, dtype="<f8").tobytes(), mask,
                 json.dumps(kvp, separators=(", ", ": ")), encode_record(data or {}), len(numbers),
                 float(abs(np.linalg.det(cell))), 0.0))
            rid = cur.lastrowid
            zs, counts = np.unique(numbers, return_counts=True)
            cur.executemany("INSERT INTO species VALUES (?,?,?)", [(int(z), int(c), rid) for z, c in zip(zs, counts)])
            cur.executemany("INSERT INTO keys VALUES (?,?)", [(k, rid) for k in kvp])
            cur.executemany("INSERT INTO number_key_values VALUES (?,?,?)",
                            [(k, float(v), rid) for k, v in kvp.items() if isinstance(v, (int, float, bool))])
            cur.executemany("INSERT INTO text_key_values VALUES (?,?,?)",
                            [(k, v, rid) for k, v in kvp.items() if isinstance(v, str)])
        con.commit()
    finally:
        con.close()


def read_traj(path):
    """Frames of an ASE trajectory file (.traj, the "ULM" container: 48-byte header, one JSON record per frame with
    ndarray payloads referenced by file offset) without ASE: a list of dictionaries
    {atoms: SimpleAtoms (FixAtoms indices in .fixed), energy, forces} -- energy / forces None when the frame carries
    no calculator results.  Header fields of the first frame (numbers, cell, pbc, constraints) apply to later frames
    that omit them, as in ase.io.trajectory."""
    import struct
    from .neighbors import SimpleAtoms
    with open(path, "rb") as fh:
        b = fh.read()
    if b[:8] != b"- of Ulm":
        raise ValueError(f"{path}: not an ASE trajectory (ULM) file")
    _, nitems, pos0 = struct.unpack("<3q", b[24:48])
    offsets = np.frombuffer(b, dtype="<i8", count=nitems, offset=pos0)

    def arr(spec):
        shape, dtype, off = spec["ndarray"]
        return np.frombuffer(b, dtype=dtype, count=int(np.prod(shape)), offset=off).reshape(shape).copy()

    def field(meta, key, default=None):
        if key + "." in meta:
            return arr(meta[key + "."])
        return np.asarray(meta[key]) if key in meta else default
    frames, head = [], {}
    for off in offsets:
        off = int(off)
        n = struct.unpack("<q", b[off:off + 8])[0]
        meta = json.loads(b[off + 8:off + 8 + n].decode())
        for key in ("numbers", "cell", "pbc"):
            v = field(meta, key)
            if v is not None:
                head[key] = v
        if "constraints" in meta:
            cons = json.loads(meta["constraints"]) if isinstance(meta["constraints"], str) else meta["constraints"]
            head["fixed"] = [i for c in cons if c.get("name") == "FixAtoms" for i in c["kwargs"]["indices"]]
        atoms = SimpleAtoms(head["numbers"], field(meta, "positions"), head.get("cell"), head.get("pbc", False),
                            fixed=head.get("fixed", ()))
        calc = meta.get("calculator.", meta.get("calculator", {})) or {}
        frames.append({"atoms": atoms, "energy": calc.get("energy"), "forces": field(calc, "forces")})
    return frames
