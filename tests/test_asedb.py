"""ASE sqlite training databases through the standard-library reader / writer (gpr_calculator_amd/asedb.py): the record
encoding, a write -> read round trip, and (where the reference is mounted) the shipped pd4-RBF.db decoded to the labels
the golden fixture holds."""
import os
import sqlite3

import numpy as np
import pytest

from gpr_calculator_amd import asedb
from gpr_calculator_amd.neighbors import SimpleAtoms, atoms_arrays

HERE = os.path.dirname(os.path.abspath(__file__))
REF_DB = "/root/reference/examples/database/pd4-RBF.db"


def test_record_round_trip_and_layout():
    rng = np.random.default_rng(0)
    rec = {"energy": -12.5, "force": rng.normal(size=(7, 3)), "energy_in": True, "force_in": [0, 3, 6],
           "nested": {"ids": np.arange(5, dtype=np.int32), "flag": np.bool_(False)}}
    blob = asedb.encode_record(rec)
    # layout of ase.db's binary records: offset word, raw arrays from byte 8 on, compact JSON at the offset
    off = int(np.frombuffer(blob[:8], dtype="<i8")[0])
    assert blob[off:off + 1] == b"{" and np.array_equal(np.frombuffer(blob, "<f8", 21, 8).reshape(7, 3), rec["force"])
    out = asedb.decode_record(blob)
    assert out["energy"] == -12.5 and out["energy_in"] is True and out["force_in"] == [0, 3, 6]
    assert np.array_equal(out["force"], rec["force"]) and out["force"].dtype == np.float64
    assert np.array_equal(out["nested"]["ids"], np.arange(5)) and out["nested"]["ids"].dtype == np.int32
    assert out["nested"]["flag"] is False
    assert asedb.decode_record('{"a": 1, "b": {"__ndarray__": [[2], "float64", [1.0, 2.0]]}}')["b"].tolist() == [1.0, 2.0]
    assert asedb.decode_record(None) == {}


def test_db_round_trip(tmp_path):
    rng = np.random.default_rng(1)
    path = str(tmp_path / "train.db")
    recs = []
    for k in range(3):
        n = 4 + k
        atoms = SimpleAtoms(numbers=rng.choice([8, 46], n), positions=rng.uniform(0, 5, (n, 3)),
                            cell=np.diag([5.0, 6.0, 7.0]) + 0.1 * k, pbc=[True, k % 2 == 0, False])
        recs.append((atoms, {"energy": float(-k), "force": rng.normal(size=(n, 3)), "energy_in": k != 1,
                             "force_in": list(range(0, n, 2))}, {"dft_energy": -k / n, "dft_fmax": 0.5 + k}))
    asedb.write_db(path, recs[:2])
    asedb.write_db(path, recs[2:], append=True)
    rows = asedb.read_db(path)
    assert [r["id"] for r in rows] == [1, 2, 3] and len(asedb.read_db(path, limit=2)) == 2
    for (atoms, data, kvp), row in zip(recs, rows):
        for a, b in zip(atoms_arrays(atoms), atoms_arrays(row["atoms"])):
            assert np.array_equal(a, b)
        assert row["data"]["energy"] == data["energy"] and np.array_equal(row["data"]["force"], data["force"])
        assert row["data"]["energy_in"] == data["energy_in"] and row["data"]["force_in"] == data["force_in"]
        assert row["key_value_pairs"] == kvp
    # the side tables ase.db queries
    con = sqlite3.connect(path)
    assert con.execute("select value from information where name='version'").fetchone() == ("9",)
    assert con.execute("select count(*) from species").fetchone()[0] == sum(len(set(r[0].numbers.tolist())) for r in recs)
    assert con.execute("select count(*) from number_key_values").fetchone()[0] == 6
    assert con.execute("select natoms, pbc from systems where id=2").fetchone() == (5, 1)
    con.close()
    asedb.write_db(path, recs[:1])                  # "w": the file is reset
    assert len(asedb.read_db(path)) == 1
    with pytest.raises(FileNotFoundError):
        asedb.read_db(str(tmp_path / "missing.db"))


@pytest.mark.skipif(not os.path.exists(REF_DB), reason="reference data not mounted")
def test_reads_the_reference_database_like_the_golden_subset():
    """the shipped 206-structure Pd4/MgO database: schema identical to the one written here, labels equal to the
    decoded subset under tests/golden (made with an independent decoder)"""
    gold = np.load(os.path.join(HERE, "golden", "pd4_subset.npz"), allow_pickle=False)
    n = int(gold["n_struct"])
    rows = asedb.read_db(REF_DB, limit=n)
    assert len(rows) == n and len(asedb.read_db(REF_DB)) == 206
    for k, row in enumerate(rows):
        numbers, pos, cell, pbc = atoms_arrays(row["atoms"])
        assert np.array_equal(numbers, gold[f"numbers_{k}"]) and np.array_equal(pos, gold[f"positions_{k}"])
        assert np.array_equal(cell, gold[f"cell_{k}"]) and np.array_equal(pbc, gold[f"pbc_{k}"])
        assert row["data"]["energy"] == float(gold[f"energy_{k}"]) and np.array_equal(row["data"]["force"], gold[f"force_{k}"])
        assert bool(row["data"]["energy_in"]) == bool(gold[f"energy_in_{k}"])
        assert np.array_equal(np.asarray(row["data"]["force_in"]), gold[f"force_in_{k}"])
    ref = sqlite3.connect(f"file:{REF_DB}?mode=ro", uri=True)
    cols = lambda con, t: [(c[1], c[2]) for c in con.execute(f"pragma table_info({t})")]
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        mine_path = os.path.join(tmp, "x.db")
        asedb.write_db(mine_path, [(rows[0]["atoms"], rows[0]["data"], rows[0]["key_value_pairs"])])
        mine = sqlite3.connect(mine_path)
        for t in ("systems", "species", "keys", "text_key_values", "number_key_values", "information"):
            assert cols(mine, t) == cols(ref, t), t
        names = lambda con: sorted(r[0] for r in con.execute("select name from sqlite_master where type='index' and sql is not null"))
        assert names(mine) == names(ref)
        # the same row written here carries the same blobs as the reference's file
        q = "select numbers, positions, cell, pbc, natoms, key_value_pairs, data, volume from systems where id=1"
        a, b = mine.execute(q).fetchone(), ref.execute(q).fetchone()
        assert a[:5] == b[:5] and a[5] == b[5] and bytes(a[6]) == bytes(b[6]) and abs(a[7] - b[7]) < 1e-9
        mine.close()
    ref.close()
