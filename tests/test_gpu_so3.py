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
    if "rdxdr" in z.files:      # stress rows of the reference's own SO3.calculate(stress=True)
        des.stress = True
        out = des.calculate(atoms)
        assert rel(out["x"], z["x"]) < REL and rel(out["dxdr"], z["dxdr"]) < REL
        assert out["rdxdr"].shape == z["rdxdr"].shape and rel(out["rdxdr"], z["rdxdr"]) < REL


def test_so3_batch_matches_oracle_on_bench_cells():
    from gpr_calculator_amd.SO3 import SO3
    from gpr_calculator_amd.synthetic import fcc_cell
    from oracle.so3_numpy import so3_calculate
    des = SO3(nmax=3, lmax=4, rcut=5.0, alpha=2.0)
    strucs = [fcc_cell(seed=s, species=(46, 78) if s % 2 else (46,)) for s in range(3)]
    out = des.calculate_batch_dev(strucs, host_index=True)
    x, dxdr = out["x"].cpu().numpy(), out["dxdr"].cpu().numpy()
    for k, s in enumerate(strucs):
        ref = so3_calculate(s.numbers, s.positions, s.cell, s.pbc, 3, 4, 5.0, 2.0)
        a0, a1, r0, r1 = out["atom0"][k], out["atom0"][k + 1], out["row0"][k], out["row0"][k + 1]
        assert np.array_equal(out["seq"][k], ref["seq"])
        assert rel(x[a0:a1], ref["x"]) < REL and rel(dxdr[r0:r1], ref["dxdr"]) < REL


def _mixed_batch():
    from gpr_calculator_amd.synthetic import fcc_cell
    from gpr_calculator_amd.neighbors import SimpleAtoms
    strucs = [fcc_cell(seed=s, species=(46, 78) if s % 2 else (46,)) for s in range(5)]
    rng = np.random.default_rng(3)
    # a cluster without periodicity, a slab (pbc in two directions), a small skewed cell that needs 2+ image shells
    strucs.append(SimpleAtoms(numbers=[46, 46, 8, 1, 1], positions=rng.normal(size=(5, 3)) * 1.5,
                              cell=np.eye(3) * 20.0, pbc=[False, False, False]))
    strucs.append(SimpleAtoms(numbers=[78] * 6, positions=rng.uniform(0, 5.5, size=(6, 3)),
                              cell=[[5.5, 0, 0], [0.7, 5.5, 0], [0, 0, 14.0]], pbc=[True, True, False]))
    strucs.append(SimpleAtoms(numbers=[46, 78], positions=[[0.1, 0.2, 0.3], [1.2, 1.1, 1.4]],
                              cell=[[2.6, 0, 0], [0.9, 2.5, 0], [0.3, 0.4, 2.7]], pbc=[True, True, True]))
    return strucs


@pytest.mark.parametrize("weight_on", [False, True], ids=["plain", "weight_on"])
def test_device_neighbour_build_is_identical_to_host_build(weight_on):
    """gpr_nbr_build_device against gpr_nbr_build: same pairs, rows and index maps, bit for bit (weight_on: neighbours
    of another species carry -Z, SO3.py:381-385)"""
    import torch
    from gpr_calculator_amd.SO3 import SO3
    des = SO3(nmax=3, lmax=4, rcut=5.0, alpha=2.0, weight_on=weight_on)
    strucs = _mixed_batch()
    host = des.neighbor_index_batch(strucs)
    nbr = des.neighbor_batch_dev(strucs)
    devb = nbr.export()
    assert nbr.P == len(host["w"]) and nbr.S == len(host["seq"])
    assert (host["w"] < 0).any() == weight_on
    for k in ("vec", "w", "centre_pair0", "centre_row0", "row_pair0", "self_row", "seq", "row0"):
        assert np.array_equal(np.asarray(devb[k], dtype=np.float64), np.asarray(host[k], dtype=np.float64)), k
    # force rows: rows with seq[:, 1] == a, ascending in the centre, for a subset of moving atoms
    atom0, row0, seq = host["atom0"], host["row0"], host["seq"]
    row_off = np.repeat(atom0[:-1], np.diff(row0))
    gi, gj = seq[:, 0] + row_off, seq[:, 1] + row_off
    atoms = np.array([a for a in range(int(atom0[-1])) if a % 3 != 1])
    counts, row_index, x_index, _ = nbr.force_rows(atoms)
    nbr.check()
    row_index, x_index = row_index.cpu().numpy(), x_index.cpu().numpy()
    off = 0
    for a, c in zip(atoms, counts):
        rows = np.flatnonzero(gj == a)
        assert c == len(rows)
        assert np.array_equal(row_index[off:off + c], rows)
        assert np.array_equal(x_index[off:off + c], gi[rows])
        off += c
    assert off == len(row_index)
    # every atom moving: no index upload, same answer as the explicit list
    c_all, r_all, x_all, _ = nbr.force_rows(None)
    c_exp, r_exp, x_exp, _ = nbr.force_rows(np.arange(int(atom0[-1])))
    assert np.array_equal(c_all, c_exp) and torch.equal(r_all, r_exp) and torch.equal(x_all, x_exp)


def test_batch_descriptor_device_index_matches_host_index():
    import torch
    from gpr_calculator_amd.SO3 import SO3
    des = SO3(nmax=3, lmax=4, rcut=5.0, alpha=2.0)
    strucs = _mixed_batch()
    a = des.calculate_batch_dev(strucs, host_index=True)
    b = des.calculate_batch_dev(strucs)
    assert torch.equal(a["x"], b["x"]) and torch.equal(a["dxdr"], b["dxdr"])


@pytest.mark.parametrize("cfg", [(3, 4, 5.0, 2.0), (3, 3, 3.5, 1.0), (2, 6, 6.0, 3.0)], ids=["n3l4", "n3l3", "n2l6"])
def test_radial_chebyshev_table_matches_direct_quadrature(cfg):
    """the piecewise Chebyshev table of F0_nl, F1_nl against the direct 10(nmax+lmax+1)-node quadrature on the
    device, and the direct quadrature against scipy's Bessel functions (oracle formula, SO3.py:608-727)"""
    from scipy.special import spherical_in
    from gpr_calculator_amd.SO3 import SO3, radial_table
    nmax, lmax, rcut, alpha = cfg
    des = SO3(nmax=nmax, lmax=lmax, rcut=rcut, alpha=alpha)
    r = np.random.default_rng(5).uniform(0.02, rcut * (1 - 1e-9), 3000)
    tab = des.radial_functions(r, mode="table")
    direct = des.radial_functions(r, mode="direct")
    scale = np.abs(direct).max(axis=0, keepdims=True)
    assert (np.abs(tab - direct) / scale).max() < 2e-13
    rq, G = radial_table(nmax, lmax, rcut, alpha)
    arg = 2 * alpha * np.outer(r, rq)
    E = 4 * np.pi * np.exp(-alpha * r ** 2)
    for l in range(lmax + 1):
        I = spherical_in(l, arg) @ G.T                                            # [P, n]
        dI = (spherical_in(l, arg, derivative=True) * (2 * alpha * rq)[None, :]) @ G.T
        F0 = E[:, None] * I
        F1 = E[:, None] * (dI - 2 * alpha * r[:, None] * I)
        assert (np.abs(direct[:, :, l, 0] - F0) / scale[0, :, l, 0]).max() < 1e-12
        assert (np.abs(direct[:, :, l, 1] - F1) / scale[0, :, l, 1]).max() < 1e-11


@pytest.mark.parametrize("scale", [1.0, 0.78], ids=["2chunks", "3chunks"])
def test_so3_dense_small_cell_matches_oracle(scale):
    """a 2-atom skewed cell at rcut = 5: 56 / ~120 neighbour pairs per centre and ~30 / ~60 periodic images per seq
    row, so the kernel's pair tables are rebuilt in several chunks of 48 and rows straddle chunk boundaries"""
    from gpr_calculator_amd.SO3 import SO3
    from gpr_calculator_amd.neighbors import SimpleAtoms
    from oracle.so3_numpy import so3_calculate
    s = SimpleAtoms(numbers=[46, 78], positions=np.array([[0.1, 0.2, 0.3], [1.2, 1.1, 1.4]]) * scale,
                    cell=np.array([[2.6, 0, 0], [0.9, 2.5, 0], [0.3, 0.4, 2.7]]) * scale, pbc=[True, True, True])
    des = SO3(nmax=3, lmax=4, rcut=5.0, alpha=2.0)
    nb = des.neighbor_index_batch([s])
    assert np.diff(nb["centre_pair0"]).min() > (48 if scale == 1.0 else 96) and np.diff(nb["row_pair0"]).max() > 20
    ref = so3_calculate(s.numbers, s.positions, s.cell, s.pbc, 3, 4, 5.0, 2.0, stress=True)
    # without stress rows: pass 2 by (pair, l) lanes, rows summed over their ~30 / ~60 pairs round by round in the LDS
    # row buffer, rows continued across chunk boundaries through global memory
    out = des.calculate(s)
    assert np.array_equal(out["seq"], ref["seq"])
    assert rel(out["x"], ref["x"]) < REL and rel(out["dxdr"], ref["dxdr"]) < REL
    des.stress = True           # the centre's own periodic images enter the (i, i) stress row only
    out = des.calculate(s)
    assert np.array_equal(out["seq"], ref["seq"])
    assert rel(out["x"], ref["x"]) < REL and rel(out["dxdr"], ref["dxdr"]) < REL
    assert rel(out["rdxdr"], ref["rdxdr"]) < REL


@pytest.mark.parametrize("cfg", [(2, 6, 4.2, 1.5), (4, 2, 4.0, 2.0), (1, 0, 3.0, 1.0), (3, 4, 6.5, 2.0)],
                         ids=["n2l6", "n4l2", "n1l0", "n3l4_rcut6.5"])
def test_so3_other_configurations_match_oracle(cfg):
    """descriptor sizes off the default: lmax = 6 and nmax = 4 take the generically sized kernel instance (l loops to 7,
    four radial channels), nmax = 1 / lmax = 0 the smallest tables, rcut = 6.5 gives ~90 pairs per centre (three chunks,
    more Chebyshev segments); with and without stress rows"""
    from gpr_calculator_amd.SO3 import SO3
    from gpr_calculator_amd.synthetic import fcc_cell
    from oracle.so3_numpy import so3_calculate
    nmax, lmax, rcut, alpha = cfg
    s = fcc_cell(seed=4, reps=1 if rcut < 6 else 2, species=(46, 78), sigma=0.1)
    for stress in (False, True):
        des = SO3(nmax=nmax, lmax=lmax, rcut=rcut, alpha=alpha)
        des.stress = stress
        out = des.calculate(s)
        ref = so3_calculate(s.numbers, s.positions, s.cell, s.pbc, nmax, lmax, rcut, alpha, stress=stress)
        assert np.array_equal(out["seq"], ref["seq"])
        assert rel(out["x"], ref["x"]) < REL and rel(out["dxdr"], ref["dxdr"]) < REL, (stress, rel(out["x"], ref["x"]), rel(out["dxdr"], ref["dxdr"]))
        if stress:
            assert rel(out["rdxdr"], ref["rdxdr"]) < REL


@pytest.mark.parametrize("cfg", [(4, 4, 4.0, 2.0), (3, 6, 4.2, 1.5), (4, 5, 4.0, 2.0), (4, 3, 5.0, 2.0), (5, 3, 4.0, 2.0),
                                 (6, 2, 5.0, 2.0), (5, 1, 5.0, 1.5), (6, 0, 4.0, 2.0), (4, 6, 5.0, 2.0), (5, 4, 5.0, 2.0),
                                 (6, 3, 4.0, 2.0)],
                         ids=["n4l4_d50", "n3l6_d42", "n4l5_d60", "n4l3_d40", "n5l3_d60", "n6l2_d63", "n5l1_d30", "n6l0_d21",
                              "n4l6_d70", "n5l4_d75", "n6l3_d84"])
def test_so3_long_descriptors_match_oracle(cfg):
    """more than 32 power-spectrum components (up to 64) and more than 16 radial-angular channels (up to 28): the wider
    record layout of the general kernel instance, periodic cell; nmax = 5 and 6 (lmax <= 3) take the third instance.  The
    reference's own SO3 pins (4, 4), (5, 3) and (6, 2) through tests/golden/so3_*_n4l4 / _n5l3 / _n6l2.npz"""
    from gpr_calculator_amd.SO3 import SO3
    from gpr_calculator_amd.synthetic import fcc_cell
    from oracle.so3_numpy import so3_calculate
    nmax, lmax, rcut, alpha = cfg
    s = fcc_cell(seed=4, reps=1 if rcut < 5 else 2, species=(46, 78), sigma=0.1)
    out = SO3(nmax=nmax, lmax=lmax, rcut=rcut, alpha=alpha).calculate(s)
    ref = so3_calculate(s.numbers, s.positions, s.cell, s.pbc, nmax, lmax, rcut, alpha)
    assert out["x"].shape[1] == nmax * (nmax + 1) // 2 * (lmax + 1) and np.array_equal(out["seq"], ref["seq"])
    assert rel(out["x"], ref["x"]) < REL and rel(out["dxdr"], ref["dxdr"]) < REL, (rel(out["x"], ref["x"]), rel(out["dxdr"], ref["dxdr"]))


def test_atom_ids_subset_and_derivative_false_follow_the_reference_layout():
    """SO3.calculate(atoms, atom_ids=...) and SO3(derivative=False) (gpr_calc/SO3.py:186-323, 353-399): x keeps its
    [n, ncoef] shape with the rows of the other centres zero, seq / dxdr list only the rows of the chosen centres in
    the order given; without derivatives the dictionary has no seq and dxdr is None"""
    from gpr_calculator_amd.SO3 import SO3
    from gpr_calculator_amd.synthetic import fcc_cell
    a = fcc_cell(seed=4, reps=1, species=(46, 78), sigma=0.1)
    des = SO3(nmax=3, lmax=4, rcut=5.0)
    full = des.calculate(a)
    sub = des.calculate(a, atom_ids=[2, 0])
    assert sub["x"].shape == full["x"].shape and np.array_equal(sub["x"][[0, 2]], full["x"][[0, 2]])
    assert not sub["x"][[1, 3]].any()
    rows = np.concatenate([np.flatnonzero(full["seq"][:, 0] == i) for i in (2, 0)])
    assert np.array_equal(sub["seq"], full["seq"][rows]) and np.array_equal(sub["dxdr"], full["dxdr"][rows])
    nod = SO3(nmax=3, lmax=4, rcut=5.0, derivative=False).calculate(a)
    assert nod["dxdr"] is None and nod["rdxdr"] is None and "seq" not in nod and np.array_equal(nod["x"], full["x"])
    with pytest.raises(NotImplementedError):
        SO3(nmax=5, lmax=5, rcut=5.0)             # ncoef = 90: beyond the device kernels, said at construction
    with pytest.raises(NotImplementedError):
        SO3(nmax=7, lmax=0, rcut=5.0)             # nmax > 6
    assert SO3(nmax=4, lmax=6, rcut=5.0).ncoefs == 70 and SO3(nmax=5, lmax=4, rcut=5.0).ncoefs == 75
    assert SO3(nmax=6, lmax=2, rcut=5.0).ncoefs == 63
    assert SO3(nmax=4, lmax=4, rcut=5.0).ncoefs == 50


@pytest.mark.parametrize("nl", [(4, 4), (6, 2), (5, 4)], ids=["n4l4_d50", "n6l2_d63", "n5l4_d75"])
def test_gp_end_to_end_with_a_long_descriptor(port_natives, nl):
    """SO3(nmax=4, lmax=4) / SO3(nmax=6, lmax=2) -> 64-column operand planes of the covariance kernels, SO3(nmax=5, lmax=4)
    -> 128-column planes (round 4) -> fit -> predict,
    against the oracle (numpy descriptor + plain-C natives + scipy) on the same structures; RBF and Dot (pair route: the
    degree-2 feature map covers 32 columns)"""
    NMAX, LMAX = nl
    NCOEF = NMAX * (NMAX + 1) // 2 * (LMAX + 1)
    from gpr_calculator_amd.gaussianprocess import GP
    from gpr_calculator_amd.SO3 import SO3
    from gpr_calculator_amd.kernels.RBF_mb import RBF_mb
    from gpr_calculator_amd.kernels.Dot_mb import Dot_mb
    from gpr_calculator_amd.synthetic import fcc_cell, lj_energy_forces
    from gpr_calculator_amd.neighbors import symbol_of
    from gpr_calculator_amd.utilities import species_numbers
    from oracle import gp_numpy as og
    from oracle.so3_numpy import so3_calculate
    strucs = [fcc_cell(seed=s, reps=1, species=(46, 78), sigma=0.08) for s in range(3)]
    labels = [lj_energy_forces(a) for a in strucs]
    test = fcc_cell(seed=9, reps=1, species=(46, 78), sigma=0.08)

    def odesc(a):
        d = so3_calculate(a.numbers, a.positions, a.cell, a.pbc, NMAX, LMAX, 5.0, 2.0)
        return d, species_numbers([symbol_of(z) for z in a.numbers])
    for kern, okern in ((RBF_mb(para=[2.5, 1.3], zeta=2.0), og.OracleRBF(para=(2.5, 1.3), zeta=2.0, natives=port_natives)),
                        (Dot_mb(para=[2.0, 2.0], zeta=2.0), og.OracleDot(para=(2.0, 2.0), zeta=2.0, natives=port_natives))):
        gp = GP(kernel=kern, descriptor=SO3(nmax=NMAX, lmax=LMAX, rcut=5.0), noise_e=0.002, noise_f=0.1, log_file="/tmp/gpr_test.log")
        gp.fit(gp.points_from_structures([(a, E, F, True, [0, 2]) for a, (E, F) in zip(strucs, labels)]), show=False, opt=False)
        assert gp.train_x["force"][0].shape[1] == NCOEF
        energy, force, y = [], [], []
        for a, (E, F) in zip(strucs, labels):
            d, ele = odesc(a)
            energy.append((d["x"], ele)); y.append(E / len(a))
        yF = []
        for a, (E, F) in zip(strucs, labels):
            d, ele = odesc(a)
            for at in (0, 2):
                ids = np.flatnonzero(d["seq"][:, 1] == at)
                ci = d["seq"][ids, 0]
                force.append((d["x"][ci], d["dxdr"][ids], ele[ci])); yF.append(F[at])
        tx = {"energy": og.list_to_tuple(energy, mode="energy"), "force": og.list_to_tuple(force)}
        yv = np.concatenate((np.array(y), np.array(yF).reshape(-1))).reshape(-1, 1)
        K = okern.k_total(tx)
        assert rel(gp.kernel.k_total(gp.train_x), K) < 1e-10
        L, alpha, K_inv = og.fit_core(K, yv, 3, 0.002, 0.1)
        d, ele = odesc(test)
        data = {"energy": og.list_to_tuple([(d["x"], ele)], mode="energy"), "force": []}
        for at in range(len(test)):
            ids = np.flatnonzero(d["seq"][:, 1] == at)
            ci = d["seq"][ids, 0]
            data["force"].append((d["x"][ci], d["dxdr"][ids], ele[ci]))
        Kt = okern.k_total(data, tx, 1e-12) if kern.name == "RBF" else okern.k_total(data, tx)
        mean = Kt.dot(alpha)[:, 0]
        E, F, _, E_std, F_std = gp.predict_structure(test, stress=False, return_std=True, f_tol=1e-12)
        assert abs(E - mean[0] * len(test)) < 1e-6 and np.abs(F - mean[1:].reshape(-1, 3)).max() < 1e-5
        assert np.isfinite(E_std) and np.isfinite(F_std).all()


def test_compile_time_instance_agrees_with_the_generic_kernel():
    """SO3(3, 4) runs so3_fixed_kernel<3, 4> (all sizes constants, re-mapped phases); gpr_so3_set_variant(1) sends the same
    batch through the generic kernel.  Ragged batch: clusters, a dense cell whose centres need several chunks of pairs,
    two species with sign-flipping weights, an isolated atom (no pairs, one row)."""
    import torch
    from gpr_calculator_amd import _lib
    from gpr_calculator_amd.SO3 import SO3
    from gpr_calculator_amd.neighbors import SimpleAtoms
    from gpr_calculator_amd.synthetic import fcc_cell, neb_images
    rng = np.random.default_rng(4)
    dense = fcc_cell(seed=5, reps=2, species=(46, 78), sigma=0.1)
    dense.positions *= 0.8; dense.cell = dense.cell * 0.8                       # ~80 pairs per centre: two chunks of 42
    cluster = SimpleAtoms(rng.choice([8, 12, 46], 9), rng.normal(0.0, 1.8, (9, 3)), np.eye(3) * 30.0, [False] * 3)
    lone = SimpleAtoms(np.array([46]), np.zeros((1, 3)), np.eye(3) * 30.0, [False] * 3)
    axis = SimpleAtoms(np.array([46, 46, 8]), np.array([[0.0, 0, 0], [0, 0, 2.5], [0, 0, -3.0]]), np.eye(3) * 30.0, [False] * 3)
    batch = neb_images(3, seed=7) + [dense, cluster, lone, axis]
    for weight_on in (False, True):
        des = SO3(nmax=3, lmax=4, rcut=5.0, alpha=2.0, weight_on=weight_on)
        try:
            _lib.lib().gpr_so3_set_variant(1)
            want = des.calculate_batch_dev(batch)
            torch.cuda.synchronize()
            want = {k: want[k].clone() for k in ("x", "dxdr")}
            _lib.lib().gpr_so3_set_variant(0)
            got = {}
            for prepass_min in (0, 1 << 30):          # radial channels from the streaming pre-pass / formed inside the centre kernel
                _lib.lib().gpr_so3_set_prepass_min(prepass_min)
                out = des.calculate_batch_dev(batch)
                torch.cuda.synchronize()
                got[prepass_min] = {k: out[k].clone() for k in ("x", "dxdr")}
        finally:
            _lib.lib().gpr_so3_set_variant(0)
            _lib.lib().gpr_so3_set_prepass_min(512)
        for k in ("x", "dxdr"):
            for g in got.values():
                assert g[k].shape == want[k].shape
                err = float((g[k] - want[k]).abs().max() / want[k].abs().max())
                assert err < 1e-13, (k, weight_on, err)
            # (the two forms of the instance run the same formulas in two kernels: the compiler contracts them differently
            #  here and there, so they agree to an ulp or two, not to the bit)
            assert float((got[0][k] - got[1 << 30][k]).abs().max() / want[k].abs().max()) < 1e-15
