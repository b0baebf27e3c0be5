"""GPU: edge cases of the device-resident prediction front end (neighbour search, descriptor, force-point gather,
feature / pair routes) -- ragged batches, fixed atoms, atoms without neighbours, structures larger than one
wavefront of atoms, non-periodic clusters, the empty batch -- each checked against the one-structure tuple route
(host descriptor download, packed sets), which in turn is pinned to the oracle in test_gpu_gp.py."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _model(kind):
    from gpr_calculator_amd.gaussianprocess import GP
    from gpr_calculator_amd.SO3 import SO3
    from gpr_calculator_amd.kernels.RBF_mb import RBF_mb
    from gpr_calculator_amd.kernels.Dot_mb import Dot_mb
    from gpr_calculator_amd.synthetic import fcc_cell, lj_energy_forces
    kern = RBF_mb(para=[2.5, 1.3], zeta=2.0) if kind == "RBF" else Dot_mb(para=[2.0, 2.0], zeta=2.0)
    gp = GP(kernel=kern, descriptor=SO3(nmax=3, lmax=4, rcut=5.0), noise_e=0.002, noise_f=0.1, log_file="/tmp/gpr_test.log")
    rows = []
    for seed in range(3):
        a = fcc_cell(seed=seed, reps=1, species=(46, 78), sigma=0.08)
        E, F = lj_energy_forces(a)
        rows.append((a, E, F, True, [0, 1, 3]))
    gp.fit(gp.points_from_structures(rows), show=False, opt=False)
    return gp


def _ragged_batch():
    from gpr_calculator_amd.synthetic import fcc_cell
    from gpr_calculator_amd.neighbors import SimpleAtoms
    rng = np.random.default_rng(11)
    big = fcc_cell(seed=5, reps=3, species=(46, 78), sigma=0.05)             # 108 atoms: more than one wavefront of j
    small = fcc_cell(seed=6, reps=1, species=(46, 78), sigma=0.08)
    small.fixed = np.array([0, 2])
    mid = fcc_cell(seed=7, reps=2, species=(46,), sigma=0.1)
    mid.fixed = np.array([31])
    cluster = SimpleAtoms(numbers=[46, 78, 46, 46, 78], positions=rng.normal(size=(5, 3)) * 1.6,
                          cell=np.eye(3) * 30.0, pbc=[False, False, False])
    # one atom 9 A away from everything: no neighbour within rcut, its descriptor row is zero
    lonely = SimpleAtoms(numbers=[46, 46, 78, 46], positions=[[0, 0, 0], [2.6, 0.1, 0], [0.2, 2.7, 0.1], [12.0, 12.0, 12.0]],
                         cell=np.eye(3) * 40.0, pbc=[False, False, False])
    return [big, small, mid, cluster, lonely]


@pytest.mark.parametrize("kind", ["Dot", "RBF"])
def test_ragged_batch_matches_one_structure_tuple_route(kind):
    gp = _model(kind)
    batch = _ragged_batch()
    Es, Fs, Es_std, Fs_std = gp.predict_structures(batch, return_std=True, f_tol=1e-10)
    gp.device_single = False
    for k, s in enumerate(batch):
        e1, f1, _, es1, fs1 = gp.predict_structure(s, stress=False, return_std=True, f_tol=1e-10)
        assert Fs[k].shape == (len(s), 3)
        assert abs(Es[k] - e1) < 1e-8 * max(1.0, abs(e1)) and np.abs(Fs[k] - f1).max() < 1e-8
        assert abs(Es_std[k] - es1) < 1e-7 and np.abs(Fs_std[k] - fs1).max() < 1e-7
    for k in (1, 2):                                                           # fixed atoms: zero force and zero std rows
        fx = batch[k].fixed
        assert np.all(Fs[k][fx] == 0.0) and np.all(Fs_std[k][fx] == 0.0)
    # the atom without neighbours: finite outputs, zero descriptor row
    d = gp.descriptor.calculate(batch[4])
    assert np.all(d["x"][3] == 0.0) and np.isfinite(Fs[4]).all() and np.isfinite(Fs_std[4]).all()
    # the device-resident single-structure route agrees as well
    gp.device_single = True
    e2, f2, _, es2, fs2 = gp.predict_structure(batch[0], stress=False, return_std=True, f_tol=1e-10)
    assert abs(e2 - Es[0]) < 1e-9 * max(1.0, abs(e2)) and np.abs(f2 - Fs[0]).max() < 1e-9


def test_empty_batch_and_results_without_std():
    gp = _model("Dot")
    assert gp.predict_structures([], return_std=True) == ([], [], [], [])
    batch = _ragged_batch()[:2]
    Es, Fs, Es_std, Fs_std = gp.predict_structures(batch, return_std=False)
    Es2, Fs2, _, _ = gp.predict_structures(batch, return_std=True)
    assert Es_std == [None, None] and Fs_std == [None, None]
    for k in range(2):
        assert abs(Es[k] - Es2[k]) < 1e-10 * max(1.0, abs(Es2[k])) and np.abs(Fs[k] - Fs2[k]).max() < 1e-10


def test_more_than_four_species_falls_back_to_pair_kernels():
    """the Dot feature map covers up to four species; beyond that the same calls take the row-pair kernels"""
    from gpr_calculator_amd.gaussianprocess import GP
    from gpr_calculator_amd.SO3 import SO3
    from gpr_calculator_amd.kernels.Dot_mb import Dot_mb
    from gpr_calculator_amd.synthetic import fcc_cell, lj_energy_forces
    sp = (8, 13, 46, 78, 79)
    gp = GP(kernel=Dot_mb(para=[2.0, 2.0], zeta=2.0), descriptor=SO3(nmax=3, lmax=4, rcut=5.0), noise_e=0.002, noise_f=0.1,
            log_file="/tmp/gpr_test.log")
    rows = []
    for seed in range(3):
        a = fcc_cell(seed=seed, reps=1, sigma=0.08)
        a.numbers = np.array([sp[(seed + i) % 5] for i in range(len(a))])
        a.numbers[0], a.numbers[1] = sp[seed % 5], sp[(seed + 3) % 5]
        E, F = lj_energy_forces(a)
        rows.append((a, E, F, True, [0, 1, 2, 3]))
    gp.fit(gp.points_from_structures(rows), show=False, opt=False)
    assert not gp._dense_pending                        # no weight-space fit: five species in the training set
    new = fcc_cell(seed=9, reps=2, sigma=0.08)
    new.numbers = np.array([sp[i % 5] for i in range(len(new))])
    Es, Fs, Es_std, Fs_std = gp.predict_structures([new], return_std=True)
    gp.device_single = False
    e1, f1, _, es1, fs1 = gp.predict_structure(new, stress=False, return_std=True)
    assert abs(Es[0] - e1) < 1e-8 * max(1.0, abs(e1)) and np.abs(Fs[0] - f1).max() < 1e-8
    assert abs(Es_std[0] - es1) < 1e-7 and np.abs(Fs_std[0] - fs1).max() < 1e-7


def test_native_batch_entry_matches_kernel_object_route():
    """gpr_predict_dot2_batch (one native call) against the same prediction assembled call by call through the kernel
    object (device descriptor, SetPairs, gpr_dot2_features_direct, gpr_gemm, gpr_ws_finish): same feature rows; with the
    variance the native call ends in the fused tail kernel (gpr_ws_tail: Psi w and the factored variance from MFMA tiles
    of Psi C^-T that are never stored), i.e. the same sums in another order; without it both routes end in
    gpr_ws_finish and agree bit for bit; with and without fixed atoms"""
    gp = _model("Dot")
    batch = _ragged_batch()
    for return_std in (True, False):
        gp.native_batch = True
        a = gp.predict_structures(batch, return_std=return_std)
        gp.native_batch = False
        b = gp.predict_structures(batch, return_std=return_std)
        for k in range(len(batch)):
            if return_std:
                assert abs(a[0][k] - b[0][k]) < 1e-11 * max(1.0, abs(b[0][k])) and np.abs(a[1][k] - b[1][k]).max() < 1e-11
                assert abs(a[2][k] - b[2][k]) < 1e-11 and np.abs(a[3][k] - b[3][k]).max() < 1e-11
            else:
                assert a[0][k] == b[0][k] and np.array_equal(a[1][k], b[1][k])
    gp.native_batch = True


def test_guessed_neighbour_capacities_and_their_overflow():
    """The native call sizes its neighbour arrays from the densest batch seen so far and does not wait for the count
    pass; results are those of the exact-size call bit for bit, and a guess that is too small is detected on the
    device, leaves every downstream kernel with empty ranges, and is repeated with exact sizes."""
    from gpr_calculator_amd.synthetic import fcc_cell
    gp = _model("Dot")
    batch = [fcc_cell(seed=70 + k, reps=1, species=(46, 78), sigma=0.08) for k in range(5)]
    # the session call keeps the same bookkeeping inside the library: exact first call, capped second, a hint that is
    # far too small (overflow flagged on the device, call repeated with exact sizes) -- identical results every time
    import ctypes as C0
    from gpr_calculator_amd import _lib as L0
    first = gp.predict_structures(batch, return_std=True)
    again = gp.predict_structures(batch, return_std=True)
    L0.lib().gpr_dot2_session_set_hint(gp._session[1], C0.c_double(0.5), C0.c_double(0.5))
    small = gp.predict_structures(batch, return_std=True)
    for other in (again, small):
        for k in range(len(batch)):
            assert first[0][k] == other[0][k] and np.array_equal(first[1][k], other[1][k])
            assert first[2][k] == other[2][k] and np.array_equal(first[3][k], other[3][k])
    gp.session_batch = False                                          # ... and the Python-side route of the same logic
    gp._nbr_hint = None
    exact = gp.predict_structures(batch, return_std=True)            # exact sizes; records pairs / rows per atom
    for k in range(len(batch)):
        assert exact[0][k] == first[0][k] and np.array_equal(exact[3][k], first[3][k])
    hint = gp._nbr_hint
    assert hint is not None and hint[0] > hint[1] > 1.0
    capped = gp.predict_structures(batch, return_std=True)           # capacities from the hint
    gp._nbr_hint = (0.5, 0.5)                                         # far too small: overflow, then the exact repeat
    over = gp.predict_structures(batch, return_std=True)
    assert gp._nbr_hint == hint                                       # the true sizes came back with the flagged result
    for other in (capped, over):
        for k in range(len(batch)):
            assert exact[0][k] == other[0][k] and np.array_equal(exact[1][k], other[1][k])
            assert exact[2][k] == other[2][k] and np.array_equal(exact[3][k], other[3][k])
    # the capped builder on its own: same table as the exact one, or the overflow flag over a table that stays inside the capacities
    import ctypes as C
    import torch
    from gpr_calculator_amd import _lib, device as dev
    from gpr_calculator_amd.neighbors import batch_arrays
    L = _lib.lib()
    atom0, numbers, pos, cells, pbc = batch_arrays(batch)
    a0, num = atom0.astype(np.int32), numbers.astype(np.int32)
    ip = lambda a: a.ctypes.data_as(_lib.c_int_p)
    dp = lambda a: a.ctypes.data_as(_lib.c_double_p)
    n = int(atom0[-1])

    def build(caps):
        h = C.c_void_p()
        if caps is None:
            dev.check(L.gpr_nbr_build_device(C.byref(h), dev.stream_ptr(), len(batch), ip(a0), dp(pos), dp(cells), ip(pbc), ip(num), 5.0, 0), "b")
        else:
            dev.check(L.gpr_nbr_build_device_capped(C.byref(h), dev.stream_ptr(), len(batch), ip(a0), dp(pos), dp(cells), ip(pbc), ip(num),
                                                    5.0, 0, caps[0], caps[1]), "b")
        info = torch.zeros(4, dtype=torch.float64, device="cuda")
        dev.check(L.gpr_nbr_info_to(h, dev.ptr(info)), "info")
        ptrs = [C.c_void_p() for _ in range(9)]
        dev.check(L.gpr_nbr_device_arrays(h, *[C.byref(q) for q in ptrs]), "arrays")
        P_, S_ = C.c_int(), C.c_int()
        L.gpr_nbr_sizes(h, C.byref(P_), C.byref(S_))
        info = info.cpu().numpy()
        P, S = int(info[2]), int(info[3])
        def grab(ptr, count, dtype):
            t = torch.empty(count, dtype=dtype, device="cuda")
            dev.check(L.gpr_memcpy_dtod(dev.stream_ptr(), dev.ptr(t), ptr, t.numel() * t.element_size()), "copy")
            return t.cpu().numpy()
        out = {"info": info, "sizes": (P_.value, S_.value), "cp0": grab(ptrs[2], n + 1, torch.int32), "cr0": grab(ptrs[3], n + 1, torch.int32)}
        if info[1] == 0:
            out.update(vec=grab(ptrs[0], 3 * P, torch.float64), w=grab(ptrs[1], P, torch.float64), rp0=grab(ptrs[4], S + 1, torch.int32),
                       self_row=grab(ptrs[5], n, torch.int32), seq=grab(ptrs[6], 2 * S, torch.int32))
        torch.cuda.synchronize()
        L.gpr_nbr_destroy(h)
        return out

    ref = build(None)
    P, S = ref["sizes"]
    assert ref["info"][1] == 0 and (int(ref["info"][2]), int(ref["info"][3])) == (P, S)
    roomy = build((P + 100, S + 7))
    assert roomy["sizes"] == (P + 100, S + 7) and roomy["info"][1] == 0
    for k in ("cp0", "cr0", "vec", "w", "rp0", "self_row", "seq"):
        assert np.array_equal(ref[k], roomy[k]), k
    tight = build((P, S))
    assert tight["info"][1] == 0 and np.array_equal(tight["seq"], ref["seq"])
    for caps in ((P - 1, S), (P, S - 1)):
        small = build(caps)
        assert small["info"][1] == 1 and (int(small["info"][2]), int(small["info"][3])) == (P, S)
        # the structures that fitted (a prefix of the batch) as in the exact table, every centre behind them empty, and
        # nothing beyond the capacities (the one-launch builder; the five launches empty the whole table)
        cp0, cr0 = small["cp0"], small["cr0"]
        assert (np.diff(cp0) >= 0).all() and (np.diff(cr0) >= 0).all() and cp0[-1] <= caps[0] and cr0[-1] <= caps[1]
        k = int(np.argmax(cp0 == cp0[-1]))                       # first centre closed at the end of the part that fitted
        assert np.array_equal(cp0[:k + 1], ref["cp0"][:k + 1]) and np.array_equal(cr0[:k + 1], ref["cr0"][:k + 1])


def test_error_paths_fail_loudly():
    """no silent fallbacks: a covariance that is not positive definite, a descriptor too long for this build, a
    mismatched counts vector and an unfitted model all raise"""
    import torch
    from gpr_calculator_amd import device as dev, _lib
    from gpr_calculator_amd.synthetic import fcc_cell
    # Cholesky of an indefinite matrix reports the failing pivot
    A = torch.eye(70, dtype=torch.float64, device="cuda")
    A[40, 40] = -1.0
    info = dev.potrf_lower_(A)
    info = info[0] if isinstance(info, tuple) else info
    assert int(info.item()) == 41
    # GP.fit turns that into LinAlgError: two identical structures with zero noise give a singular K
    gp = _model("RBF")
    s = fcc_cell(seed=1, reps=1, species=(46, 78), sigma=0.08)
    from gpr_calculator_amd.synthetic import lj_energy_forces
    E, F = lj_energy_forces(s)
    gp.noise_e, gp.noise_f = 0.0, 0.0
    with pytest.raises(np.linalg.LinAlgError):
        gp.fit(gp.points_from_structures([(s, E, F, True, [0, 1]), (s, E, F, True, [0, 1])]), show=False, opt=False)
    # descriptor longer than the packed operand width
    with pytest.raises(_lib.GprHipError):
        dev.PackedSet(_lib.ROLE_STREAM, np.ones((4, 130)), None, np.full(4, 46), [4])         # 128 columns at most
    # counts that do not add up
    with pytest.raises(ValueError):
        dev.PackedSet(_lib.ROLE_STREAM, np.ones((4, 30)), None, np.full(4, 46), [3])
    # SO3 configuration outside the descriptor kernel's limits
    from gpr_calculator_amd.SO3 import SO3
    with pytest.raises(NotImplementedError):                  # said when the object is configured
        SO3(nmax=5, lmax=5, rcut=5.0)


@pytest.mark.parametrize("kind", ["Dot", "RBF"])
def test_species_unseen_in_training_contributes_nothing(kind):
    """a predicted structure with an element the training set never saw (the kernels are zero across species): the
    batch routes and the tuple route agree, and the energy of the foreign atoms' rows is pure prior"""
    from gpr_calculator_amd.synthetic import fcc_cell
    gp = _model(kind)
    s = fcc_cell(seed=21, reps=1, species=(46, 78), sigma=0.08)
    s.numbers = np.array([46, 78, 29, 46])                     # Cu is not in the training set
    Es, Fs, Es_std, Fs_std = gp.predict_structures([s], return_std=True)
    gp.device_single = False
    e1, f1, _, es1, fs1 = gp.predict_structure(s, stress=False, return_std=True)
    assert abs(Es[0] - e1) < 1e-9 * max(1.0, abs(e1)) and np.abs(Fs[0] - f1).max() < 1e-9
    assert abs(Es_std[0] - es1) < 1e-8 and np.abs(Fs_std[0] - fs1).max() < 1e-8


def test_base_potential_is_subtracted_and_added_back():
    """GP(base_potential=LJ): the model is trained on the residuals (gaussianprocess.py:579-583, 930-936) and the base
    is added to every prediction (:895-901); labels produced by that very LJ leave zero residuals, so predictions
    reproduce the base potential and the batch route agrees with the one-structure route"""
    from gpr_calculator_amd.calculator import LJ
    from gpr_calculator_amd.gaussianprocess import GP
    from gpr_calculator_amd.SO3 import SO3
    from gpr_calculator_amd.kernels.Dot_mb import Dot_mb
    from gpr_calculator_amd.synthetic import fcc_cell, lj_energy_forces
    lj = LJ({"epsilon": 0.5, "sigma": 2.5, "rc": 5.0})
    gp = GP(kernel=Dot_mb(para=[2.0, 2.0], zeta=2.0), descriptor=SO3(nmax=3, lmax=4, rcut=5.0), base_potential=lj,
            noise_e=0.002, noise_f=0.1, log_file="/tmp/gpr_test.log")
    for seed in range(3):
        a = fcc_cell(seed=seed, reps=1, species=(46, 78), sigma=0.08)
        E, F = lj_energy_forces(a)
        gp.add_structure((a, E, F))
    gp.fit(show=False, opt=False)
    assert np.abs(gp.y_train).max() < 1e-10                     # residuals
    new = fcc_cell(seed=33, reps=1, species=(46, 78), sigma=0.08)
    E0, F0 = lj_energy_forces(new)
    Es, Fs, _, _ = gp.predict_structures([new], return_std=True)
    assert abs(Es[0] - E0) < 1e-8 and np.abs(Fs[0] - F0).max() < 1e-8
    gp.device_single = False
    e1, f1, _ = gp.predict_structure(new, stress=False)
    assert abs(e1 - E0) < 1e-8 and np.abs(f1 - F0).max() < 1e-8


@pytest.mark.parametrize("kind", ["Dot", "RBF"])
def test_save_and_load_reproduce_the_model(kind, tmp_path):
    """GP.save -> json + ASE sqlite db (gaussianprocess.py:632-724), GP.load -> descriptors recomputed from the
    stored structures, same training set, same predictions after the same fit; with an LJ base potential in the
    json as well"""
    from gpr_calculator_amd.gaussianprocess import GP
    from gpr_calculator_amd.calculator import LJ
    from gpr_calculator_amd.synthetic import fcc_cell
    from gpr_calculator_amd import asedb
    gp = _model(kind)
    gp.base_potential = None
    js, db = str(tmp_path / "m.json"), str(tmp_path / "m.db")
    gp.save(js, db, verbose=False)
    rows = asedb.read_db(db)
    assert len(rows) == 3 and rows[0]["data"]["force_in"] == [0, 1, 3] and rows[0]["data"]["energy_in"] is True
    assert set(rows[0]["key_value_pairs"]) == {"dft_energy", "dft_fmax"}
    gp2 = GP.load(js)
    assert type(gp2.kernel) is type(gp.kernel) and gp2.kernel.parameters() == gp.kernel.parameters()
    assert (gp2.noise_e, gp2.noise_f, gp2.N_energy, gp2.N_forces) == (gp.noise_e, gp.noise_f, gp.N_energy, gp.N_forces)
    assert np.array_equal(gp2.y_train, gp.y_train)
    gp2.fit(show=False, opt=False)
    new = [fcc_cell(seed=60 + k, reps=1, species=(46, 78), sigma=0.08) for k in range(2)]
    a, b = gp.predict_structures(new, return_std=True), gp2.predict_structures(new, return_std=True)
    for k in range(2):
        assert abs(a[0][k] - b[0][k]) < 1e-9 and np.abs(a[1][k] - b[1][k]).max() < 1e-9
        assert abs(a[2][k] - b[2][k]) < 1e-8 and np.abs(a[3][k] - b[3][k]).max() < 1e-8
    # N_max keeps the first structures only; a base potential travels through the json
    assert GP.load(js, N_max=2).N_energy == 2
    gp.base_potential = LJ(parameters={"rc": 5.0, "sigma": 2.3, "epsilon": 0.1})
    gp.save(js, db, verbose=False)
    gp3 = GP.load(js)
    assert isinstance(gp3.base_potential, LJ) and gp3.base_potential.sigma == 2.3
    e_lj = gp.base_potential.calculate(rows[0]["atoms"])[0]
    got = asedb.read_db(db)[0]["key_value_pairs"]["dft_energy"]
    assert abs(got - (rows[0]["data"]["energy"] + e_lj) / len(rows[0]["atoms"])) < 1e-12


def test_shared_shift_tables_and_positions_outside_the_cell():
    """Structures that share the cell of their predecessor share its image table on the device (runs of equal cells in
    any order in the batch), and atoms outside the cell -- by whole lattice vectors, or sitting exactly on a face --
    are wrapped the slow, exact way: predictions are those of the structures one at a time."""
    from gpr_calculator_amd.synthetic import fcc_cell
    gp = _model("Dot")
    a = [fcc_cell(seed=20 + k, reps=1, species=(46, 78), sigma=0.08) for k in range(3)]          # equal cells
    b = fcc_cell(seed=30, reps=1, species=(46, 78), sigma=0.08)
    b.cell = b.cell * 1.05                                                                        # another cell in between
    b.positions = b.positions * 1.05
    far = fcc_cell(seed=22, reps=1, species=(46, 78), sigma=0.08)                                 # a[2] again, displaced
    far.positions = far.positions + 2.0 * far.cell[0] - 3.0 * far.cell[1] + far.cell[2]
    far.positions[1] = far.cell[0] * 1.0                                                          # exactly on a lattice point
    edge = fcc_cell(seed=22, reps=1, species=(46, 78), sigma=0.08)
    edge.positions[1] = 0.0
    batch = [a[0], a[1], b, a[2], far, edge]
    Es, Fs, Es_std, Fs_std = gp.predict_structures(batch, return_std=True)
    gp.device_single = False
    for k, s in enumerate(batch):
        e1, f1, _, es1, fs1 = gp.predict_structure(s, stress=False, return_std=True)
        assert abs(Es[k] - e1) < 1e-8 * max(1.0, abs(e1)) and np.abs(Fs[k] - f1).max() < 1e-8
        assert abs(Es_std[k] - es1) < 1e-7 and np.abs(Fs_std[k] - fs1).max() < 1e-7
    # the displaced copy is the same crystal: same energy, same forces
    assert abs(Es[4] - Es[5]) < 1e-9 * max(1.0, abs(Es[5])) and np.abs(Fs[4] - Fs[5]).max() < 1e-9
