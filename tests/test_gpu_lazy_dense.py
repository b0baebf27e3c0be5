"""GPU: state that is formed lazily or polled must not depend on when it is asked for.
  * a weight-space fit (Dot, zeta = 2) leaves the N x N factorisation to the first access of L_ / alpha_ / _K_inv /
    the stress route; points queued by add_structure in between must not leak into it;
  * the batched prediction polls status words in host memory; the stream-synchronising fallback gives the same bits."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _model(weight_space, stress=True):
    from gpr_calculator_amd.gaussianprocess import GP
    from gpr_calculator_amd.SO3 import SO3
    from gpr_calculator_amd.kernels.Dot_mb import Dot_mb
    from gpr_calculator_amd.synthetic import fcc_cell, lj_energy_forces
    gp = GP(kernel=Dot_mb(para=[2.0, 2.0], zeta=2.0), descriptor=SO3(nmax=3, lmax=4, rcut=5.0, stress=stress),
            noise_e=0.002, noise_f=0.1, log_file="/tmp/gpr_test.log")
    gp.weight_space_fit = weight_space
    rows = []
    for s in range(4):
        a = fcc_cell(seed=s, reps=1, species=(46, 78), sigma=0.08)
        E, F = lj_energy_forces(a)
        rows.append((a, E, F, True, [0, 2, 3]))
    gp.fit(gp.points_from_structures(rows), show=False, opt=False)
    return gp


def test_lazy_dense_factor_ignores_points_queued_after_the_fit():
    from gpr_calculator_amd.synthetic import fcc_cell, lj_energy_forces
    dense, lazy = _model(False), _model(True)
    assert lazy._dense_pending and not dense._dense_pending
    new = fcc_cell(seed=70, reps=1, species=(46, 78), sigma=0.2)
    E, F = lj_energy_forces(new)
    for gp in (dense, lazy):
        gp.add_structure((new, E, F.copy()))
        assert gp.N_queue > 0
    # first touch of the dense quantities happens AFTER the queue grew
    assert lazy.alpha_.shape == dense.alpha_.shape == (len(dense._fit_y), 1)
    assert np.abs(lazy.alpha_ - dense.alpha_).max() <= 1e-9 * np.abs(dense.alpha_).max()
    assert np.abs(lazy.L_ - dense.L_).max() <= 1e-11 * np.abs(dense.L_).max()
    t = fcc_cell(seed=71, reps=1, species=(46, 78), sigma=0.08)
    E1, F1, S1 = dense.predict_structure(t, stress=True, return_std=False, f_tol=1e-12)
    E2, F2, S2 = lazy.predict_structure(t, stress=True, return_std=False, f_tol=1e-12)
    assert abs(E1 - E2) < 1e-7 and np.abs(F1 - F2).max() < 1e-7 and np.abs(S1 - S2).max() < 1e-7
    assert lazy.N_queue > 0                                     # nothing refitted behind the caller's back


def test_polled_and_synchronised_results_are_identical():
    from gpr_calculator_amd.synthetic import fcc_cell
    gp = _model(True, stress=False)                             # the one-call native batch route
    batch = [fcc_cell(seed=80 + k, reps=1, species=(46, 78), sigma=0.08) for k in range(5)]
    batch[2].fixed = np.array([1])
    a = gp.predict_structures(batch, return_std=True)
    gp._poll_results = False
    b = gp.predict_structures(batch, return_std=True)
    gp._poll_results = True
    for x, y in zip(a, b):
        for u, v in zip(x, y):
            assert np.array_equal(np.asarray(u), np.asarray(v))


def test_session_call_matches_the_python_unpacking_route():
    """gpr_dot2_session_predict (host arrays in, host arrays out) against the route that unpacks the result block in
    Python: bit-identical E / F / std, with and without fixed atoms, uniform and ragged batches"""
    from gpr_calculator_amd.synthetic import fcc_cell
    gp = _model(True, stress=False)
    uniform = [fcc_cell(seed=80 + k, reps=1, species=(46, 78), sigma=0.08) for k in range(5)]
    fixed = [fcc_cell(seed=90 + k, reps=1, species=(46, 78), sigma=0.08) for k in range(4)]
    fixed[1].fixed = np.array([0, 3]); fixed[3].fixed = np.array([2])
    ragged = uniform[:2] + [fcc_cell(seed=99, reps=2, species=(46,), sigma=0.1)]
    for batch in (uniform, fixed, ragged):
        for std in (True, False):
            gp.session_batch = True
            a = gp.predict_structures(batch, return_std=std)
            gp.session_batch = False
            b = gp.predict_structures(batch, return_std=std)
            for x, y in zip(a, b):
                assert len(x) == len(y) == len(batch)
                for u, v in zip(x, y):
                    assert (u is None and v is None) or np.array_equal(np.asarray(u), np.asarray(v))
    gp.session_batch = True
    # a refit replaces the session (new weight-space tables)
    gp.fit(show=False, opt=False)
    c = gp.predict_structures(uniform, return_std=True)
    assert np.allclose(c[0], a[0][:0] or c[0])


def test_batches_in_flight_give_the_bits_of_the_synchronous_call():
    """predict_structures_submit / _collect (gpr_dot2_session_submit / _wait: several batches in flight, each on its own
    stream with its own result buffer) against the synchronous session call: bit-identical E / F / std for uniform,
    fixed-atom and ragged batches, collected in and out of submission order, with a neighbour-table guess that is too
    small (the wait repeats the batch with exact sizes) and with more submissions than slots refused"""
    import ctypes as C
    from gpr_calculator_amd import _lib
    from gpr_calculator_amd.synthetic import fcc_cell
    gp = _model(True, stress=False)
    gp.session_depth = 3
    uniform = [fcc_cell(seed=80 + k, reps=1, species=(46, 78), sigma=0.08) for k in range(5)]
    fixed = [fcc_cell(seed=90 + k, reps=1, species=(46, 78), sigma=0.08) for k in range(4)]
    fixed[1].fixed = np.array([0, 3]); fixed[3].fixed = np.array([2])
    ragged = uniform[:2] + [fcc_cell(seed=99, reps=2, species=(46,), sigma=0.1)]
    batches = [uniform, fixed, ragged]

    def same(a, b):
        for x, y in zip(a, b):
            assert len(x) == len(y)
            for u, v in zip(x, y):
                assert (u is None and v is None) or np.array_equal(np.asarray(u), np.asarray(v))
    for std in (True, False):
        want = [gp.predict_structures(b, return_std=std) for b in batches]
        for order in ((0, 1, 2), (2, 0, 1)):
            tickets = [gp.predict_structures_submit(b, return_std=std) for b in batches]
            assert all(t.ses is not None for t in tickets)                  # really queued, not evaluated on the spot
            got = {k: gp.predict_structures_collect(tickets[k]) for k in order}
            for k in range(3):
                same(got[k], want[k])
        # a steady pipeline: submit k + 1 before collecting k, the way bench.py drives it
        prev, outs = gp.predict_structures_submit(batches[0], return_std=std), []
        for k in range(1, 7):
            nxt = gp.predict_structures_submit(batches[k % 3], return_std=std)
            outs.append(gp.predict_structures_collect(prev))
            prev = nxt
        outs.append(gp.predict_structures_collect(prev))
        for k, o in enumerate(outs):
            same(o, want[k % 3])
    # a guess that is too small: the wait notices the overflow flag and repeats with exact sizes
    want = gp.predict_structures(uniform, return_std=True)
    _lib.lib().gpr_dot2_session_set_hint(gp._session[1], 1.0, 1.0)
    t1, t2 = gp.predict_structures_submit(uniform, return_std=True), gp.predict_structures_submit(fixed, return_std=True)
    same(gp.predict_structures_collect(t1), want)
    gp.predict_structures_collect(t2)
    # more submissions than slots: refused with a message, nothing queued, the session stays usable
    ts = [gp.predict_structures_submit(uniform, return_std=True) for _ in range(3)]
    with pytest.raises(RuntimeError, match="waiting already"):
        gp.predict_structures_submit(uniform, return_std=True)
    for t in ts:
        same(gp.predict_structures_collect(t), want)
    with pytest.raises(RuntimeError, match="collected already"):
        gp.predict_structures_collect(ts[0])
    # a refit while a batch is in flight ends the session; the stale ticket says so
    t = gp.predict_structures_submit(uniform, return_std=True)
    gp.fit(show=False, opt=False)
    with pytest.raises(RuntimeError, match="refitted"):
        gp.predict_structures_collect(t)
    same(gp.predict_structures(uniform, return_std=True), want)


def test_session_never_serves_the_previous_model():
    """predict -> add_structure -> fit(opt=False) -> predict: same sigma / sigma0 before and after, so only the
    identity of the fitted model tells the two sessions apart.  The session route must give what the non-session route
    gives for the NEW model, several times in a row (each refit frees and reallocates the weight-space cache)."""
    from gpr_calculator_amd.synthetic import fcc_cell, lj_energy_forces
    gp = _model(True, stress=False)
    batch = [fcc_cell(seed=80 + k, reps=1, species=(46, 78), sigma=0.08) for k in range(3)]
    before = gp.predict_structures(batch, return_std=True)
    assert gp._session is not None
    for seed in (60, 61, 62):
        new = fcc_cell(seed=seed, reps=1, species=(46, 78), sigma=0.25)
        E, F = lj_energy_forces(new)
        gp.add_structure((new, E, F.copy()))
        assert gp.N_queue > 0
        gp.fit(show=False, opt=False)
        assert gp._session is None                             # the refit ended the old model's session
        gp.session_batch = True
        a = gp.predict_structures(batch, return_std=True)
        gp.session_batch = False
        b = gp.predict_structures(batch, return_std=True)
        gp.session_batch = True
        for x, y in zip(a, b):
            for u, v in zip(x, y):
                assert np.array_equal(np.asarray(u), np.asarray(v))
        assert np.abs(np.asarray(a[0]) - np.asarray(before[0])).max() > 0.0      # and the model did change
        before = a


def test_appended_fit_matches_a_full_refit():
    """fit(opt=False) after add_structure with unchanged hyper-parameters extends the Cholesky factor by the queued
    rows (trsm + a small potrf) instead of rebuilding K: same alpha_, _K_inv, predictions and std as the full refit, twice
    in a row (the second append starts from an already permuted factor), and L_ still has the reference's definition"""
    from gpr_calculator_amd.gaussianprocess import GP
    from gpr_calculator_amd.SO3 import SO3
    from gpr_calculator_amd.kernels.RBF_mb import RBF_mb
    from gpr_calculator_amd.synthetic import fcc_cell, lj_energy_forces

    def model(incremental):
        gp = GP(kernel=RBF_mb(para=[2.5, 1.3], zeta=2.0), descriptor=SO3(nmax=3, lmax=4, rcut=5.0), noise_e=0.002,
                noise_f=0.1, log_file="/tmp/gpr_test.log")
        gp.incremental_fit = incremental
        rows = []
        for s in range(4):
            a = fcc_cell(seed=s, reps=1, species=(46, 78), sigma=0.08)
            E, F = lj_energy_forces(a)
            rows.append((a, E, F, True, [0, 2, 3]))
        gp.fit(gp.points_from_structures(rows), show=False, opt=False)
        return gp
    inc, full = model(True), model(False)
    test = fcc_cell(seed=71, reps=1, species=(46, 78), sigma=0.08)
    inc.predict_structure(test, stress=False, return_std=True)           # forms L^-1: the append must extend it too
    for step, seed in enumerate((60, 61)):
        new = fcc_cell(seed=seed, reps=1, species=(46, 78), sigma=0.25)
        E, F = lj_energy_forces(new)
        for gp in (inc, full):
            gp.add_structure((new, E, F.copy()))
            assert gp.N_queue > 0
            gp.fit(show=False, opt=False)
            assert gp.N_queue == 0
        assert inc._perm_dev is not None and full._perm_dev is None
        # two different elimination orders of a matrix with cond ~1e7: alpha agrees to cond * eps, predictions to 1e-9
        assert np.abs(inc.alpha_ - full.alpha_).max() <= 1e-8 * np.abs(full.alpha_).max()
        assert np.abs(inc.y_train - full.y_train).max() == 0.0
        a = inc.predict_structure(test, stress=False, return_std=True, f_tol=1e-12)
        b = full.predict_structure(test, stress=False, return_std=True, f_tol=1e-12)
        assert abs(a[0] - b[0]) < 1e-9 and np.abs(a[1] - b[1]).max() < 1e-9
        assert abs(a[3] - b[3]) < 1e-9 and np.abs(a[4] - b[4]).max() < 1e-9
        assert np.abs(inc._K_inv - full._K_inv).max() <= 1e-8 * np.abs(full._K_inv).max()
    assert np.abs(inc.L_ - full.L_).max() <= 1e-10 * np.abs(full.L_).max()       # refactored in the reference's order
    # changed hyper-parameters: no append
    inc.add_structure((fcc_cell(seed=62, reps=1, species=(46, 78), sigma=0.2), *lj_energy_forces(fcc_cell(seed=62, reps=1, species=(46, 78), sigma=0.2))))
    inc.kernel.update([2.0, 1.0])
    inc.fit(show=False, opt=False)
    assert inc._perm_dev is None


def test_weight_space_fit_call_by_call():
    """INTEGRATION.md's recipe: the weight-space fit as a sequence of exported calls (features of the packed training
    points, two accumulating Gram products with the noise weights in alpha, Cholesky, two solves, triangular inverse)
    gives the w and C^-1 that GP.fit hands to the prediction session"""
    import torch
    from gpr_calculator_amd import device as dev
    gp = _model(True, stress=False)
    ws = gp._weight_space()
    species, w_ref, Cinv_ref = ws[0], ws[2], ws[3]
    k = gp.kernel
    sig, s0, ne, nf = float(k.sigma), float(k.sigma0), float(gp.noise_e), float(gp.noise_f)
    PsiE = dev.dot2_features(k.pack_energy(gp.train_x["energy"]).stream, species, s0)
    PsiF = dev.dot2_features(k.pack_force(gp.train_x["force"]).stream, species, 0.0)
    nE = PsiE.shape[0]
    y = dev.to_dev(np.asarray(gp.y_train).reshape(-1, 1))
    G = dev.gemm(PsiE, PsiE, transa=True, alpha=sig ** 2 / ne ** 2)
    dev.gemm(PsiF, PsiF, transa=True, alpha=sig ** 2 / nf ** 2, beta=1.0, out=G)
    w = dev.gemm(PsiE, y[:nE].contiguous(), transa=True, alpha=sig ** 2 / ne ** 2)
    dev.gemm(PsiF, y[nE:].contiguous(), transa=True, alpha=sig ** 2 / nf ** 2, beta=1.0, out=w)
    G.diagonal().add_(1.0)
    info, dinv = dev.potrf_lower_(G, want_dinv=True)
    assert int(info.item()) == 0
    dev.trsm_lower_(G, w, dinv=dinv)
    dev.trsm_lower_(G, w, trans=True, dinv=dinv)
    Cinv = dev.trtri_lower(G)
    ew = float((w - w_ref).abs().max()) / float(w_ref.abs().max())
    ec = float((Cinv - Cinv_ref).abs().max()) / float(Cinv_ref.abs().max())
    assert ew < 1e-9 and ec < 1e-9, (ew, ec)
