"""GPU: the end of a prediction session.
  * Round 3 crashed once in `gpr_set_destroy` (PackedSet.__del__) after a session had been destroyed: a cached
    stream-ordered temporary remembered the slot stream it was freed on and, evicted later by an unrelated free, was
    handed to the driver on the destroyed handle.  gpr_dot2_session_destroy now returns a slot stream's cached blocks
    first (gpr_dev_cache_drop_stream); this file walks exactly that path: slot streams with live cached blocks ->
    destroy -> frees on other streams that evict everything the cache held.
  * A synchronous predict while tickets are outstanding takes a free slot instead of refusing."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _model():
    from gpr_calculator_amd.gaussianprocess import GP
    from gpr_calculator_amd.SO3 import SO3
    from gpr_calculator_amd.kernels.Dot_mb import Dot_mb
    from gpr_calculator_amd.synthetic import fcc_cell, lj_energy_forces
    gp = GP(kernel=Dot_mb(para=[2.0, 2.0], zeta=2.0), descriptor=SO3(nmax=3, lmax=4, rcut=5.0),
            noise_e=0.002, noise_f=0.1, log_file="/tmp/gpr_test.log")
    rows = []
    for s in range(4):
        a = fcc_cell(seed=s, reps=1, species=(46, 78), sigma=0.08)
        E, F = lj_energy_forces(a)
        rows.append((a, E, F, True, [0, 2, 3]))
    gp.fit(gp.points_from_structures(rows), show=False, opt=False)
    return gp


def test_destroying_a_session_with_cached_blocks_on_its_slot_streams_then_freeing_sets_elsewhere():
    import torch
    from gpr_calculator_amd import _lib, device as dev
    from gpr_calculator_amd.kernels.RBF_mb import RBF_mb
    from gpr_calculator_amd.synthetic import fcc_cell
    L = _lib.lib()
    gp = _model()
    gp.session_depth = 4
    batches = [[fcc_cell(seed=300 + 10 * b + k, reps=1, species=(46, 78), sigma=0.08) for k in range(2 + b)] for b in range(4)]
    want = [gp.predict_structures(b, return_std=True) for b in batches]
    for _ in range(3):                                            # every slot has run, freed and cached temporaries
        tickets = [gp.predict_structures_submit(b, return_std=True) for b in batches]
        for t, w in zip(tickets, want):
            got = gp.predict_structures_collect(t)
            assert all(np.array_equal(np.asarray(a), np.asarray(b)) for x, y in zip(got, w) for a, b in zip(x, y))
    ses = gp._session[1]
    slot_streams = [L.gpr_dot2_session_slot_stream(ses, k) for k in range(1, 4)]
    assert all(slot_streams), "slots 1..3 own streams by now"
    assert sum(L.gpr_dev_cache_count(s, 0) for s in slot_streams) > 0, "the path under test: slot streams hold cached blocks"
    gp._drop_session()                                            # what every refit does
    assert gp._session is None
    for s in slot_streams:                                        # nothing may remember a destroyed stream
        assert L.gpr_dev_cache_count(s, 0) == 0
    # frees on another stream, more of them than the cache has places: every block cached so far is evicted and released
    # on the stream it remembers (the call that crashed did this from PackedSet.__del__)
    kern = RBF_mb(para=[2.5, 1.3], zeta=2.0)
    side = torch.cuda.Stream()
    X, ELE, counts = gp.train_x["energy"]
    with torch.cuda.stream(side):
        for k in range(40):
            sets = [dev.PackedSet(_lib.ROLE_STREAM if k % 2 else _lib.ROLE_RESIDENT, X, None, ELE, counts) for _ in range(2)]
            for ps in sets:
                ps.close()                                        # gpr_set_destroy: the call the crash surfaced in
    side.synchronize()
    torch.cuda.synchronize()
    K = kern.k_total(gp.train_x)                                  # the process is alive and computes
    assert np.isfinite(K).all()
    again = gp.predict_structures(batches[0], return_std=True)   # a new session on the same model
    assert all(np.array_equal(np.asarray(a), np.asarray(b)) for x, y in zip(again, want[0]) for a, b in zip(x, y))


def test_synchronous_calls_and_tickets_share_a_session():
    from gpr_calculator_amd.synthetic import fcc_cell
    gp = _model()
    gp.session_depth = 2
    b0 = [fcc_cell(seed=400 + k, reps=1, species=(46, 78), sigma=0.08) for k in range(3)]
    b1 = [fcc_cell(seed=410 + k, reps=1, species=(46, 78), sigma=0.08) for k in range(5)]
    w0, w1 = gp.predict_structures(b0, return_std=True), gp.predict_structures(b1, return_std=True)
    same = lambda a, b: all(np.array_equal(np.asarray(u), np.asarray(v)) for x, y in zip(a, b) for u, v in zip(x, y))
    t0 = gp.predict_structures_submit(b0, return_std=True)       # holds slot 0, the session's own stream
    assert t0.ses is not None
    assert same(gp.predict_structures(b1, return_std=True), w1)  # synchronous call beside it: a free slot
    e, f, _, es, fs = gp.predict_structure(b1[0], stress=False, return_std=True)     # what GPR.calculate does
    assert e == w1[0][0] and np.array_equal(f, w1[1][0]) and es == w1[2][0] and np.array_equal(fs, w1[3][0])
    t1 = gp.predict_structures_submit(b1, return_std=True)       # both submission slots taken now
    assert same(gp.predict_structures(b0, return_std=True), w0)  # still served: slots beyond the submission depth
    assert same(gp.predict_structures_collect(t1), w1) and same(gp.predict_structures_collect(t0), w0)
