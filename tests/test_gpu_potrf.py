"""Cholesky with look-ahead (csrc/potrf.hip: n > 768 -- diagonal blocks polling for their last update, the top rows of a
panel solved beside the diagonal block through stage counters, narrow updates on syrk_narrow_kernel) and the recursive
triangular inverse, against scipy / numpy; replaces scipy.linalg.cholesky of gaussianprocess.py:175, 297."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


@pytest.fixture(scope="module")
def gpu():
    from gpr_calculator_amd import _lib
    _lib.require_gpu()
    return _lib.lib()


def _spd(n, seed):
    rng = np.random.default_rng(seed)
    b = rng.standard_normal((n, max(n // 2, 4)))
    return b @ b.T + (0.01 * n + 1.0) * np.eye(n)


@pytest.mark.parametrize("n", [769, 1025, 1300, 2049, 2304])
def test_cholesky_with_lookahead(gpu, n):
    """block edges: one row into the fourth block, a ragged last block, a last block of one row, whole blocks only"""
    import torch
    from scipy.linalg import cholesky
    from gpr_calculator_amd import device as dev
    A = _spd(n, n)
    L = cholesky(A, lower=True)
    for rep in range(2):                                   # the second call reuses the counters of the first
        t = torch.tensor(A, device="cuda")
        info, dinv = dev.potrf_lower_(t, want_dinv=True)
        got = t.cpu().numpy()
        assert int(info.item()) == 0
        assert np.abs(got - L).max() / np.abs(L).max() < 1e-13
        assert np.abs(np.triu(got, 1)).max() == 0.0
    W = dinv.cpu().numpy()[:((n + 63) // 64) * 4096].reshape(-1, 64, 64)
    for b in (0, len(W) // 2, len(W) - 1):                 # stored inverses of the 64 x 64 diagonal blocks
        k0, kb = 64 * b, min(64, n - 64 * b)
        assert np.abs(W[b][:kb, :kb] @ L[k0:k0 + kb, k0:k0 + kb] - np.eye(kb)).max() < 1e-12


def test_cholesky_reports_the_first_bad_pivot_with_lookahead(gpu):
    import torch
    from gpr_calculator_amd import device as dev
    A = np.eye(1500)
    A[1100, 1100] = -1.0
    assert int(dev.potrf_lower_(torch.tensor(A, device="cuda")).item()) == 1101


@pytest.mark.parametrize("n", [1, 100, 257, 466, 700, 1300, 2049])
def test_triangular_inverse(gpu, n):
    import torch
    from scipy.linalg import cholesky
    from gpr_calculator_amd import device as dev
    L = cholesky(_spd(n, 7 * n + 1), lower=True)
    X = dev.trtri_lower(torch.tensor(np.ascontiguousarray(L), device="cuda")).cpu().numpy()    # scipy hands back column-major
    assert np.abs(np.triu(X, 1)).max() == 0.0
    assert np.abs(X @ L - np.eye(n)).max() < 1e-11


def test_event_only_schedule_matches(gpu):
    """GPR_POTRF_NO_POLL=1: the schedule a profiler that serialises kernels gets (every hand-over an event); fresh
    process, because the choice is made once per process"""
    env = dict(os.environ, GPR_POTRF_NO_POLL="1")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "potrf_check.py"), "1025", "2049"], env=env, cwd=ROOT,
                         capture_output=True, text=True, timeout=600)
    assert res.returncode == 0 and "ALL OK" in res.stdout, res.stdout + res.stderr


def test_poll_timeout_is_a_schedule_event_not_a_verdict_on_the_matrix(gpu):
    """A hand-over between two running kernels that is never answered (forced: a poll budget of one) makes the native
    call return info = -1 with the matrix partly overwritten.  device.potrf_factor then orders the kernels by events
    for the rest of the process, builds the matrix again and factors it: a correct L, info = 0, and -- for a matrix that
    really is indefinite -- still the pivot index, never -1."""
    import torch
    from scipy.linalg import cholesky
    from gpr_calculator_amd import device as dev
    n = 2049
    A = _spd(n, 11)
    L = cholesky(A, lower=True)
    if not gpu.gpr_potrf_polling():
        # fill the context (first factorisation with look-ahead probes the device)
        dev.potrf_lower_(torch.tensor(A, device="cuda"))
    if not gpu.gpr_potrf_polling():
        pytest.skip("kernels of two streams do not overlap here (profiler attached?): nothing polls, nothing can time out")
    try:
        assert gpu.gpr_potrf_debug_max_polls(1) == 0
        t = torch.tensor(A, device="cuda")
        assert int(dev.potrf_lower_(t).item()) == -1                  # the raw call: time-out reported as such
        assert gpu.gpr_potrf_polling() == 1                           # ... and nothing switched behind the caller's back
        built = []

        def make():
            built.append(1)
            return torch.tensor(A, device="cuda")
        got, info, dinv = dev.potrf_factor(make)
        assert len(built) == 2 and int(info.item()) == 0
        assert gpu.gpr_potrf_polling() == 0                           # events only from here on
        assert np.abs(got.cpu().numpy() - L).max() / np.abs(L).max() < 1e-13
        B = A.copy()
        B[1500, 1500] = -1.0
        _, info, _ = dev.potrf_factor(lambda: torch.tensor(B, device="cuda"))
        assert int(info.item()) == 1501
    finally:
        gpu.gpr_potrf_debug_max_polls(0)
        gpu.gpr_potrf_set_polling(-1)
    # back on the polled schedule (probed again), same factor
    t = torch.tensor(A, device="cuda")
    assert int(dev.potrf_lower_(t).item()) == 0
    assert gpu.gpr_potrf_polling() == 1
    assert np.abs(t.cpu().numpy() - L).max() / np.abs(L).max() < 1e-13


@pytest.mark.parametrize("n,nrhs", [(1516, 1), (6063, 1), (2049, 5)])
def test_one_launch_solve_that_gives_up_waiting_is_finished_not_poisoned(gpu, n, nrhs):
    """The one-launch triangular solve (trsv_chain_kernel) with a poll budget of one: some workgroup gives up on its
    predecessor almost surely.  The caller must still get the solution (trsv_fixup_kernel finishes the substitution),
    the status must be GPR_OK, and the rescue must be counted -- round 3 returned GPR_OK with NaNs."""
    import torch
    from scipy.linalg import cholesky, solve_triangular
    from gpr_calculator_amd import device as dev
    rng = np.random.default_rng(n)
    L = np.ascontiguousarray(cholesky(_spd(n, 3), lower=True))       # scipy hands back column-major storage
    B = rng.standard_normal((n, nrhs))
    Ld = torch.tensor(L, device="cuda")
    dinv = dev.tri_inv64(Ld)
    want = {False: solve_triangular(L, B, lower=True), True: solve_triangular(L.T, B, lower=False)}
    for trans in (False, True):                                   # healthy run first: nothing to rescue
        before = gpu.gpr_trsv_chain_fallbacks()
        x = dev.trsm_lower_(Ld, torch.tensor(B, device="cuda"), trans=trans, dinv=dinv)
        torch.cuda.synchronize()
        assert np.abs(x.cpu().numpy() - want[trans]).max() / np.abs(want[trans]).max() < 1e-12
        assert gpu.gpr_trsv_chain_fallbacks() == before
    try:
        assert gpu.gpr_trsv_debug_max_polls(1) == 0
        rescued = 0
        for trans in (False, True):
            before = gpu.gpr_trsv_chain_fallbacks()
            x = dev.trsm_lower_(Ld, torch.tensor(B, device="cuda"), trans=trans, dinv=dinv)
            torch.cuda.synchronize()
            got = x.cpu().numpy()
            assert np.isfinite(got).all()
            assert np.abs(got - want[trans]).max() / np.abs(want[trans]).max() < 1e-12
            rescued += gpu.gpr_trsv_chain_fallbacks() - before
        assert rescued >= 1, "a poll budget of one did not make any workgroup give up: the path under test did not run"
    finally:
        gpu.gpr_trsv_debug_max_polls(0)


def test_two_host_threads_factor_at_once(gpu):
    """two factorisations with look-ahead enqueued from two host threads on two streams: they share the look-ahead
    context of the device (flag words, private streams) and are serialised by it -- both factors are right"""
    import threading
    import torch
    from scipy.linalg import cholesky
    from gpr_calculator_amd import device as dev
    mats = [_spd(1800, 21), _spd(2100, 22)]
    refs = [cholesky(a, lower=True) for a in mats]
    out, err = [None, None], []

    def work(k):
        try:
            with torch.cuda.stream(torch.cuda.Stream()):
                for _ in range(3):
                    t = torch.tensor(mats[k], device="cuda")
                    info = dev.potrf_lower_(t)
                    out[k] = (int(info.item()), t.cpu().numpy())
        except Exception as e:      # noqa: BLE001
            err.append(e)
    ts = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not err, err
    for k in range(2):
        assert out[k][0] == 0
        assert np.abs(out[k][1] - refs[k]).max() / np.abs(refs[k]).max() < 1e-13
