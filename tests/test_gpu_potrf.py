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
