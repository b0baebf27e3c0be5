"""GPU parity of the L1 covariance blocks, kernel objects and FFI-glue mirrors against the oracle's
restatement of the reference wrappers (oracle/gp_numpy.py on top of the plain-C natives).
Tolerance 1e-12 relative to the largest magnitude of each matrix."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
REL = 1e-12


def rel(a, b):
    return np.abs(np.asarray(a) - np.asarray(b)).max() / (np.abs(np.asarray(b)).max() + 1e-300)


def make_data(seed, nE, nF, d=30, species=(8, 46), rows_e=(3, 9), rows_f=(1, 50), zero_row=False):
    rng = np.random.default_rng(seed)
    energy, force = [], []
    for _ in range(nE):
        n = int(rng.integers(*rows_e))
        energy.append((rng.random((n, d)), rng.choice(species, n)))
    for _ in range(nF):
        n = int(rng.integers(*rows_f))
        force.append((rng.random((n, d)), rng.standard_normal((n, d, 3)), rng.choice(species, n)))
    if zero_row and force:
        force[0][0][0, :] = 0.0
    return {"energy": energy, "force": force}


def tupled(data):
    from oracle.gp_numpy import list_to_tuple
    out = {}
    if "energy" in data:
        out["energy"] = list_to_tuple(data["energy"], mode="energy")
    if "force" in data:
        out["force"] = list_to_tuple(data["force"])
    return out


@pytest.fixture(scope="module")
def gpu():
    from gpr_calculator_amd import _lib
    _lib.require_gpu()
    return _lib.lib()


@pytest.mark.parametrize("zeta", [2.0, 3.0])
def test_rbf_kernel_object_matches_oracle(gpu, port_natives, zeta):
    from gpr_calculator_amd.kernels.RBF_mb import RBF_mb
    from oracle.gp_numpy import OracleRBF
    train = tupled(make_data(1, 5, 23, zero_row=True))
    new = make_data(2, 2, 7)
    k = RBF_mb(para=[2.5, 1.3], zeta=zeta)
    o = OracleRBF(para=(2.5, 1.3), zeta=zeta, natives=port_natives)
    K, Ko = k.k_total(train), o.k_total(train)
    assert K.shape == Ko.shape and rel(K, Ko) < REL
    assert rel(K, K.T) < 1e-13       # off-diagonal bins are mirrored exactly; 3x3 self blocks to rounding
    Kt, Kto = k.k_total(new, train, f_tol=1e-8), o.k_total(new, train, f_tol=1e-8)
    assert rel(Kt, Kto) < REL
    # transposed device layout used by predict
    assert rel(k.k_total_dev(new, train, 1e-8, transpose=True).cpu().numpy().T, Kto) < REL
    Kg, dK = k.k_total_with_grad(train)
    Kgo, dKo = o.k_total_with_grad(train)
    assert rel(Kg, Kgo) < REL and rel(dK[:, :, 0], dKo[:, :, 0]) < REL and rel(dK[:, :, 1], dKo[:, :, 1]) < REL
    assert rel(k.diag(new), o.diag(new)) < REL
    # only-force / only-energy dictionaries
    for sub in ({"force": new["force"]}, {"energy": new["energy"]}):
        assert rel(k.k_total(sub, train), o.k_total(sub, train)) < REL
        assert rel(k.diag(sub), o.diag(sub)) < REL


@pytest.mark.parametrize("zeta", [2.0, 3.0])
def test_dot_kernel_object_matches_oracle(gpu, port_natives, zeta):
    from gpr_calculator_amd.kernels.Dot_mb import Dot_mb
    from oracle.gp_numpy import OracleDot
    train = tupled(make_data(3, 4, 19))
    new = make_data(4, 2, 6)
    k = Dot_mb(para=[2.0, 2.0], zeta=zeta)
    o = OracleDot(para=(2.0, 2.0), zeta=zeta, natives=port_natives)
    assert rel(k.k_total(train), o.k_total(train)) < REL          # includes the zeta=2 quirk of ef/ff
    assert rel(k.k_total(new, train), o.k_total(new, train)) < REL
    assert rel(k.k_total_dev(new, train, transpose=True).cpu().numpy().T, o.k_total(new, train)) < REL
    if zeta == 2.0:      # the explicit feature map (GEMM) and the row-pair MFMA blocks give the same matrix
        k.use_feature_map = False
        assert rel(k.k_total(train), o.k_total(train)) < REL and rel(k.k_total(new, train), o.k_total(new, train)) < REL
        k.use_feature_map = True
        for sub in ({"force": new["force"]}, {"energy": new["energy"]}):
            assert rel(k.k_total(sub, train), o.k_total(sub, train)) < REL
    Kg, dK = k.k_total_with_grad(train)
    Kgo, dKo = o.k_total_with_grad(train)
    assert rel(Kg, Kgo) < REL and rel(dK[:, :, 0], dKo[:, :, 0]) < REL
    assert np.array_equal(dK[:, :, 1], dKo[:, :, 1])
    assert rel(k.diag(new), o.diag(new)) < 1e-11                   # numpy-eps convention on both sides


def test_ffi_glue_mirrors(gpu, port_natives):
    from gpr_calculator_amd.kernels import rbf_kernel, dot_kernel
    from oracle.gp_numpy import Wrappers, list_to_tuple
    w = Wrappers(port_natives)
    rng = np.random.default_rng(9)
    d = 30
    e1 = [(rng.random((n, d)), rng.choice((8, 46), n)) for n in (4, 7, 2)]
    e2 = list_to_tuple([(rng.random((n, d)), rng.choice((8, 46), n)) for n in (5, 3)], mode="energy")
    f2 = list_to_tuple([(rng.random((n, d)), rng.standard_normal((n, d, 3)), rng.choice((8, 46), n)) for n in (17, 4, 33)])
    f9 = [(rng.random((n, d)), rng.standard_normal((n, d, 9)), rng.choice((8, 46), n)) for n in (6, 21)]
    assert rel(rbf_kernel.kee_C(e1, e2, 2.5, 1.3, 2.0), w.rbf_kee(e1, e2, 2.5, 1.3, 2.0)) < REL
    for a, b in zip(rbf_kernel.kee_C(e1, e2, 2.5, 1.3, 2.0, grad=True), w.rbf_kee(e1, e2, 2.5, 1.3, 2.0, grad=True)):
        assert rel(a, b) < REL
    assert rel(rbf_kernel.kef_C(e1, f2, 2.5, 1.3, 2.0, transpose=True), w.rbf_kef(e1, f2, 2.5, 1.3, 2.0, transpose=True)) < REL
    for a, b in zip(rbf_kernel.kef_C(e1, f9, 2.5, 1.3, 2.0, stress=True, transpose=True),
                    w.rbf_kef(e1, f9, 2.5, 1.3, 2.0, stress=True, transpose=True)):
        assert a.shape == b.shape and rel(a, b) < REL
    for a, b in zip(rbf_kernel.kff_C(f9, f2, 2.5, 1.3, 2.0, stress=True, tol=1e-10),
                    w.rbf_kff(f9, f2, 2.5, 1.3, 2.0, stress=True, tol=1e-10)):
        assert a.shape == b.shape and rel(a, b) < REL
    assert rel(dot_kernel.kee_C(e1, e2, 2.0, 1.5, 3.0), w.dot_kee(e1, e2, 2.0, 1.5, 3.0)) < REL
    for a, b in zip(dot_kernel.kef_C(e1, f9, 2.0, 1.5, 2.0, stress=True), w.dot_kef(e1, f9, 2.0, 1.5, 2.0, stress=True)):
        assert rel(a, b) < REL
    for a, b in zip(dot_kernel.kff_C(f9, f2, 2.0, 1.5, 2.0, stress=True), w.dot_kff(f9, f2, 2.0, 1.5, 2.0, stress=True)):
        assert rel(a, b) < REL


def test_dense_linear_algebra(gpu):
    import torch
    from scipy.linalg import cholesky, solve_triangular
    from gpr_calculator_amd import device as dev
    rng = np.random.default_rng(0)
    for n in (1, 37, 64, 200, 333):
        a = rng.standard_normal((n, n))
        spd = a @ a.T + n * np.eye(n)
        t = torch.tensor(spd, device="cuda")
        info = dev.potrf_lower_(t)
        L = cholesky(spd, lower=True)
        assert int(info.item()) == 0 and rel(t.cpu().numpy(), L) < 1e-13
        b = rng.standard_normal((n, 45))
        t2 = torch.tensor(spd, device="cuda")
        info2, dinv = dev.potrf_lower_(t2, want_dinv=True)
        for trans in (False, True):
            ref = solve_triangular(L, b, lower=True, trans=1 if trans else 0)
            x = dev.trsm_lower_(t, torch.tensor(b, device="cuda"), trans=trans).cpu().numpy()
            assert rel(x, ref) < 1e-11
            x = dev.trsm_lower_(t2, torch.tensor(b, device="cuda"), trans=trans, dinv=dinv).cpu().numpy()
            assert rel(x, ref) < 1e-11
        assert rel(dev.colsumsq(torch.tensor(b, device="cuda")).cpu().numpy(), (b * b).sum(0)) < 1e-14
        assert abs(dev.sum_log_diag(t).item() - np.log(np.diag(L)).sum()) < 1e-12 * max(1, n)
    A, B = rng.standard_normal((70, 33)), rng.standard_normal((33, 51))
    tA, tB = torch.tensor(A, device="cuda"), torch.tensor(B, device="cuda")
    assert rel(dev.gemm(tA, tB).cpu().numpy(), A @ B) < 1e-14
    assert rel(dev.gemm(tB, tA, transa=True, transb=True).cpu().numpy(), B.T @ A.T) < 1e-14
    # the 128x128-tile kernel (m, n >= 256), ragged sizes, all transposes, beta and the triangular product
    m, n, k = 300, 333, 257
    for ta in (False, True):
        for tb in (False, True):
            A = rng.standard_normal((k, m) if ta else (m, k)); B = rng.standard_normal((n, k) if tb else (k, n))
            C0 = rng.standard_normal((m, n))
            ref = 0.5 * (A.T if ta else A) @ (B.T if tb else B) - 2.0 * C0
            out = dev.gemm(torch.tensor(A, device="cuda"), torch.tensor(B, device="cuda"), transa=ta, transb=tb,
                           alpha=0.5, beta=-2.0, out=torch.tensor(C0, device="cuda"))
            assert rel(out.cpu().numpy(), ref) < 1e-13
    T = np.tril(rng.standard_normal((420, 420))); Bm = rng.standard_normal((420, 301))
    Td = T.copy()                          # columns beyond a row block's own 128-wide diagonal block are never read
    for r0 in range(0, 420, 128):
        Td[r0:r0 + 128, r0 + 128:] = np.nan
    dirty = torch.tensor(Td, device="cuda")
    assert rel(dev.trmm_lower(dirty, torch.tensor(Bm, device="cuda")).cpu().numpy(), T @ Bm) < 1e-13
    a = rng.standard_normal((700, 700)); spd = a @ a.T + 700 * np.eye(700)
    t = torch.tensor(spd, device="cuda")
    assert int(dev.potrf_lower_(t).item()) == 0 and rel(t.cpu().numpy(), cholesky(spd, lower=True)) < 1e-12
    bad = torch.tensor(np.array([[1.0, 2.0], [2.0, 1.0]]), device="cuda")
    assert int(dev.potrf_lower_(bad).item()) == 2          # not positive definite: pivot 2


# shapes: one workgroup; the short-chain form for few rows (up to 32 row blocks; n = 1864: four operand batches per wavefront);
# the staged split form (33..64 row blocks); the row-resident and the paired forms for many rows
@pytest.mark.parametrize("m,n", [(1, 1), (5, 3), (33, 130), (100, 257), (70, 466), (200, 932), (97, 1864), (1024, 131),
                                 (1200, 466), (2048, 257), (2100, 466), (2100, 932)])
def test_ws_tail_matches_numpy(m, n):
    """gpr_ws_tail (fused weight-space tail: P w and diag - vsq (|p|^2 - |p C^-T|^2) from MFMA tiles that are never
    stored) against numpy, at sizes that are not multiples of its 32 x 128 x 32 tiles, and against the two-call route
    gpr_gemm_nt_lower + gpr_ws_finish"""
    import torch
    from gpr_calculator_amd import _lib, device as dev
    rng = np.random.default_rng(m * 1000 + n)
    P = rng.normal(size=(m, n))
    Ci = np.tril(rng.normal(size=(n, n))) / np.sqrt(n)
    w = rng.normal(size=n)
    vsq = 1.7
    Q = P @ Ci.T
    core = (P * P).sum(1) - (Q * Q).sum(1)
    diag = vsq * core + rng.uniform(-0.5, 2.0, size=m)              # some rows clamp at zero
    want_mean, want_var = P @ w, np.maximum(diag - vsq * core, 0.0)
    dP, dC, dw, dd = dev.to_dev(P), dev.to_dev(Ci), dev.to_dev(w), dev.to_dev(diag)
    out = torch.full((2 * m,), np.nan, dtype=torch.float64, device="cuda")
    dev.check(_lib.lib().gpr_ws_tail(dev.stream_ptr(), m, n, dev.ptr(dP), n, dev.ptr(dC), n, dev.ptr(dw), dev.ptr(dd),
                                     dev.ptr(out), vsq), "gpr_ws_tail")
    got = out.cpu().numpy()
    scale = max(1.0, np.abs(core).max() * vsq)
    assert np.abs(got[:m] - want_mean).max() < 1e-12 * max(1.0, np.abs(want_mean).max())
    assert np.abs(got[m:] - want_var).max() < 1e-12 * scale
    # the two-call route computes the same numbers
    dQ = torch.empty((m, n), dtype=torch.float64, device="cuda")
    dev.check(_lib.lib().gpr_gemm_nt_lower(dev.stream_ptr(), m, n, 1.0, dev.ptr(dP), n, dev.ptr(dC), n, dev.ptr(dQ), n), "gemm")
    out2 = torch.empty(2 * m, dtype=torch.float64, device="cuda")
    dev.check(_lib.lib().gpr_ws_finish(dev.stream_ptr(), m, n, dev.ptr(dP), n, dev.ptr(dQ), n, dev.ptr(dw), dev.ptr(dd),
                                       dev.ptr(out2), vsq), "finish")
    assert np.abs(out2.cpu().numpy() - got).max() < 1e-12 * scale


@pytest.mark.parametrize("n", [65, 129, 200, 466, 1516])
def test_few_rhs_triangular_solves_in_one_launch(n):
    """gpr_trsm_lower_dinv with <= 8 right-hand sides (alpha = K^-1 y): the chained one-launch form against
    torch.linalg.solve_triangular, both directions, sizes around the 64-row block edges; a NaN in the data ends as
    non-finite numbers in the rows that depend on it, not as a wait"""
    import torch
    from gpr_calculator_amd import device as dev
    torch.manual_seed(n)
    d = dev.device()
    B = torch.randn(n, max(n // 2, 4), dtype=torch.float64, device=d)
    A = B @ B.T + torch.eye(n, dtype=torch.float64, device=d) * (0.01 * n + 1.0)
    L = A.clone()
    info, dinv = dev.potrf_lower_(L, want_dinv=True)
    assert int(info.item()) == 0
    Lt = torch.tril(L)
    for nrhs in (1, 3, 8):
        y = torch.randn(n, nrhs, dtype=torch.float64, device=d)
        for trans in (False, True):
            x = dev.trsm_lower_(L, y.clone(), trans=trans, dinv=dinv)
            ref = torch.linalg.solve_triangular(Lt.T if trans else Lt, y, upper=trans)
            assert float((x - ref).abs().max() / ref.abs().max()) < 1e-12
    y = torch.randn(n, 2, dtype=torch.float64, device=d)
    y[n // 3, 0] = float("nan")
    x = dev.trsm_lower_(L, y.clone(), trans=False, dinv=dinv).cpu().numpy()
    assert np.isfinite(x[:, 1]).all()                      # the other right-hand side is untouched by it
    assert not np.isfinite(x[n // 3:, 0]).all() and np.isfinite(x[:(n // 3 // 64) * 64, 0]).all()
