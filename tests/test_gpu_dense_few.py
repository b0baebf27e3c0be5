"""Few-column forms of the dense reductions around a single structure's prediction (1 + 3 n quantities): the fused
|L^-1 k*|^2 kernel (gpr_trmm_lower_sumsq), gpr_gemv_t and gpr_colsumsq -- against torch FP64 products of the same
operands, tolerance 1e-12 relative (sums of n products in a different order), and against the wide two-call path."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,cols", [(64, 1), (100, 5), (777, 13), (1516, 61), (2050, 17), (2050, 33), (2049, 64), (1000, 97), (900, 256)])
def test_fused_variance_term_matches_the_product(n, cols):
    from gpr_calculator_amd import device as dev
    torch.manual_seed(n + cols)
    d = dev.device()
    T = torch.tril(torch.randn(n, n, dtype=torch.float64, device=d)) / n ** 0.5
    T += torch.triu(torch.full((n, n), float("nan"), dtype=torch.float64, device=d), 1)        # the upper triangle must never be read
    B = torch.randn(n, cols, dtype=torch.float64, device=d)
    ref = ((torch.tril(T) @ B) ** 2).sum(0)
    got = dev.trmm_lower_sumsq(T, B)
    assert torch.isfinite(got).all()
    assert float(((got - ref).abs() / ref).max()) < 1e-12
    # a strided B (a column block of a wider matrix) and a strided T (leading dimension > n)
    wide = torch.randn(n, cols + 7, dtype=torch.float64, device=d)
    Bv = wide[:, 3:3 + cols]
    Tbig = torch.zeros(n, n + 5, dtype=torch.float64, device=d)
    Tbig[:, :n] = torch.tril(T)
    got2 = dev.trmm_lower_sumsq(Tbig[:, :n], Bv)
    ref2 = ((torch.tril(T) @ Bv) ** 2).sum(0)
    assert float(((got2 - ref2).abs() / ref2).max()) < 1e-12


def test_small_and_wide_operands_take_the_two_call_path():
    from gpr_calculator_amd import device as dev
    torch.manual_seed(5)
    d = dev.device()
    for n, cols in ((20, 4), (63, 61), (500, 300)):
        T = torch.tril(torch.randn(n, n, dtype=torch.float64, device=d))
        B = torch.randn(n, cols, dtype=torch.float64, device=d)
        ref = ((T @ B) ** 2).sum(0)
        got = dev.trmm_lower_sumsq(T, B)
        assert float(((got - ref).abs() / ref).max()) < 1e-12


@pytest.mark.parametrize("n,cols", [(1, 1), (37, 3), (700, 61), (6063, 61), (1200, 256), (1200, 257), (333, 1000)])
def test_column_reductions_few_and_wide(n, cols):
    from gpr_calculator_amd import device as dev
    torch.manual_seed(n * 7 + cols)
    d = dev.device()
    A = torch.randn(n, cols, dtype=torch.float64, device=d)
    x = torch.randn(n, dtype=torch.float64, device=d)
    gv, ref = dev.gemv_t(A, x), A.T @ x
    assert float((gv - ref).abs().max()) <= 1e-12 * max(1.0, float(ref.abs().max())) * n ** 0.5
    cs, ref2 = dev.colsumsq(A), (A * A).sum(0)
    assert float(((cs - ref2).abs() / ref2).max()) < 1e-12
    # deterministic: the same call twice gives the same bits (fixed summation order)
    assert torch.equal(dev.gemv_t(A, x), gv) and torch.equal(dev.colsumsq(A), cs)


@pytest.mark.parametrize("ta,tb", [(False, True), (False, False), (True, False), (True, True)])
@pytest.mark.parametrize("m,n,k", [(300, 333, 16), (300, 333, 250), (257, 256, 257), (1000, 640, 1024), (129, 511, 40)])
def test_tiled_product_reads_nothing_outside_its_operands(m, n, k, ta, tb):
    """C <- beta C + alpha op(A) op(B) through the 128 x 128-tile kernel in its predicate-free form (gemm128_kernel<.., .., true>;
    the NT form is the trailing update of gaussianprocess.py:175's Cholesky, scipy there, the others carry the triangular inverse
    and the Gram products of the fit): operands are views into NaN-filled arrays, so a read beyond the k range, or beyond the
    last row or column, that reached an accumulator would show; alpha = -1, beta = 1 starts the accumulators from C, the other
    combinations go through the plain and the read-modify-write epilogue; 1e-13 relative to torch"""
    from gpr_calculator_amd import device as dev
    torch.manual_seed(m + n + k)
    d = dev.device()
    nan = float("nan")

    def operand(rows, cols):
        big = torch.full((rows + 3, cols + 5), nan, dtype=torch.float64, device=d)
        big[:rows, :cols] = torch.randn(rows, cols, dtype=torch.float64, device=d)
        return big[:rows, :cols]

    A = operand(k, m) if ta else operand(m, k)
    B = operand(n, k) if tb else operand(k, n)
    opA = A.contiguous().T if ta else A.contiguous()
    opB = B.contiguous().T if tb else B.contiguous()
    C0 = torch.randn(m, n, dtype=torch.float64, device=d)
    for alpha, beta in ((-1.0, 1.0), (1.0, 0.0), (0.5, -2.0), (1.0, 0.25)):
        ref = alpha * (opA @ opB) + beta * C0
        out = dev.gemm(A, B, transa=ta, transb=tb, alpha=alpha, beta=beta, out=C0.clone())
        assert torch.isfinite(out).all()
        assert float((out - ref).abs().max() / ref.abs().max()) < 1e-13


@pytest.mark.parametrize("m,k", [(1, 64), (5, 100), (97, 466), (1000, 63), (6063, 2500)])
def test_product_with_one_right_hand_side(m, k):
    """gpr_gemm with n = 1 and A not transposed takes the row-per-wavefront matrix-vector kernel (k >= 64; below that the tiled
    one): against torch, 1e-13 relative to |A| |x|; strided A and x (a column of a wider matrix), beta != 0, and the [1, k] row
    form of x"""
    from gpr_calculator_amd import device as dev
    torch.manual_seed(m + k)
    d = dev.device()
    big = torch.randn(m, k + 3, dtype=torch.float64, device=d)
    A = big[:, :k]
    xs = torch.randn(k, 4, dtype=torch.float64, device=d)
    x = xs[:, 2:3]
    scale = float((A.abs() @ x.abs()).max())
    ref = A @ x
    out = dev.gemm(A, x)
    assert float((out - ref).abs().max()) < 1e-13 * scale
    c0 = torch.randn(m, 1, dtype=torch.float64, device=d)
    out2 = dev.gemm(A, x, alpha=-0.5, beta=2.0, out=c0.clone())
    assert float((out2 - (-0.5 * ref + 2.0 * c0)).abs().max()) < 1e-13 * (scale + float(c0.abs().max()))
    xr = x.reshape(1, k).contiguous()
    out3 = dev.gemm(A, xr, transb=True)
    assert float((out3 - ref).abs().max()) < 1e-13 * scale
