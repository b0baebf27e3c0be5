"""The subtracted term of the predictive variance for ONE structure, |L^-1 k*|^2 per predicted quantity: the fused few-column
kernel (gpr_trmm_lower_sumsq) against gpr_trmm_lower + gpr_colsumsq, and the predictive-mean reduction gpr_gemv_t, at the
sizes of BASELINE cfg2 / cfg3.  Times are HIP-event averages over 50 calls; the check is against torch in FP64.
usage: python tools/variance_few.py"""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
from gpr_calculator_amd import device as dev


def timed(f, reps=50):
    for _ in range(5):
        f()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(reps):
        f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


torch.manual_seed(0)
d = dev.device()
print(f"{'n':>6s} {'cols':>5s} {'fused us':>9s} {'trmm+colsumsq us':>17s} {'T bytes/fused GB/s':>19s} {'gemv_t us':>10s} {'max rel err':>12s}")
for n, cols in ((1516, 61), (6063, 1), (6064, 1), (6144, 1), (6063, 16), (6063, 61), (6063, 64), (6064, 64), (6144, 64), (6063, 97), (6063, 193), (6063, 256), (6063, 300), (8192, 61), (8192, 64), (777, 13)):
    T = torch.tril(torch.randn(n, n, dtype=torch.float64, device=d)) / n ** 0.5
    B = torch.randn(n, cols, dtype=torch.float64, device=d)
    x = torch.randn(n, dtype=torch.float64, device=d)
    ref = ((T @ B) ** 2).sum(0)
    got = dev.trmm_lower_sumsq(T, B)
    two = dev.colsumsq(dev.trmm_lower(T, B))
    err = float(((got - ref).abs() / ref).max())
    err2 = float(((two - ref).abs() / ref).max())
    gv = dev.gemv_t(B, x)
    errg = float((gv - B.T @ x).abs().max() / (B.T @ x).abs().max())
    assert err < 1e-12 and err2 < 1e-12 and errg < 1e-12, (n, cols, err, err2, errg)
    tf = timed(lambda: dev.trmm_lower_sumsq(T, B))
    tt = timed(lambda: dev.colsumsq(dev.trmm_lower(T, B)))
    tg = timed(lambda: dev.gemv_t(B, x))
    print(f"{n:6d} {cols:5d} {tf:9.1f} {tt:17.1f} {n * n * 4 / tf / 1e3:19.1f} {tg:10.1f} {max(err, err2, errg):12.2e}")
