"""Triangular solves with a handful of right-hand sides (alpha = K^-1 y is one): the one-launch chained form against
torch.linalg.solve_triangular, sizes around the block edges, and its time against the launch-per-block form
(GPR_TRSV_NO_CHAIN=1).  usage: python tools/trsv_check.py [n ...]"""
import os, sys, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
from gpr_calculator_amd import device as dev

sizes = [int(a) for a in sys.argv[1:]] or [65, 128, 129, 200, 466, 1000, 1516, 6063, 24250]
torch.manual_seed(0)
d = dev.device()
ok = True
for n in sizes:
    B = torch.randn(n, max(n // 2, 4), dtype=torch.float64, device=d)
    A = B @ B.T + torch.eye(n, dtype=torch.float64, device=d) * (0.01 * n + 1.0)
    L = A.clone()
    info, dinv = dev.potrf_lower_(L, want_dinv=True)
    Lt = torch.tril(L)
    for nrhs in (1, 3, 8):
        y = torch.randn(n, nrhs, dtype=torch.float64, device=d)
        errs = []
        for trans in (False, True):
            x = dev.trsm_lower_(L, y.clone(), trans=trans, dinv=dinv)
            ref = torch.linalg.solve_triangular(Lt.T if trans else Lt, y, upper=trans)
            errs.append(float((x - ref).abs().max() / ref.abs().max()))
        ts = []
        for _ in range(5):
            z = y.clone(); torch.cuda.synchronize(); t0 = time.perf_counter()
            dev.trsm_lower_(L, z, trans=False, dinv=dinv); dev.trsm_lower_(L, z, trans=True, dinv=dinv)
            torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        good = max(errs) < 1e-11
        ok &= good
        print(f"n={n:6d} nrhs={nrhs}: |x - ref|/|ref| = {errs[0]:.1e} (L), {errs[1]:.1e} (L^T); both solves {min(ts) * 1e3:.3f} ms  {'ok' if good else 'FAIL'}")
print("ALL OK" if ok else "FAILED")
sys.exit(0 if ok else 1)
