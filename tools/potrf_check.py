"""Cholesky on the device against torch.linalg.cholesky, sizes around the block edges, and timing.
usage: python tools/potrf_check.py [n ...]"""
import os, sys, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
from gpr_calculator_amd import device as dev

sizes = [int(a) for a in sys.argv[1:]] or [1, 5, 16, 17, 64, 100, 255, 256, 257, 300, 466, 513, 1000, 1516, 6063]
torch.manual_seed(0)
d = dev.device()
ok = True
for n in sizes:
    B = torch.randn(n, max(n // 2, 4), dtype=torch.float64, device=d)
    A = B @ B.T + torch.eye(n, dtype=torch.float64, device=d) * (0.01 * n + 1.0)
    ref = torch.linalg.cholesky(A)
    L = A.clone()
    info, dinv = dev.potrf_lower_(L, want_dinv=True)
    torch.cuda.synchronize()
    err = float((L - ref).abs().max() / ref.abs().max())
    rec = float((L @ L.T - A).abs().max() / A.abs().max())
    # 64 x 64 diagonal-block inverses
    derr = 0.0
    for b in range((n + 63) // 64):
        k0, kb = b * 64, min(64, n - b * 64)
        W = dinv[b * 4096:(b + 1) * 4096].reshape(64, 64)[:kb, :kb]
        derr = max(derr, float((W @ L[k0:k0 + kb, k0:k0 + kb] - torch.eye(kb, dtype=torch.float64, device=d)).abs().max()))
    X = dev.trtri_lower(L)
    ierr = float((X @ L - torch.eye(n, dtype=torch.float64, device=d)).abs().max())
    terr = float(torch.triu(X, 1).abs().max())
    tt = []
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter(); dev.trtri_lower(L); torch.cuda.synchronize(); tt.append(time.perf_counter() - t0)
    to = []
    for _ in range(2):
        I = torch.eye(n, dtype=torch.float64, device=d); torch.cuda.synchronize(); t0 = time.perf_counter()
        dev.trsm_lower_(L, I, dinv=dinv); torch.cuda.synchronize(); to.append(time.perf_counter() - t0)
    derr = max(derr, 0.0 if (ierr < 1e-9 and terr == 0.0) else 1.0)
    ts = []
    for _ in range(5 if n > 256 else 2):
        L2 = A.clone(); torch.cuda.synchronize(); t0 = time.perf_counter()
        dev.potrf_lower_(L2, want_dinv=True); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    flag = "ok" if (err < 1e-11 and rec < 1e-12 and derr < 1e-9 and int(info.item()) == 0) else "FAIL"
    ok &= flag == "ok"
    print(f"n={n:6d} |L-ref|/|ref|={err:.2e} |LL^T-A|/|A|={rec:.2e} dinv={derr:.2e} info={int(info.item())} "
          f"{min(ts) * 1e3:8.3f} ms  {n ** 3 / 3 / min(ts) / 1e12:6.2f} TF  trtri |XL-I|={ierr:.1e} {min(tt) * 1e3:7.3f} ms (trsm on I: {min(to) * 1e3:7.3f} ms)  {flag}", flush=True)
# not positive definite: info = first bad leading minor
A = torch.eye(300, dtype=torch.float64, device=d); A[200, 200] = -1.0
info = dev.potrf_lower_(A)
print("indefinite: info =", int(info.item()), "(expect 201)")
ok &= int(info.item()) == 201
print("ALL OK" if ok else "FAILURES")
sys.exit(0 if ok else 1)
