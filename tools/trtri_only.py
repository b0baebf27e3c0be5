"""gpr_trtri_lower alone (N = 6063 by default): target of a rocprofv3 kernel trace.  usage: python tools/trtri_only.py [n] [reps]"""
import os, sys, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
from gpr_calculator_amd import device as dev
n = int(sys.argv[1]) if len(sys.argv) > 1 else 6063
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
d = dev.device()
B = torch.randn(n, max(n // 2, 4), dtype=torch.float64, device=d)
A = B @ B.T + torch.eye(n, dtype=torch.float64, device=d) * (0.01 * n + 1.0)
L = torch.linalg.cholesky(A).contiguous()
X = dev.trtri_lower(L); torch.cuda.synchronize()
ts = []
for _ in range(reps):
    t0 = time.perf_counter(); X = dev.trtri_lower(L); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
err = float((X @ L - torch.eye(n, dtype=torch.float64, device=d)).abs().max())
print(f"trtri n={n}: {min(ts) * 1e3:.3f} ms ({n ** 3 / 3 / min(ts) / 1e12:.1f} TF by n^3/3), |XL - I| = {err:.1e}")
