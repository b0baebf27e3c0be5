"""One factorisation at the given size, three times (for rocprofv3 --kernel-trace --stats): usage python tools/potrf_only.py [n=24250]"""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
from gpr_calculator_amd import device as dev
n = int(sys.argv[1]) if len(sys.argv) > 1 else 24250
d = dev.device()
torch.manual_seed(0)
B = torch.randn(n, 64, dtype=torch.float64, device=d)
A = B @ B.T + torch.eye(n, dtype=torch.float64, device=d) * (0.01 * n + 1.0)
del B
L = torch.empty_like(A)
for _ in range(3):
    L.copy_(A)
    info, dinv = dev.potrf_lower_(L, want_dinv=True)
    torch.cuda.synchronize()
print("info", int(info.item()))
