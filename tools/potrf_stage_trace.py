"""Phase stamps of one 256 x 256 diagonal block of the Cholesky (debug instance of potrf_diag256_kernel).
usage: python tools/potrf_stage_trace.py"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np, torch
from gpr_calculator_amd import device as dev, _lib

lib = _lib.lib()
lib.gpr_potrf_diag_trace.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_longlong, C.c_void_p]
d = dev.device()
B = torch.randn(256, 128, dtype=torch.float64, device=d)
A = B @ B.T + torch.eye(256, dtype=torch.float64, device=d) * 4
tr = np.zeros((2, 16, 8), dtype=np.int64)
for _ in range(3):
    L = A.clone()
    rc = lib.gpr_potrf_diag_trace(dev.stream_ptr(), 256, L.data_ptr(), 256, tr.ctypes.data)
assert rc == 0
print("err", float((torch.tril(L) - torch.linalg.cholesky(A)).abs().max()))
t0 = tr[0, 0, 0]
f, t = (tr[0] - t0) * 10, (tr[1] - t0) * 10       # ns
print("factor wave: stage | B1 done | +factor | +W->LDS | +B2 | +stores | +B3   (ns, deltas)")
for k in range(16):
    r = f[k]
    print(f"  {k:2d} at {r[0]:7d} | {r[1]-r[0]:5d} {r[2]-r[1]:5d} {r[3]-r[2]:5d} {r[4]-r[3]:5d} {r[5]-r[4]:5d}   stage {f[k+1][0]-r[0] if k<15 else 0:5d}")
print("tile wave 0: stage | wait B1 | trailing | wait B2 | panel | wait B3 | diag update+handover")
for k in range(16):
    r = t[k]
    nxt = t[k + 1][0] if k < 15 else r[5]
    print(f"  {k:2d} at {r[0]:7d} | {r[1]-r[0]:5d} {r[2]-r[1]:5d} {r[3]-r[2]:5d} {r[4]-r[3]:5d} {r[5]-r[4]:5d} {nxt-r[5]:5d}")
print("total us", (f[15][5]) / 1e3)
