"""the one-launch neighbour build alone on the headline batch (64 structures of 32 atoms): time per call; GPR_NBR_DBG=1..4
ends the kernel after geometry / count / offsets / fill (nothing runs behind it here).  usage: python tools/nbr_bench.py"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np, torch
from gpr_calculator_amd import _lib, device as dev
from gpr_calculator_amd.synthetic import neb_images
L = _lib.lib()
b = neb_images(64, seed=100)
atom0 = np.concatenate(([0], np.cumsum([len(s) for s in b]))).astype(np.int32)
pos = np.ascontiguousarray(np.concatenate([s.positions for s in b]))
cells = np.ascontiguousarray(np.stack([np.asarray(s.cell, dtype=np.float64) for s in b]))
pbc = np.ascontiguousarray(np.stack([np.asarray(s.pbc, dtype=np.int32) for s in b]).astype(np.int32))
num = np.ascontiguousarray(np.concatenate([s.numbers for s in b]).astype(np.int32))
natoms = int(atom0[-1]); capP, capR = natoms * 48, natoms * 32
ri = torch.empty(2 * capR, dtype=torch.int32, device=dev.device())
def call():
    h, done = C.c_void_p(), C.c_int(0)
    rc = L.gpr_nbr_build_device_rows(C.byref(h), dev.stream_ptr(), 64, atom0.ctypes.data, pos.ctypes.data, cells.ctypes.data, pbc.ctypes.data,
                                     num.ctypes.data, 5.0, 0, capP, capR, ri.data_ptr(), ri.data_ptr() + 4 * capR, C.byref(done))
    assert rc == 0, L.gpr_last_error()
    L.gpr_nbr_destroy(h)
    return done.value
for _ in range(20): d = call()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
best = 1e9
for rep in range(5):
    e0.record()
    for _ in range(100): call()
    e1.record(); torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) / 100)
print(f"GPR_NBR_DBG={os.environ.get('GPR_NBR_DBG', '0')} NO_FUSED={os.environ.get('GPR_NBR_NO_FUSED', '')} rows_done={d}: {best * 1e3:.1f} us per build")
