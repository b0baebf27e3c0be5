"""One predict-shaped force-force block launch (cfg3 shapes: 64x32 points of 28 rows vs 2000 points of 28 rows)
repeated a few times -- the target of rocprofv3 --pmc runs.  usage: python tools/kff_only.py [reps] [rbf|dot]"""
import sys, os
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np
import torch
from gpr_calculator_amd import device as dev, _lib

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
kind = sys.argv[2] if len(sys.argv) > 2 else "dot"
rng = np.random.default_rng(0)
d, rpp = 30, 28
def mk(npts):
    R = npts * rpp
    return rng.random((R, d)), rng.standard_normal((R, d, 3)), np.full(R, 46, np.int32), np.full(npts, rpp, np.int32)
a = mk(2048); b = mk(2000)
s1 = dev.PackedSet(_lib.ROLE_STREAM, *a)
s2 = dev.PackedSet(_lib.ROLE_RESIDENT, *b)
kp = dev.kernel_params(_lib.KERNEL_DOT, 2.0, 2.0, sigma0=2.0) if kind == "dot" else dev.kernel_params(_lib.KERNEL_RBF, 2.0, 2.5, 1.3, tol=1e-8)
out = torch.empty((3 * 2000, 3 * 2048), dtype=torch.float64, device="cuda")
dev.profile_enable(True)
for _ in range(reps):
    dev.block_ff(kp, s1, s2, out=out, transpose=True)
torch.cuda.synchronize()
ms, n = dev.profile_fetch()["ff"]
pairs = 2048 * rpp * 2000 * rpp
print(f"kff {kind}: {ms/n:.3f} ms/launch, {pairs/(ms/n*1e-3)/1e9:.2f} Gpairs/s, {pairs*1.1e3/(ms/n*1e-3)/1e12:.2f} TF")
