"""Host-side inclusive wall time per function of one predict step (no device synchronisation inside
the step, so device time only shows up where the host actually waits).  usage: python tools/host_timeline.py [Dot|RBF]"""
import sys, os, time, collections
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np, torch
import bench
from gpr_calculator_amd import device as dev, SO3 as so3mod, gaussianprocess as gpmod
from gpr_calculator_amd.synthetic import neb_images
from gpr_calculator_amd.parallel import Comm

acc, cnt = collections.defaultdict(float), collections.Counter()

def wrap(obj, name, label=None):
    fn = getattr(obj, name)
    label = label or f"{getattr(obj, '__name__', obj)}.{name}"
    def w(*a, **k):
        t0 = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            acc[label] += time.perf_counter() - t0; cnt[label] += 1
    setattr(obj, name, w)

kind = sys.argv[1] if len(sys.argv) > 1 else "Dot"
gp = bench.build_model(kind, 2000, Comm())
gp.fit(show=False, opt=False)
batch = neb_images(64, seed=100)
for _ in range(3): gp.predict_structures(batch, return_std=True)
S = so3mod.SO3
for n in ("neighbor_index_batch", "calculate_batch_dev"): wrap(S, n, "SO3." + n)
for n in ("_batch_data_dev", "_predict_dev", "get_train_x", "_get_fixed_atoms"): wrap(gpmod.GP, n, "GP." + n)
wrap(dev.PackedSet, "__init__", "PackedSet.__init__"); wrap(dev.PackedSet, "close", "PackedSet.close")
for n in ("to_dev", "dot2_features", "gemm", "rowdot", "diag_ff", "diag_ee", "gemv_t", "trmm_lower", "colsumsq", "block_ff", "block_ef", "block_ee"):
    wrap(dev, n, "dev." + n)
wrap(type(gp.kernel), "diag_dev", "kernel.diag_dev"); wrap(type(gp.kernel), "_psi", "kernel._psi") if hasattr(gp.kernel, "_psi") else None
wrap(torch.Tensor, "cpu", "Tensor.cpu"); wrap(torch.Tensor, "index_select", "Tensor.index_select")
wrap(np, "argsort", "np.argsort")
N = 20
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(N): gp.predict_structures(batch, return_std=True)
torch.cuda.synchronize(); tot = (time.perf_counter() - t0) / N * 1e3
print(f"step {tot:.3f} ms")
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print(f"{k:32s} {v / N * 1e3:8.3f} ms  x{cnt[k] / N:.0f}")
