"""Wall-clock split of one native batch prediction (Dot cfg3): host preparation, the C call (queues everything; one
mid-call wait for the pair count), the result download (waits for the device), the result slicing.
usage: python tools/native_split.py"""
import sys, os, time, collections
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np, torch
import bench
from gpr_calculator_amd import gaussianprocess as gpmod, neighbors, _lib
from gpr_calculator_amd.synthetic import neb_images
from gpr_calculator_amd.parallel import Comm

gp = bench.build_model("Dot", 2000, Comm())
gp.fit(show=False, opt=False)
batch = neb_images(64, seed=100)
for _ in range(5): gp.predict_structures(batch, return_std=True)
acc = collections.defaultdict(float)
def wrap(obj, name, label):
    fn = getattr(obj, name)
    def w(*a, **k):
        t0 = time.perf_counter()
        try: return fn(*a, **k)
        finally: acc[label] += time.perf_counter() - t0
    setattr(obj, name, w)
wrap(gpmod.GP, "_predict_batch_native", "native total")
orig_ba = neighbors.batch_arrays
wrap(neighbors, "batch_arrays", "batch_arrays")
wrap(torch.Tensor, "cpu", "out.cpu (device wait + D2H)")
L = _lib.lib()
cfn = L.gpr_predict_dot2_batch
class Shim:
    def __getattr__(self, n):
        if n == "gpr_predict_dot2_batch":
            def w(*a):
                t0 = time.perf_counter(); r = cfn(*a); acc["C call"] += time.perf_counter() - t0; return r
            return w
        return getattr(L, n)
shim = Shim()
_lib.lib = lambda: shim
N = 50
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(N): gp.predict_structures(batch, return_std=True)
torch.cuda.synchronize(); tot = (time.perf_counter() - t0) / N * 1e3
print(f"step {tot:.3f} ms")
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]): print(f"  {k:32s} {v / N * 1e3:7.3f} ms")
