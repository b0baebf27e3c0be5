"""Wall-clock split of one native batch prediction (Dot cfg3): gathering the structures' arrays, the C call (queues
everything), waiting for the device (polling the status words in page-locked memory), the rest (argument marshalling,
result slicing).  usage: python tools/native_split.py [batch]"""
import sys, os, time, collections
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np, torch
import bench
from gpr_calculator_amd import gaussianprocess as gpmod, _lib
from gpr_calculator_amd.synthetic import neb_images
from gpr_calculator_amd.parallel import Comm

gp = bench.build_model("Dot", 2000, Comm())
gp.fit(show=False, opt=False)
batch = neb_images(int(sys.argv[1]) if len(sys.argv) > 1 else 64, seed=100)
for _ in range(30): gp.predict_structures(batch, return_std=True)
acc = collections.defaultdict(float)

class Timed:
    def __init__(self, obj, names): self._o, self._n = obj, names
    def __getattr__(self, n):
        f = getattr(self._o, n)
        if n not in self._n: return f
        def w(*a, **k):
            t0 = time.perf_counter(); r = f(*a, **k); acc[self._n[n]] += time.perf_counter() - t0; return r
        return w

L = _lib.lib()
_lib.lib = lambda: Timed(L, {"gpr_predict_dot2_batch": "C call (queues the step)"})
if gpmod._pygather is not None:
    gpmod._pygather = Timed(gpmod._pygather, {"gather": "gather (csrc/pygather.c)"})
orig_isnan = np.isnan
def isnan(x):
    t0 = time.perf_counter(); r = orig_isnan(x); acc["polling for the device"] += time.perf_counter() - t0; return r
gpmod.np = Timed(np, {})          # keep numpy, but route isnan through the timer
gpmod.np.isnan = isnan
orig_native = gpmod.GP._predict_batch_native
def native(self, *a, **k):
    t0 = time.perf_counter(); r = orig_native(self, *a, **k); acc["native total"] += time.perf_counter() - t0; return r
gpmod.GP._predict_batch_native = native
N = 200
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(N): gp.predict_structures(batch, return_std=True)
torch.cuda.synchronize(); tot = (time.perf_counter() - t0) / N * 1e3
print(f"step {tot:.3f} ms (timers add a few microseconds)")
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]): print(f"  {k:32s} {v / N * 1e3:7.3f} ms")
