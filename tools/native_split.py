"""Host-side timeline of the native batch prediction (Dot cfg3): where the time between two device runs goes.
Uses the timestamp hooks of GP._predict_batch_native (GP._trace).  usage: python tools/native_split.py [batch]"""
import sys, os, time, collections
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np, torch
import bench
from gpr_calculator_amd.synthetic import neb_images
from gpr_calculator_amd.parallel import Comm

gp = bench.build_model("Dot", 2000, Comm())
gp.fit(show=False, opt=False)
batch = neb_images(int(sys.argv[1]) if len(sys.argv) > 1 else 64, seed=100)
for _ in range(30): gp.predict_structures(batch, return_std=True)
N = 300
gp._trace = tr = []
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(N):
    tr.append(("call", time.perf_counter_ns()))
    gp.predict_structures(batch, return_std=True)
    tr.append(("return", time.perf_counter_ns()))
torch.cuda.synchronize(); tot = (time.perf_counter() - t0) / N * 1e3
gp._trace = None
seg = collections.defaultdict(float)
for (a, ta), (b, tb) in zip(tr[:-1], tr[1:]):
    seg[f"{a} -> {b}"] += (tb - ta) * 1e-3
print(f"step {tot:.3f} ms")
for k, v in seg.items(): print(f"  {k:42s} {v / N:8.1f} us")
