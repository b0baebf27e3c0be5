"""Latency of predicting ONE structure (what an ASE optimiser sees per calculator call: GPR.calculate -> predict_structure),
Dot cfg3 model, with the kernels' own times beside it.
usage: python tools/single_latency.py [batch=1] [reps=300]"""
import sys, os, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np, torch
import bench
from gpr_calculator_amd import device as dev
from gpr_calculator_amd.synthetic import neb_images
from gpr_calculator_amd.parallel import Comm

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 1
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
gp = bench.build_model("Dot", 2000, Comm())
gp.fit(show=False, opt=False)
batch = neb_images(nb, seed=100)
for _ in range(30): gp.predict_structures(batch, return_std=True)
torch.cuda.synchronize()
lat = []
for _ in range(reps):
    a = time.perf_counter_ns(); gp.predict_structures(batch, return_std=True); lat.append(time.perf_counter_ns() - a)
lat = np.array(lat) * 1e-3
dev.profile_enable(1)
for _ in range(50): gp.predict_structures(batch, return_std=True)
torch.cuda.synchronize()
prof = dev.profile_fetch()
dev.profile_enable(False)
ks = ", ".join(f"{k} {ms / n * 1e3:.1f} us x{n // 50}" for k, (ms, n) in prof.items() if n)
print(f"batch {nb}: latency median {np.median(lat):.1f} us, min {lat.min():.1f}, p90 {np.percentile(lat, 90):.1f}; kernels: {ks}")
