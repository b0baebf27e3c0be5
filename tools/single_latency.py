"""Latency of predicting ONE structure (what an ASE optimiser sees per calculator call: GPR.calculate -> predict_structure),
Dot cfg3 model, with the kernels' own times beside it.
usage: python tools/single_latency.py [batch=1] [reps=300] [--profile]"""
import sys, os, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np, torch
import bench
from gpr_calculator_amd import device as dev
from gpr_calculator_amd.synthetic import neb_images
from gpr_calculator_amd.parallel import Comm

prof_host = "--profile" in sys.argv
argv = [a for a in sys.argv[1:] if not a.startswith("--")]
nb = int(argv[0]) if len(argv) > 0 else 1
reps = int(argv[1]) if len(argv) > 1 else 300
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
if nb == 1:
    # the same structure through the ASE calculator (GPR.calculate: copy of the atoms, prediction, uncertainty test)
    from gpr_calculator_amd.calculator import GPR
    calc = GPR(ff=gp, return_std=True)
    calc.verbose = False
    calc.update_gpr = False
    at = batch[0].copy()
    for _ in range(30): calc.calculate(at)
    lat = []
    for _ in range(reps):
        a = time.perf_counter_ns(); calc.calculate(at); lat.append(time.perf_counter_ns() - a)
    lat = np.array(lat) * 1e-3
    print(f"GPR.calculate: latency median {np.median(lat):.1f} us, min {lat.min():.1f}, p90 {np.percentile(lat, 90):.1f}")
    if prof_host:
        import cProfile, pstats
        pr = cProfile.Profile(); pr.enable()
        for _ in range(200): calc.calculate(at)
        pr.disable()
        pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
