"""one structure per call through the pair-kernel route (RBF, cfg2 size: N_train = 500): where a GP.predict_structure call spends
its host time (cProfile) next to the kernels' own time.  usage: python tools/rbf_single_profile.py [n_train]"""
import cProfile, io, os, pstats, sys, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np, torch
import bench
from gpr_calculator_amd import device as dev
from gpr_calculator_amd.parallel import Comm
from gpr_calculator_amd.synthetic import neb_images
n_train = int(sys.argv[1]) if len(sys.argv) > 1 else 500
gp = bench.build_model("RBF", n_train, Comm())
gp.fit(show=False, opt=False)
s = neb_images(1, seed=100)[0]
for _ in range(20): gp.predict_structure(s, stress=False, return_std=True, f_tol=1e-8)
torch.cuda.synchronize()
dev.profile_enable(True)
t0 = time.perf_counter()
for _ in range(200): gp.predict_structure(s, stress=False, return_std=True, f_tol=1e-8)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 200
prof = dev.profile_fetch(); dev.profile_enable(False)
print(f"N_train = {n_train}: predict_structure {dt * 1e3:.3f} ms per call; kernels (HIP events): " +
      ", ".join(f"{k} {ms / 200 * 1e3:.0f} us x{n / 200:.0f}" for k, (ms, n) in prof.items() if n))
pr = cProfile.Profile(); pr.enable()
for _ in range(200): gp.predict_structure(s, stress=False, return_std=True, f_tol=1e-8)
torch.cuda.synchronize(); pr.disable()
out = io.StringIO(); pstats.Stats(pr, stream=out).sort_stats("tottime").print_stats(28); print(out.getvalue()[:6000])
