"""Where fit(opt=True, maxiter=10) spends its time at cfg3 size: evaluations of the log-marginal likelihood (count, ms each),
the optimiser around them, the final fit.  usage: python tools/fit_opt_profile.py [Dot|RBF ...]"""
import sys, os, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np, torch
import bench
from gpr_calculator_amd.parallel import Comm
for kind in (sys.argv[1:] or ["Dot", "RBF"]):
    gp = bench.build_model(kind, 2000, Comm())
    gp.fit(show=False, opt=False)                      # warm: allocator pools, lazy tables
    calls = []
    orig = gp.log_marginal_likelihood
    def timed(*a, **k):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = orig(*a, **k)
        torch.cuda.synchronize(); calls.append(time.perf_counter() - t0)
        return r
    gp.log_marginal_likelihood = timed
    torch.cuda.synchronize(); t0 = time.perf_counter()
    gp.fit(show=False, opt=True, maxiter=10)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    c = np.array(calls) * 1e3
    print(f"{kind}: fit(opt=True, maxiter=10) {dt * 1e3:7.1f} ms = {len(c)} LML evaluations x {c.mean():.2f} ms (first {c[0]:.2f}, "
          f"min {c.min():.2f}, sum {c.sum():.1f}) + {dt * 1e3 - c.sum():.1f} ms outside them")
