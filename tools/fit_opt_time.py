"""Wall time of fit(opt=True, maxiter=10) -- the refit the calculator triggers on the fly -- at cfg3 size."""
import sys, os, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np, torch
import bench
from gpr_calculator_amd.parallel import Comm
for kind in (sys.argv[1:] or ["Dot", "RBF"]):
    gp = bench.build_model(kind, 2000, Comm())
    p0 = list(gp.kernel.parameters())
    torch.cuda.synchronize(); t0 = time.perf_counter()
    gp.fit(show=False, opt=True, maxiter=10)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{kind}: fit(opt=True, maxiter=10) {dt * 1e3:8.1f} ms; parameters {p0} -> {[round(float(v), 4) for v in gp.kernel.parameters()]}")
