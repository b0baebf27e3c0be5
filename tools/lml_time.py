"""Wall time of one log-marginal-likelihood + gradient evaluation at cfg3 size (what every L-BFGS-B step of
fit(opt=True) costs): weight-space route (Dot, zeta = 2) against the dense N x N route."""
import sys, os, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np, torch
import bench
from gpr_calculator_amd.parallel import Comm
for kind in (sys.argv[1:] or ["Dot", "RBF"]):
    gp = bench.build_model(kind, 2000, Comm())
    p = np.array(list(gp.kernel.parameters()))
    for ws in ((True, False) if kind == "Dot" else (False,)):
        gp.weight_space_fit = ws
        gp.log_marginal_likelihood(p, eval_gradient=True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3): lml, g = gp.log_marginal_likelihood(p, eval_gradient=True)
        torch.cuda.synchronize()
        print(f"{kind} weight_space={ws}: {(time.perf_counter() - t0) / 3 * 1e3:8.2f} ms per LML+gradient   lml={lml:.6f} grad={g}")
