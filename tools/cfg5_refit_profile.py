"""Where the on-the-fly refit of the cfg5-shaped loop (tools/cfg5_loop.py) spends its time: synchronised stage timers
around the host entry points of one `GPR.calculate` that labels a structure and refits, and the number of device
synchronisations / launches per stage.  Run plainly for the stage table; run under
`rocprofv3 --kernel-trace --stats -- python3 tools/cfg5_refit_profile.py` for the per-kernel summary of the same loop.
usage: python tools/cfg5_refit_profile.py [--cprofile]"""
import collections, contextlib, cProfile, io, os, pstats, sys, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
sys.argv, ARGS = sys.argv[:1], sys.argv[1:]
import numpy as np, torch
import tools.cfg5_loop as loop                      # runs the loop twice on import (warm)
from gpr_calculator_amd import device as dev, gaussianprocess as gpmod, synthetic
from gpr_calculator_amd.calculator import GPR
from gpr_calculator_amd.gaussianprocess import GP
from gpr_calculator_amd.kernels.RBF_mb import RBF_mb

acc = collections.OrderedDict()
depth = [0]


def timed(owner, name, label=None):
    f = getattr(owner, name)
    label = label or name

    def g(*a, **k):
        torch.cuda.synchronize(); t0 = time.perf_counter(); depth[0] += 1
        try:
            return f(*a, **k)
        finally:
            torch.cuda.synchronize(); depth[0] -= 1
            e = acc.setdefault(label, [0, 0.0]); e[0] += 1; e[1] += time.perf_counter() - t0
    setattr(owner, name, g)


timed(loop.Base, "_eval", "base calculator (numpy LJ on the host; not GPR)")
timed(GP, "predict_structure", "predict_structure (surrogate)")
timed(GP, "add_structure", "add_structure")
timed(gpmod, "convert_train_data", "  add_structure: descriptor + packing")
timed(GP, "validate_data", "validate_data (in add_structure and after the refit)")
timed(GP, "fit", "fit(opt=True, maxiter=10)")
timed(GP, "log_marginal_likelihood", "  one LML + gradient evaluation")
timed(RBF_mb, "k_total_with_grad_dev", "    k_total_with_grad_dev")
timed(RBF_mb, "k_total_dev", "    k_total_dev")
timed(dev, "potrf_factor", "    potrf_factor (build + factor)")
timed(dev, "trtri_lower", "    trtri_lower")
timed(dev, "xtx_lower", "    xtx_lower")
timed(dev, "trsm_lower_", "    trsm_lower_")
timed(GP, "_factor_dense", "  final factorisation (_factor_dense)")

pr = cProfile.Profile() if "--cprofile" in ARGS else None
if pr:
    pr.enable()
gp, secs = loop.run()
if pr:
    pr.disable()
print("\nstage timers over the 12 steps (synchronised around every stage, so the stages add up to more than the free-running loop):")
print(f"{'stage':62s} {'calls':>6s} {'total ms':>10s} {'ms/call':>9s}")
for k, (n, s) in acc.items():
    print(f"{k:62s} {n:6d} {s * 1e3:10.2f} {s * 1e3 / n:9.3f}")
print(f"loop total {sum(secs) * 1e3:.1f} ms; N at the end = {len(gp.y_train)} (energy {gp.N_energy}, force {gp.N_forces})")
if pr:
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(70)
    pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(40)
    print(s.getvalue())
