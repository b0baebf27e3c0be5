"""What a refit costs when the training set has just grown by one structure (the on-the-fly loop of calculator.py:95-117):
add_structure + fit(opt=False) and + fit(opt=True, maxiter=10), cfg3-sized model.  usage: python tools/refit_cost.py [Dot|RBF]"""
import sys, os, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np, torch
import bench
from gpr_calculator_amd.parallel import Comm
from gpr_calculator_amd.synthetic import fcc_cell, lj_energy_forces

kind = sys.argv[1] if len(sys.argv) > 1 else "Dot"
gp = bench.build_model(kind, 2000, Comm())
gp.fit(show=False, opt=False)
def sync(): torch.cuda.synchronize()
for opt in (False, False, False, True, True):
    a = fcc_cell(seed=9000 + int(time.time() * 1e3) % 1000)
    E, F = lj_energy_forces(a)
    sync(); t0 = time.perf_counter()
    gp.add_structure((a, E, F), N_max=4, tol_e_var=-1.0, tol_f_var=-1.0)
    sync(); t1 = time.perf_counter()
    gp.fit(show=False, opt=opt, maxiter=10)
    sync(); t2 = time.perf_counter()
    print(f"{kind}: add_structure {1e3 * (t1 - t0):7.2f} ms   fit(opt={opt}) {1e3 * (t2 - t1):8.2f} ms   N = {len(gp.y_train)}", flush=True)
