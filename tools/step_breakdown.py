"""Wall-clock breakdown of one predict step (host + device), cfg3."""
import sys, os, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np, torch
import bench
from gpr_calculator_amd import device as dev
from gpr_calculator_amd.synthetic import neb_images
from gpr_calculator_amd.parallel import Comm

gp = bench.build_model("Dot", 2000, Comm())
gp.fit(show=False, opt=False)
batch = neb_images(64, seed=100)
gp.predict_structures(batch, return_std=True)
def T(label, fn, n=3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): r = fn()
    torch.cuda.synchronize(); print(f"{label:40s} {(time.perf_counter()-t0)/n*1e3:8.2f} ms"); return r
T("full step", lambda: gp.predict_structures(batch, return_std=True))
T("neighbor_index x64 (host)", lambda: [gp.descriptor.neighbor_index(s) for s in batch])
desc = T("calculate_batch_dev (host idx + so3)", lambda: gp.descriptor.calculate_batch_dev(batch))
X, frees = T("_batch_data_dev (desc+gather+pack)", lambda: gp._batch_data_dev(batch))
train_x = gp.get_train_x()
Kt = T("k_total_dev transpose", lambda: gp.kernel.k_total_dev(X, train_x, transpose=True))
T("mean gemm", lambda: dev.gemm(Kt, gp._alpha_dev, transa=True))
T("diag_dev", lambda: gp.kernel.diag_dev(X))
T("trmm", lambda: dev.trmm_lower(gp._Linv(), Kt))
V = dev.trmm_lower(gp._Linv(), Kt)
T("colsumsq", lambda: dev.colsumsq(V))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable(); gp.predict_structures(batch, return_std=True); torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(25)
