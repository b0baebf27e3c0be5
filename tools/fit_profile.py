"""One dense fit (K assembly + Cholesky + alpha) under the in-library profiler and wall clock; usage: [RBF|Dot]"""
import sys, os, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
import bench
from gpr_calculator_amd import device as dev
from gpr_calculator_amd.parallel import Comm
kind = sys.argv[1] if len(sys.argv) > 1 else "RBF"
gp = bench.build_model(kind, 2000, Comm())
gp.weight_space_fit = False
gp.fit(show=False, opt=False)
for _ in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    gp.fit(show=False, opt=False)
    torch.cuda.synchronize(); print(f"{kind} dense fit {(time.perf_counter() - t0) * 1e3:.2f} ms")
