"""The weight-space fit (Dot, zeta = 2) alone, repeated: target of a rocprofv3 kernel trace (tools/prof_cmd.sh) -- which
launches a fit consists of.  usage: python tools/ws_fit_only.py [reps=20] [dense]"""
import sys, os, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
import bench
from gpr_calculator_amd.parallel import Comm
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
gp = bench.build_model("Dot", 2000, Comm())
gp.weight_space_fit = "dense" not in sys.argv
for _ in range(3): gp.fit(show=False, opt=False)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(reps): gp.fit(show=False, opt=False)
torch.cuda.synchronize()
print(f"fit x{reps}: {(time.perf_counter() - t0) / reps * 1e3:.3f} ms each ({'weight space' if gp.weight_space_fit else 'dense'})")
