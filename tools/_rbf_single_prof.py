import sys, os, time, cProfile, pstats, io
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np, torch
import importlib.util
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
spec = importlib.util.spec_from_file_location("cfg5", os.path.join(ROOT, "tools", "cfg5_loop.py"))
import contextlib
with contextlib.redirect_stdout(io.StringIO()):
    m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
gp, _ = m.run()
at = m.make(int(m.T["script"][2][0]), 5, 0.01)
for _ in range(20): gp.predict_structure(at, stress=False, return_std=True)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(100): gp.predict_structure(at, stress=False, return_std=True)
torch.cuda.synchronize(); print(f"RBF predict_structure (220 atoms, N = {len(gp.train_x['energy'][-1])} + 3 x {len(gp.train_x['force'][-1])}): {(time.perf_counter() - t0) * 10:.3f} ms per call")
pr = cProfile.Profile(); pr.enable()
for _ in range(100): gp.predict_structure(at, stress=False, return_std=True)
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28); print(s.getvalue()[:6000])
