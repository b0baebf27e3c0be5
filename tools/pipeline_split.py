"""Host-side cost of the batches-in-flight loop (Dot cfg3): how long a submission and a collection hold the host thread,
against the step time -- tells whether the pipeline is bound by the host (launch calls) or by the device.
usage: python tools/pipeline_split.py [depth] [batch]"""
import sys, os, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from collections import deque
import numpy as np, torch
import bench
from gpr_calculator_amd.synthetic import neb_images
from gpr_calculator_amd.parallel import Comm

depth = int(sys.argv[1]) if len(sys.argv) > 1 else 3
gp = bench.build_model("Dot", 2000, Comm())
gp.fit(show=False, opt=False)
gp.session_depth = depth
batch = neb_images(int(sys.argv[2]) if len(sys.argv) > 2 else 64, seed=100)
for _ in range(30): gp.predict_structures(batch, return_std=True)
bench.run_pipelined(gp, batch, 50, depth)
N = 400
ts, tc = [], []
q = deque()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(N):
    if len(q) == depth:
        a = time.perf_counter_ns(); gp.predict_structures_collect(q.popleft()); tc.append(time.perf_counter_ns() - a)
    a = time.perf_counter_ns(); q.append(gp.predict_structures_submit(batch, return_std=True)); ts.append(time.perf_counter_ns() - a)
while q:
    a = time.perf_counter_ns(); gp.predict_structures_collect(q.popleft()); tc.append(time.perf_counter_ns() - a)
torch.cuda.synchronize(); tot = (time.perf_counter() - t0) / N * 1e3
print(f"depth {depth}: step {tot * 1e3:.1f} us; submit median {np.median(ts) * 1e-3:.1f} us (min {np.min(ts) * 1e-3:.1f}), "
      f"collect median {np.median(tc) * 1e-3:.1f} us (min {np.min(tc) * 1e-3:.1f})")
