"""Stress of the prediction session: many calls with several batches in flight, batches of different sizes in rotation,
every result compared bit for bit with the synchronous call on the same batch (a late result row, a stale counter or
a shared temporary shows up as a mismatch or as 'force rows missing').
usage: python tools/session_stress.py [calls=20000] [depth=3]"""
import sys, os, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from collections import deque
import numpy as np, torch
import bench
from gpr_calculator_amd.synthetic import neb_images
from gpr_calculator_amd.parallel import Comm

calls = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 3
gp = bench.build_model("Dot", 2000, Comm())
gp.fit(show=False, opt=False)
gp.session_depth = depth
sizes = [1, 1, 2, 8, 1, 12, 3, 16, 1, 40, 64, 5]          # short-chain tail, staged split tail, row-resident tail; both feature forms
batches = [neb_images(n, seed=300 + i) for i, n in enumerate(sizes)]
want = []
for b in batches:
    for _ in range(3): r = gp.predict_structures(b, return_std=True)
    want.append([np.array(x, copy=True) for x in r])
bad, q, t0 = 0, deque(), time.perf_counter()
for c in range(calls):
    if len(q) == depth:
        i, tk = q.popleft()
        got = gp.predict_structures_collect(tk)
        bad += any(not np.array_equal(np.asarray(a), b) for a, b in zip(got, want[i]))
    i = (c * 7 + c // 13) % len(batches)
    q.append((i, gp.predict_structures_submit(batches[i], return_std=True)))
while q:
    i, tk = q.popleft()
    got = gp.predict_structures_collect(tk)
    bad += any(not np.array_equal(np.asarray(a), b) for a, b in zip(got, want[i]))
dt = time.perf_counter() - t0
print(f"{calls} calls, depth {depth}: {bad} mismatches, {dt / calls * 1e6:.1f} us per call")
sys.exit(1 if bad else 0)
