"""Where the host's time of one session submission goes (64-structure batch, cfg3 model): Python around the call (gathering the
batch's arrays) against the C entry point gpr_dot2_session_submit itself (staging blob, allocations, six launches).
usage: python tools/submit_split.py"""
import os, sys, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np
import bench
from gpr_calculator_amd import _lib
from gpr_calculator_amd.parallel import Comm
from gpr_calculator_amd.synthetic import neb_images

gp = bench.build_model("Dot", 2000, Comm())
gp.fit(show=False, opt=False)
batch = neb_images(64, seed=5)
for _ in range(20):
    gp.predict_structures(batch, return_std=True)
L = _lib.lib()
inner = []
orig = L.gpr_dot2_session_submit


def timed(*a):
    t0 = time.perf_counter()
    rc = orig(*a)
    inner.append(time.perf_counter() - t0)
    return rc


L.gpr_dot2_session_submit = timed
outer, coll = [], []
for _ in range(400):
    t0 = time.perf_counter()
    tk = gp.predict_structures_submit(batch, return_std=True)
    t1 = time.perf_counter()
    gp.predict_structures_collect(tk)
    t2 = time.perf_counter()
    outer.append(t1 - t0); coll.append(t2 - t1)
L.gpr_dot2_session_submit = orig
med = lambda v: float(np.median(v)) * 1e6
print(f"submit: {med(outer):.1f} us in all, of which the C call {med(inner):.1f} us (Python around it {med(outer) - med(inner):.1f}); "
      f"collect (waits for the device) {med(coll):.1f} us; step {med(outer) + med(coll):.1f} us")
