"""Which slot streams of a prediction session share a hardware queue (see tools/stream_queues.py for the probe), in the very
process state bench.py times: model built and fitted first, then the session at the given depth.
usage: python tools/session_queues.py [depth=4]"""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
import bench
from gpr_calculator_amd import _lib, device as dev
from gpr_calculator_amd.parallel import Comm
from gpr_calculator_amd.synthetic import neb_images
depth = int(sys.argv[1]) if len(sys.argv) > 1 else 4
gp = bench.build_model("Dot", 2000, Comm())
gp.fit(show=False, opt=False)
gp.session_depth = depth
batch = neb_images(64, seed=11)
for _ in range(3):
    tk = [gp.predict_structures_submit(batch, return_std=True) for _ in range(depth)]
    for t in tk:
        gp.predict_structures_collect(t)
ses = gp._session[1]
ptrs = [_lib.lib().gpr_dot2_session_slot_stream(ses, k) for k in range(depth)]
streams = [torch.cuda.current_stream()] + [torch.cuda.ExternalStream(p) for p in ptrs if p]
x = torch.zeros(1, device=dev.device())
spin = 20_000_000
n = len(streams)
print(f"depth {depth}: stream 0 = torch's current stream, 1.. = session slots; X: j waits behind i   (GPU_MAX_HW_QUEUES = {os.environ.get('GPU_MAX_HW_QUEUES', 'default')})")
for i, si in enumerate(streams):
    row = ""
    for j, sj in enumerate(streams):
        if i == j:
            row += "X"; continue
        ei, ej = torch.cuda.Event(), torch.cuda.Event()
        torch.cuda.synchronize()
        with torch.cuda.stream(si):
            torch.cuda._sleep(spin); ei.record(si)
        with torch.cuda.stream(sj):
            x.add_(1.0); ej.record(sj)
        ej.synchronize()
        row += "X" if ei.query() else "."
        torch.cuda.synchronize()
    print(f"{i:2d} {row}")
