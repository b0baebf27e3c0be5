"""What an initialised RCCL communicator costs the pipelined prediction of the SAME process (one rank; no collective inside the
timed loop): host time of a submission, of a collection, and the step, with and without the process group.
usage (nccl / gloo: under torch.distributed.run --nproc-per-node 1): python tools/rccl_presence.py [nccl|gloo|none]
  PRESENCE_DENSE_FIT=1        a dense fit first (its look-ahead context creates eight streams before the first prediction)
  PRESENCE_EXTRA_STREAMS=n    n unrelated torch streams created before the first prediction
  GPR_SESSION_POOL_LEAD=n     n unused streams ahead of the session's slot-stream pool; GPR_SESSION_POOL=0: streams per slot, lazily
what came out of it: profiles/r04_stream_order.log, DESIGN section 7 "Round 4, last"."""
import os, sys, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np, torch
mode = sys.argv[1] if len(sys.argv) > 1 else "nccl"
import torch.distributed as dist
if mode != "none":
    torch.cuda.set_device(0)
    dist.init_process_group(backend=mode)
    t = torch.ones(4, device="cuda" if mode == "nccl" else "cpu")
    dist.all_reduce(t); dist.barrier()
    torch.cuda.synchronize()
import bench
from gpr_calculator_amd.synthetic import neb_images
from gpr_calculator_amd.parallel import Comm
gp = bench.build_model("Dot", 2000, Comm())
if os.environ.get("PRESENCE_DENSE_FIT"):      # the dense route first: its look-ahead context creates streams before the session does
    gp.weight_space_fit = False; gp.fit(show=False, opt=False); gp.weight_space_fit = True
for _ in range(int(os.environ.get("PRESENCE_EXTRA_STREAMS", "0"))):
    _keep = globals().setdefault("_keep", []); _keep.append(torch.cuda.Stream())
if os.environ.get("PRESENCE_SKIP_WS_FIT") is None:
    gp.fit(show=False, opt=False)
gp.comm = bench._Serial()
gp.session_depth = 4
batch = neb_images(64, seed=5)
for _ in range(30):
    gp.predict_structures(batch, return_std=True)
def loop(n):
    sub, col = [], []
    q = []
    t0 = time.perf_counter()
    for _ in range(n):
        a = time.perf_counter(); q.append(gp.predict_structures_submit(batch, return_std=True)); b = time.perf_counter(); sub.append(b - a)
        if len(q) == 4:
            a = time.perf_counter(); gp.predict_structures_collect(q.pop(0)); b = time.perf_counter(); col.append(b - a)
    while q:
        gp.predict_structures_collect(q.pop(0))
    return (time.perf_counter() - t0) / n, np.median(sub), np.median(col)
loop(50)
for _ in range(3):
    s, a, b = loop(400)
    print(f"{mode}: step {s * 1e6:.1f} us, submit median {a * 1e6:.1f} us, collect median {b * 1e6:.1f} us", flush=True)
t0 = time.perf_counter()
for _ in range(100):
    gp.predict_structures(batch, return_std=True)
print(f"{mode}: synchronous step {(time.perf_counter() - t0) * 1e4:.1f} us")
if mode != "none":
    dist.destroy_process_group()
