"""ws_tail alone at the headline shape (m = 6208 predicted rows, F = 466): time per call (tail + combine launch) and, with
GPR_TAIL_DBG, the ablations of the LDS-DMA kernel (1 no loads, 2 no matrix products, 4 no LDS operand reads).
usage: python tools/tail_bench.py [m] [n]"""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np, torch
from gpr_calculator_amd import _lib, device as dev
m = int(sys.argv[1]) if len(sys.argv) > 1 else 6208
n = int(sys.argv[2]) if len(sys.argv) > 2 else 466
rng = np.random.default_rng(0)
P = dev.to_dev(rng.standard_normal((m, n)))
C = dev.to_dev(np.tril(rng.standard_normal((n, n))) * 0.05)
w = dev.to_dev(rng.standard_normal(n)); d = dev.to_dev(np.full(m, 1e3))
out = torch.empty(2 * m, dtype=torch.float64, device=P.device)
call = lambda: dev.check(_lib.lib().gpr_ws_tail(dev.stream_ptr(), m, n, dev.ptr(P), n, dev.ptr(C), n, dev.ptr(w), dev.ptr(d), dev.ptr(out), 4.0), "gpr_ws_tail")
for _ in range(20): call()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
best = 1e9
for rep in range(5):
    e0.record()
    for _ in range(100): call()
    e1.record(); torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) / 100)
flop = m * n * (n + 1) + 4.0 * m * n
print(f"GPR_TAIL_DBG={os.environ.get('GPR_TAIL_DBG', '0')} NO_GLDS={os.environ.get('GPR_TAIL_NO_GLDS', '')}: {best * 1e3:.1f} us per call, {flop / best / 1e9:.1f} TF")
if not os.environ.get("GPR_TAIL_DBG"):
    Pn, Cn = P.cpu().numpy(), C.cpu().numpy()
    Q = Pn @ Cn.T
    mean = Pn @ w.cpu().numpy()
    var = np.maximum(1e3 - 4.0 * ((Pn * Pn).sum(1) - (Q * Q).sum(1)), 0)
    o = out.cpu().numpy()
    print("  max rel err mean %.2e var %.2e" % (np.abs(o[:m] - mean).max() / np.abs(mean).max(), np.abs(o[m:] - var).max() / np.abs(var).max()))
