"""cfg4 size (N_train = 8000, N = 24 250) with four ranks AS THREADS on one GPU (tests/dist_helpers.ThreadComm stands where
RCCL stands on a node): the block-cyclic Cholesky of distributed.py on the device kernels against the single-GPU
factorisation, and the sharded covariance assembly against the single-rank matrix.  Correctness at full size, not a
scaling measurement (the four ranks share one device, the collectives are host-side copies).
usage: python tools/cfg4_sharded_check.py [ranks]"""
import os, sys, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from dist_helpers import run_ranks
from gpr_calculator_amd import device as dev
from gpr_calculator_amd.distributed import DeviceOps, potrf_distributed
import bench
from gpr_calculator_amd.parallel import Comm

size = int(sys.argv[1]) if len(sys.argv) > 1 else 4
d = dev.device()
n = 24250
torch.manual_seed(0)
B = torch.randn(n, 2048, dtype=torch.float64, device=d)
A = B @ B.T
A.diagonal().add_(0.05 * n)
del B
ref = A.clone()
torch.cuda.synchronize(); t0 = time.perf_counter()
info, dinv = dev.potrf_lower_(ref, want_dinv=True)
torch.cuda.synchronize(); t1 = time.perf_counter()
print(f"single GPU potrf N={n}: {1e3 * (t1 - t0):.1f} ms, info {int(info.item())}", flush=True)
t0 = time.perf_counter()
outs = run_ranks(size, lambda comm: potrf_distributed(A, comm, DeviceOps())[0])
torch.cuda.synchronize(); t1 = time.perf_counter()
errs = [float((L - ref).abs().max() / ref.abs().max()) for L in outs]
print(f"block-cyclic, {size} thread ranks on one GPU: {t1 - t0:.1f} s wall; |L - L_single| / |L| per rank: {['%.1e' % e for e in errs]}", flush=True)
assert max(errs) < 1e-12
del outs, ref, A
torch.cuda.empty_cache()

gp = bench.build_model("RBF", 8000, Comm())
torch.cuda.synchronize(); t0 = time.perf_counter()
K1 = gp.kernel.k_total_dev(gp.train_x)
torch.cuda.synchronize(); t1 = time.perf_counter()
print(f"single-rank RBF k_total N={K1.shape[0]}: {1e3 * (t1 - t0):.0f} ms", flush=True)
outs = run_ranks(size, lambda comm: float((gp.kernel.k_total_dev(gp.train_x, comm=comm) - K1).abs().max() / K1.abs().max()))
print(f"sharded assembly, {size} ranks: |K - K_single| / |K| per rank: {['%.1e' % e for e in outs]}", flush=True)
assert max(outs) < 1e-13
print("OK")
