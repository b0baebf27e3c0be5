"""FP64 tiled GEMM (gpr_gemm; NT form as in the Cholesky trailing update, or GEMM_FORM=nn / tn / tt) throughput against k;
usage: [m]; GEMM_KS=256,1024 picks the k values, GEMM_NO_VENDOR=1 skips the torch.addmm line"""
import os, sys, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
from gpr_calculator_amd import device as dev
m = int(sys.argv[1]) if len(sys.argv) > 1 else 5807
d = dev.device()
for k in [int(x) for x in os.environ.get("GEMM_KS", "64,128,256,512,1024,4096").split(",")]:
    A = torch.randn(m, k, dtype=torch.float64, device=d)
    C = torch.zeros(m, m, dtype=torch.float64, device=d)
    for beta in (0.0, 1.0):
        form = os.environ.get("GEMM_FORM", "nt")
        At = A.T.contiguous()
        X, Y = {"nt": (A, A), "nn": (A, At), "tn": (At, At), "tt": (At, A)}[form]
        run = lambda: dev.gemm(X, Y, transa=form[0] == "t", transb=form[1] == "t", alpha=-1.0, beta=beta, out=C)
        run()
        torch.cuda.synchronize(); ts = []
        for _ in range(5):
            t0 = time.perf_counter(); run(); torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        t = min(ts)
        print(f"{form} m=n={m} k={k:5d} beta={beta}: {t * 1e3:8.3f} ms  {2.0 * m * m * k / t / 1e12:6.2f} TF", flush=True)
    if os.environ.get("GEMM_NO_VENDOR"): continue
    t0 = time.perf_counter(); R = A @ A.T; torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); R = torch.addmm(C, A, A.T, beta=1.0, alpha=-1.0); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    print(f"     torch addmm (hipBLASLt/rocBLAS) k={k}: {min(ts) * 1e3:8.3f} ms  {2.0 * m * m * k / min(ts) / 1e12:6.2f} TF", flush=True)
