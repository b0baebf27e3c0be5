"""Kernel micro-benchmarks on random rows (SURVEY.md section 8(d) micro-bench inputs): row-pairs/s of the
force-force block and TFLOP/s of the dense kernels.  Usage: python tools/microbench.py [npts] [rows_per_point]"""
import sys, os, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np
import torch
from gpr_calculator_amd import device as dev, _lib


def timeit(fn, reps=3):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e-3)
    return min(ts)


def main():
    npts = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    rpp = int(sys.argv[2]) if len(sys.argv) > 2 else 43
    rng = np.random.default_rng(0)
    R, d = npts * rpp, 30
    x = rng.random((R, d)); dx = rng.standard_normal((R, d, 3)); ele = np.full(R, 46, np.int32)
    counts = np.full(npts, rpp, np.int32)
    pair = dev.SetPair(x, dx, ele, counts)
    print("packed:", pair.stream.info(), pair.resident.info())
    out = torch.empty((3 * npts, 3 * npts), dtype=torch.float64, device="cuda")
    for name, kp in (("rbf", dev.kernel_params(_lib.KERNEL_RBF, 2.0, 2.5, 1.3, tol=1e-12)),
                     ("dot", dev.kernel_params(_lib.KERNEL_DOT, 2.0, 2.0, sigma0=2.0))):
        for sym in (False, True):
            t = timeit(lambda: dev.block_ff(kp, pair.stream, pair.resident, symmetric=sym, out=out))
            pairs = R * R * (0.5 if sym else 1.0)
            print(f"kff {name} sym={sym}: {t*1e3:9.2f} ms  {pairs/t/1e9:8.2f} Gpairs/s  "
                  f"{pairs*1.10e3/t/1e12:6.2f} TF (1.10 kflop/pair) = {pairs*1.10e3/t/78.6e12*100:5.1f}% of 78.6 TF")
    # small predict-like block: 32 points vs npts
    sub = dev.PackedSet(_lib.ROLE_STREAM, x[:32 * rpp], dx[:32 * rpp], ele[:32 * rpp], counts[:32])
    o2 = torch.empty((96, 3 * npts), dtype=torch.float64, device="cuda")
    kp = dev.kernel_params(_lib.KERNEL_DOT, 2.0, 2.0, sigma0=2.0)
    t = timeit(lambda: dev.block_ff(kp, sub, pair.resident, out=o2))
    print(f"kff predict-shaped 32 x {npts} points: {t*1e3:.3f} ms  {32*rpp*R/t/1e9:.2f} Gpairs/s")
    # dense
    n = 3 * npts
    a = torch.randn(n, n, dtype=torch.float64, device="cuda")
    spd = dev.gemm(a, a, transb=True); dev.add_diag_(spd, 0, 0.0, float(n))
    work = spd.clone()
    t = timeit(lambda: (work.copy_(spd), dev.potrf_lower_(work)))
    print(f"potrf n={n}: {t*1e3:.2f} ms  {n**3/3/t/1e12:.2f} TF")
    b = torch.randn(n, 6208, dtype=torch.float64, device="cuda")
    t = timeit(lambda: dev.trsm_lower_(work, b.clone()))
    print(f"trsm n={n} nrhs=6208: {t*1e3:.2f} ms  {n*n*6208/t/1e12:.2f} TF")
    t = timeit(lambda: dev.gemm(a, a))
    print(f"gemm n={n}: {t*1e3:.2f} ms  {2*n**3/t/1e12:.2f} TF")


if __name__ == "__main__":
    main()
