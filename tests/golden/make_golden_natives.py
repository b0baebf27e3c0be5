"""Generate tests/golden/natives_*.npz from the UNMODIFIED reference C++ natives.

Run in the build container only (needs /root/reference to build oracle/_ref):
    make -C oracle ref && python tests/golden/make_golden_natives.py
Each file holds the inputs of one case (oracle.natives.make_case) and, under "out/<name>", the
output array every one of the 13 reference natives produced for it (oracle.natives.run_all_natives
with NativeSet("ref")).  Fixtures are data only; no reference source is stored.
"""
import os, sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.abspath(os.path.join(HERE, "..", "..")))
from oracle.natives import NativeSet, make_case, run_all_natives

CASES = {
    # name: kwargs of make_case
    "single_d30":   dict(seed=0, d=30, counts1=[5, 9, 1, 7], counts2=[6, 4, 11, 3, 8]),
    "mixed_d30":    dict(seed=1, d=30, counts1=[4, 13, 2, 6], counts2=[17, 1, 5, 9], species=(8, 12, 46)),
    "zeta3_mixed":  dict(seed=2, d=30, counts1=[3, 8, 5], counts2=[7, 2, 9, 4], species=(8, 46), zeta=3.0,
                         sigma=1.7, l=0.9, sigma0=1.5),
    "d12_zero_row": dict(seed=3, d=12, counts1=[6, 5, 8], counts2=[4, 9, 7], species=(13, 79),
                         zero_rows=(("1", 2), ("2", 10))),
    "subrange_tol": dict(seed=4, d=30, counts1=[10, 12], counts2=[8, 16, 6], sigma=1.0, l=0.2,
                         tol=2e-3, n2_range=(5, 23)),
    "pd4_params":   dict(seed=5, d=30, counts1=[21, 19], counts2=[18, 23, 16], species=(8, 12, 46),
                         sigma=24.99290767873284, l=3.107283211515612),
    # longer descriptors: SO3(nmax=4, lmax=4) has d = 50; 64 is the widest plane of the device kernels
    "d50_mixed":    dict(seed=6, d=50, counts1=[5, 11, 3], counts2=[9, 4, 13, 6], species=(8, 46)),
    "d64_zeta3":    dict(seed=7, d=64, counts1=[7, 6], counts2=[5, 10, 8], species=(13, 79), zeta=3.0, sigma=1.7,
                         l=0.9, sigma0=1.5),
    # beyond 64 columns (round 4): SO3(nmax=5, lmax=4) has d = 75, SO3(6, 4) 105; 128 is the widest plane now
    "d75_mixed":    dict(seed=8, d=75, counts1=[6, 9, 4], counts2=[8, 5, 12, 3], species=(8, 12, 46)),
    "d105_zeta3":   dict(seed=9, d=105, counts1=[5, 7], counts2=[6, 9, 4], species=(8, 46), zeta=3.0, sigma=1.7, l=0.9,
                         sigma0=1.5),
    "d128_subrange": dict(seed=10, d=128, counts1=[9, 8], counts2=[7, 11, 6], sigma=1.2, l=0.6, tol=1e-3, n2_range=(4, 20)),
}

if __name__ == "__main__":
    ref = NativeSet("ref")
    port = NativeSet("port")
    only = sys.argv[1:]
    for name, kw in CASES.items():
        if only and name not in only:
            continue
        c = make_case(**kw)
        out = run_all_natives(ref, c)
        chk = run_all_natives(port, c)
        worst = max(np.abs(out[k] - chk[k]).max() / (np.abs(out[k]).max() + 1e-300) for k in out)
        payload = dict(c)
        payload.update({"out/" + k: v for k, v in out.items()})
        path = os.path.join(HERE, f"natives_{name}.npz")
        np.savez_compressed(path, **payload)
        print(f"{name:14s} -> {os.path.basename(path)} ({os.path.getsize(path)/1024:.0f} KiB); "
              f"port vs ref worst rel = {worst:.2e}")
