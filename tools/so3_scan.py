"""descriptor kernel alone against the number of centres: is the launch time (rounds of resident workgroups) x (life of a workgroup)?
usage: python tools/so3_scan.py"""
import sys, os
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
from gpr_calculator_amd.SO3 import SO3
from gpr_calculator_amd.synthetic import neb_images
from gpr_calculator_amd import device as dev
des = SO3(nmax=3, lmax=4, rcut=5.0, alpha=2.0)
for nb in (4, 8, 12, 16, 24, 32, 40, 48, 56, 64, 72, 96, 128):
    batch = neb_images(nb, seed=100)
    des.calculate_batch_dev(batch); torch.cuda.synchronize()
    dev.profile_enable(True)
    for _ in range(20): des.calculate_batch_dev(batch)
    torch.cuda.synchronize()
    prof = dev.profile_fetch()
    ms, n = prof["so3"]
    msr, nr = prof.get("so3_radial", (0.0, 0))          # from 512 centres on: radial pre-pass + centre kernel (five workgroups per CU)
    dev.profile_enable(False)
    print(f"{nb:4d} structures = {32 * nb:5d} centres: {(ms + msr) / n * 1e3:7.1f} us" + (f"  = radial {msr / nr * 1e3:5.1f} + centre {ms / n * 1e3:5.1f}" if nr else "  (one launch)"))
