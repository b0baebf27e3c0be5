import sys, os
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
from gpr_calculator_amd.SO3 import SO3
from gpr_calculator_amd.synthetic import neb_images
from gpr_calculator_amd import device as dev
des = SO3(nmax=3, lmax=4, rcut=5.0, alpha=2.0)
batch = neb_images(64, seed=100)
des.calculate_batch_dev(batch); torch.cuda.synchronize()
dev.profile_enable(True)
for _ in range(5): des.calculate_batch_dev(batch)
torch.cuda.synchronize()
ms, n = dev.profile_fetch()["so3"]
print(f"so3 {os.environ.get('GPR_SO3_DBG','0')}: {ms/n:.3f} ms")
