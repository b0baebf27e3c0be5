import sys, os
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
from gpr_calculator_amd.SO3 import SO3
from gpr_calculator_amd.synthetic import neb_images
from gpr_calculator_amd import device as dev
des = SO3(nmax=3, lmax=4, rcut=5.0, alpha=2.0)
NB = int(os.environ.get('SO3_NB', '64'))
batch = neb_images(NB, seed=100)
from gpr_calculator_amd import _lib
for variant in (0, 2, 1):                           # 0: compile-time instance (radial pre-pass for large batches); 2: the same in one launch; 1: the generic kernel
    _lib.lib().gpr_so3_set_variant(1 if variant == 1 else 0)
    _lib.lib().gpr_so3_set_prepass_min(1 << 30 if variant == 2 else 512)
    out = des.calculate_batch_dev(batch); torch.cuda.synchronize()
    dev.profile_enable(True)
    for _ in range(20): des.calculate_batch_dev(batch)
    torch.cuda.synchronize()
    prof = dev.profile_fetch()
    ms, n = prof["so3"]                                 # the centre kernel; the radial pre-pass (variant 0) has a bracket of its own
    msr, nr = prof.get("so3_radial", (0.0, 0))
    dev.profile_enable(False)
    parts = f" = radial pre-pass {msr/nr:.4f} + centre kernel {ms/n:.4f}" if nr else ""
    print(f"so3 variant {variant} (phase {os.environ.get('GPR_SO3_DBG','0')}): {(ms + msr)/n:.4f} ms per call{parts} over {n} calls, "
          f"{out['x'].shape[0]} centres, {out['dxdr'].shape[0]} seq rows")
    if variant == 0: ref = out
    else:
        for k in ("x", "dxdr"):
            print(f"   fixed vs generic, {k}: max |diff| / max |value| = {float((ref[k] - out[k]).abs().max() / out[k].abs().max()):.2e}")
_lib.lib().gpr_so3_set_variant(0); _lib.lib().gpr_so3_set_prepass_min(512)
