"""Block stacking helpers of the kernel objects (gpr_calc/kernels/base.py:3-38)."""
import numpy as np


def build_covariance(c_ee, c_ef, c_fe, c_ff, c_se=None, c_sf=None):
    """np.block of whichever blocks exist (gpr_calc/kernels/base.py:3-30)."""
    exist = [x is not None for x in (c_ee, c_ef, c_fe, c_ff)]
    if False not in exist:
        return np.block([[c_ee, c_ef], [c_fe, c_ff]])
    if exist == [False, False, True, True]:     # F in train, E/F in predict
        return np.hstack((c_fe, c_ff))
    if exist == [True, True, False, False]:     # E in train, E/F in predict
        return np.hstack((c_ee, c_ef))
    if exist == [False, True, False, False]:
        return c_ef
    if exist == [True, False, False, False]:
        return c_ee
    if exist == [False, False, False, True]:
        return c_ff
    if exist == [False, False, True, False]:
        return c_fe
    return None


def get_mask(ele1, ele2):
    """indices of species-mismatched row pairs (gpr_calc/kernels/base.py:32-38)"""
    ans = ele1[:, None] - ele2[None, :]
    ids = np.where(ans != 0)
    return None if len(ids[0]) == 0 else ids
