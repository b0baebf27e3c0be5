"""kee_C / kef_C / kff_C of the Dot kernel with the reference's signatures and return values
(gpr_calc/kernels/dot_kernel.py:9-270), evaluated by the HIP blocks."""
import numpy as np
import torch

from .. import _lib
from .. import device as dev
from ._common import as_force_tuple
from .rbf_kernel import _eset, _fsets


def _kp(sigma, sigma0, zeta):
    return dev.kernel_params(_lib.KERNEL_DOT, zeta, sigma, 1.0, sigma0, 0.0)


def kee_C(X1, X2, sigma=1.0, sigma0=1.0, zeta=2.0, grad=False):
    s1, s2 = _eset(X1, _lib.ROLE_STREAM), _eset(X2, _lib.ROLE_RESIDENT)
    C = dev.block_ee(_kp(sigma, sigma0, zeta), s1, s2).cpu().numpy()
    if grad:
        return C, 2 * C / sigma, 0.8 * 2 * sigma ** 2 * sigma0 * np.ones(C.shape)   # dot_kernel.py:56-58
    return C


def kef_C(X1, X2, sigma=1.0, sigma0=1.0, zeta=2.0, grad=False, stress=False, transpose=False):
    s1 = _eset(X1, _lib.ROLE_STREAM)
    f2 = _fsets(X2, _lib.ROLE_RESIDENT, stress=stress)
    kp = _kp(sigma, sigma0, zeta)
    m1, m2 = s1.npts, f2[0].npts
    C = dev.block_ef(kp, s1, f2[0]).cpu().numpy()
    if stress:
        B1 = dev.block_ef(kp, s1, f2[1]).view(m1, m2, 3)
        B2 = dev.block_ef(kp, s1, f2[2]).view(m1, m2, 3)
        Cs = torch.cat((B1, B2), dim=2).reshape(m1, m2 * 6).cpu().numpy()
    else:
        Cs = np.zeros([m1, m2 * 6])
    if transpose:
        C, Cs = C.T, Cs.T
    if grad:
        return C, 2 * C / sigma, np.zeros([m1, m2 * 3])
    return (C, Cs) if stress else C


def kff_C(X1, X2, sigma=1.0, sigma0=1.0, zeta=2.0, grad=False, stress=False):
    f1 = _fsets(X1, _lib.ROLE_STREAM, stress=stress)
    x2, dx2, e2, c2 = as_force_tuple(X2)
    f2 = dev.PackedSet(_lib.ROLE_RESIDENT, x2, np.ascontiguousarray(dx2[:, :, :3]), e2, c2)
    kp = _kp(sigma, sigma0, zeta)
    m1, m2 = f1[0].npts, f2.npts
    C = dev.block_ff(kp, f1[0], f2)
    if grad:
        C = C.cpu().numpy()
        return C, 2 * C / sigma, np.zeros([m1 * 3, m2 * 3])
    if stress:
        parts = [dev.block_ff(kp, f1[t], f2).view(m1, 3, m2 * 3) for t in (1, 2)]
        return C.cpu().numpy(), torch.cat(parts, dim=1).reshape(m1 * 6, m2 * 3).cpu().numpy()
    return C.cpu().numpy()
