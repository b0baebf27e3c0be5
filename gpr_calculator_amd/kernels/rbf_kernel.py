"""kee_C / kef_C / kff_C of the RBF kernel with the reference's signatures and return values
(gpr_calc/kernels/rbf_kernel.py:7-337).  The reference marshals numpy -> cffi -> CPU natives here;
this module packs the rows on the device and runs the HIP blocks instead.  Results come back as
numpy arrays, like the reference's."""
import numpy as np
import torch

from .. import _lib
from .. import device as dev
from ._common import as_energy_tuple, as_force_tuple


def _kp(sigma, l, zeta, tol=0.0):
    return dev.kernel_params(_lib.KERNEL_RBF, zeta, sigma, l, 0.0, tol)


def _eset(X, role):
    x, ele, counts = as_energy_tuple(X)
    return dev.PackedSet(role, x, None, ele, counts)


def _fsets(X, role, stress=False):
    x, dxdr, ele, counts = as_force_tuple(X, stress=stress)
    base = dev.PackedSet(role, x, dxdr, ele, counts, col0=0)
    sets = [base]
    if stress:
        for c0 in (3, 6):
            sets.append(dev.PackedSet(role, base.x, base.dxdr, base.ele, counts, col0=c0))
    return sets


def kee_C(X1, X2, sigma=1.0, l=1.0, zeta=2.0, grad=False):
    s1, s2 = _eset(X1, _lib.ROLE_STREAM), _eset(X2, _lib.ROLE_RESIDENT)
    if grad:
        C, C_l = dev.block_ee(_kp(sigma, l, zeta), s1, s2, grad=True)
        C = C.cpu().numpy()
        return C, (2 / sigma) * C, C_l.cpu().numpy()
    return dev.block_ee(_kp(sigma, l, zeta), s1, s2).cpu().numpy()


def kef_C(X1, X2, sigma=1.0, l=1.0, zeta=2.0, grad=False, stress=False, transpose=False):
    s1 = _eset(X1, _lib.ROLE_STREAM)
    f2 = _fsets(X2, _lib.ROLE_RESIDENT, stress=stress)
    kp = _kp(sigma, l, zeta)
    m1, m2 = s1.npts, f2[0].npts
    if grad:
        C, C_l = dev.block_ef(kp, s1, f2[0], grad=True)
        C, C_l = C.cpu().numpy(), C_l.cpu().numpy()
        if transpose:
            C = C.T           # the reference transposes only C here (rbf_kernel.py:180-185)
        return C, (2 / sigma) * (C.T if transpose else C), C_l
    C = dev.block_ef(kp, s1, f2[0]).cpu().numpy()
    if stress:
        B1 = dev.block_ef(kp, s1, f2[1]).view(m1, m2, 3)
        B2 = dev.block_ef(kp, s1, f2[2]).view(m1, m2, 3)
        Cs = torch.cat((B1, B2), dim=2).reshape(m1, m2 * 6).cpu().numpy()
    else:
        Cs = np.zeros([m1, m2 * 6])
    if transpose:
        C, Cs = C.T, Cs.T
    return (C, Cs) if stress else C


def kff_C(X1, X2, sigma=1.0, l=1.0, zeta=2.0, grad=False, stress=False, diag=False, tol=1e-12):
    f1 = _fsets(X1, _lib.ROLE_STREAM, stress=stress)
    # set 2 is the (3-wide) training tuple in every reference call (gaussianprocess.py:875)
    x2, dx2, e2, c2 = as_force_tuple(X2)
    f2 = [dev.PackedSet(_lib.ROLE_RESIDENT, x2, np.ascontiguousarray(dx2[:, :, :3]), e2, c2)]
    m1, m2 = f1[0].npts, f2[0].npts
    if diag:
        raise NotImplementedError("rbf_kff_many_diag is not part of the reference's native interface "
                                  "(commented out of rbf_kernel.h:40-43)")
    if grad:
        C, C_l = dev.block_ff(_kp(sigma, l, zeta), f1[0], f2[0], grad=True)
        C = C.cpu().numpy()
        return C, (2 / sigma) * C, C_l.cpu().numpy()
    kp = _kp(sigma, l, zeta, tol)
    C = dev.block_ff(kp, f1[0], f2[0])
    if stress:
        parts = [dev.block_ff(kp, f1[t], f2[0]).view(m1, 3, m2 * 3) for t in (1, 2)]
        Cs = torch.cat(parts, dim=1).reshape(m1 * 6, m2 * 3)
        return C.cpu().numpy(), Cs.cpu().numpy()
    return C.cpu().numpy()
