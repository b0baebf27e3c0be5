"""RBF kernel object with the reference's protocol (gpr_calc/kernels/RBF_mb.py:7-524):
k(x_i, x_j) = sigma^2 exp(-(1 - d^zeta) / (2 l^2)), d the cosine of two descriptor rows.
Every block is evaluated by libgprhip.so on the GPU; there is no numpy fallback."""
import numpy as np

from .. import _lib
from .. import device as dev
from ._common import DeviceKernel, as_energy_tuple, as_force_tuple, n_points


class RBF_mb(DeviceKernel):
    def __init__(self, para=[1., 1.], bounds=[[1e-2, 5e+1], [1e-1, 1e+1]], zeta=2, ncpu=1, device='gpu'):
        super().__init__()
        self.name = 'RBF'
        self.bounds = bounds
        self.update(para)
        self.zeta = zeta
        self.device = device
        self.ncpu = ncpu

    def __str__(self):
        return "{:.5f}**2 *RBF({:.5f})".format(self.sigma, self.l)

    def load_from_dict(self, dict0):
        self.sigma = dict0["sigma"]
        self.l = dict0["l"]
        self.zeta = dict0["zeta"]
        self.bounds = dict0["bounds"]
        self.name = dict0["name"]

    def save_dict(self):
        return {"name": self.name, "sigma": self.sigma, "l": self.l, "zeta": self.zeta, "bounds": self.bounds}

    def parameters(self):
        return [self.sigma, self.l]

    def update(self, para):
        self.sigma, self.l = para[0], para[1]

    def _kp(self, tol=0.0):
        return dev.kernel_params(_lib.KERNEL_RBF, self.zeta, self.sigma, self.l, 0.0, tol)

    # ------------------------------------------------------------------ device-resident entry points
    def k_total_dev(self, data1, data2=None, f_tol=1e-10, transpose=False, comm=None):
        kp = self._kp(f_tol)
        return self._assemble(data1, data2, kp, kp, kp, transpose=transpose, comm=comm)

    def k_total_with_grad_dev(self, data1, f_tol=1e-10, comm=None):
        """(K, dK/dsigma, dK/dl, nE) on the device (RBF_mb.py:173-204; the gradient natives have
        no tol skip, rbf_kernel.cpp:534-536)."""
        kp = self._kp(f_tol)
        K, Kl, nE = self._assemble(data1, None, kp, kp, kp, grad=True, comm=comm)
        return K, K * (2.0 / self.sigma), Kl, nE

    def diag_dev(self, data):
        """prior variance of each predicted quantity (RBF_mb.py:62-133): energies through the numpy-eps
        cosine of K_ee_RBF (base.py:107-130), forces through kff_C(dat, dat) with its default tol 1e-12."""
        parts = []
        if "energy" in data and n_points(data["energy"]) > 0:
            parts.append(dev.diag_ee(self._kp(), self.pack_energy(data["energy"], flags=_lib.SET_NUMPY_EPS)))
        if "force" in data and n_points(data["force"]) > 0:
            parts.append(dev.diag_ff(self._kp(1e-12), self.pack_force(data["force"])))
        import torch
        return torch.cat(parts) if parts else None

    # ------------------------------------------------------------------ reference protocol (numpy out)
    def diag(self, data):
        out = self.diag_dev(data)
        return None if out is None else out.cpu().numpy()

    def k_total(self, data1, data2=None, f_tol=1e-10):
        return self.k_total_dev(data1, data2, f_tol).cpu().numpy()

    def k_total_with_grad(self, data1, f_tol=1e-10):
        K, Ks, Kl, _ = self.k_total_with_grad_dev(data1, f_tol)
        return K.cpu().numpy(), np.dstack((Ks.cpu().numpy(), Kl.cpu().numpy()))

    def k_total_with_stress(self, data1, data2, tol=1e-10):
        """(C, C1) with the stress rows of data1's force points (RBF_mb.py:206-229, marked obsolete
        upstream).  data1["force"] must carry 9-wide derivatives."""
        from .rbf_kernel import kee_C, kef_C, kff_C
        from .base import build_covariance
        C_ee = C_ef = C_fe = C_ff = C_se = C_sf = None
        for key1, d1 in data1.items():
            for key2, d2 in data2.items():
                if n_points(d1) > 0 and n_points(d2) > 0:
                    if key1 == 'energy' and key2 == 'energy':
                        C_ee = kee_C(d1, d2, self.sigma, self.l, self.zeta)
                    elif key1 == 'energy' and key2 == 'force':
                        C_ef = kef_C(d1, d2, self.sigma, self.l, self.zeta)
                    elif key1 == 'force' and key2 == 'energy':
                        C_fe, C_se = kef_C(d2, d1, self.sigma, self.l, self.zeta, stress=True, transpose=True)
                    elif key1 == 'force' and key2 == 'force':
                        C_ff, C_sf = kff_C(d1, d2, self.sigma, self.l, self.zeta, stress=True, tol=tol)
        return build_covariance(C_ee, C_ef, C_fe, C_ff), build_covariance(None, None, C_se, C_sf)
