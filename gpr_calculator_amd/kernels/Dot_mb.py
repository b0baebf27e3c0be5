"""Dot-product kernel object with the reference's protocol (gpr_calc/kernels/Dot_mb.py:5-173):
k(x_i, x_j) = sigma^2 (d^zeta + sigma0^2).  GPU only."""
import numpy as np
import torch

from .. import _lib
from .. import device as dev
from ._common import DeviceKernel, as_energy_tuple, as_force_tuple, n_points


class Dot_mb(DeviceKernel):
    def __init__(self, para=[1., 1.], bounds=[[1e-2, 5e+1], [1e-2, 1e+1]], zeta=3, device="gpu"):
        super().__init__()
        self.name = 'Dot'
        self.bounds = bounds
        self.update(para)
        self.zeta = zeta
        self.device = device

    def __str__(self):
        return "{:.3f}**2 *Dot({:.3f})".format(self.sigma, self.sigma0)

    def load_from_dict(self, dict0):
        self.sigma = dict0["sigma"]
        self.sigma0 = dict0["sigma0"]
        self.zeta = dict0["zeta"]
        self.bounds = dict0["bounds"]
        self.name = dict0["name"]

    def save_dict(self):
        return {"name": self.name, "sigma": self.sigma, "sigma0": self.sigma0, "zeta": self.zeta,
                "bounds": self.bounds}

    def parameters(self):
        return [self.sigma, self.sigma0]

    def update(self, para):
        self.sigma, self.sigma0 = para[0], para[1]

    def _kp(self, zeta):
        return dev.kernel_params(_lib.KERNEL_DOT, zeta, self.sigma, 1.0, self.sigma0, 0.0)

    # ------------------------------------------------------------------ device-resident entry points
    # ------------------------------------------------------------------ zeta = 2: explicit feature map
    use_feature_map = True      # set False to force the row-pair (MFMA block) kernels at zeta = 2 as well

    def _species_of(self, *datas):
        """sorted species over the energy / force entries of the given dictionaries (host ints)"""
        out = set()
        for data in datas:
            for key in ("energy", "force"):
                if key in data and n_points(data[key]) > 0:
                    ent = data[key]
                    ele = ent.ele if isinstance(ent, dev.SetPair) else \
                        (as_energy_tuple(ent)[1] if key == "energy" else as_force_tuple(ent)[2])
                    if isinstance(ele, dev.DevArray):
                        ele = ele.tensor()
                    if isinstance(ele, torch.Tensor):
                        out.update(torch.unique(ele).cpu().tolist())
                    else:
                        out.update(np.unique(np.asarray(ele)).tolist())
        return tuple(sorted(int(z) for z in out))

    def feature_dim(self, data):
        """features per species of the zeta = 2 map: d (d + 1) / 2 quadratic monomials + the constant"""
        if "force" in data and n_points(data["force"]) > 0:
            ent = data["force"]
            X = ent.x if isinstance(ent, dev.SetPair) else as_force_tuple(ent)[0]
        else:
            ent = data["energy"]
            X = ent.x if isinstance(ent, dev.SetPair) else as_energy_tuple(ent)[0]
        d = int(X.shape[1])
        return d * (d + 1) // 2 + 1

    def _psi(self, data, species):
        """Psi = [Psi_E ; Psi_F] of a data dictionary (csrc/features.hip), cached with the packed sets"""
        parts = []
        if "energy" in data and n_points(data["energy"]) > 0:
            pair = self.pack_energy(data["energy"])
            key = ("psiE", species, self.sigma0)
            if getattr(pair, "_psi_key", None) != key:
                pair._psi, pair._psi_key = dev.dot2_features(pair.stream, species, self.sigma0), key
            parts.append(pair._psi)
        if "force" in data and n_points(data["force"]) > 0:
            pair = self.pack_force(data["force"])
            key = ("psiF", species)
            if getattr(pair, "_psi_key", None) != key:
                pair._psi, pair._psi_key = dev.dot2_features(pair.stream, species, 0.0), key
            parts.append(pair._psi)
        return parts[0] if len(parts) == 1 else torch.cat(parts, dim=0)

    def psi_diag_dev(self, data, species, want_diag=True):
        """(Psi, diag) of a data dictionary whose entries are unpacked device SetPairs (the batch-prediction
        form): features and prior variances in one pass over the descriptor output, no packed sets
        (csrc/features.hip, gpr_dot2_features_direct).  None when an entry is not in that form."""
        ents = []
        for key in ("energy", "force"):
            if key in data and n_points(data[key]) > 0:
                ent = data[key]
                if not isinstance(ent, dev.SetPair) or ent.flags != 0:
                    return None
                if ent.dxdr is not None and int(ent.dxdr.shape[-1]) != 3:
                    return None
                ents.append((ent, ent.npts * (1 if ent.dxdr is None else 3), self.sigma0 if key == "energy" else 0.0))
        if not ents:
            return None
        d = int(ents[0][0].x.shape[1])
        F = len(species) * (d * (d + 1) // 2 + 1)
        rows = sum(e[1] for e in ents)
        P = torch.empty((rows, F), dtype=dev.F64, device=dev.device())      # both parts write into one buffer
        diag = torch.empty(rows, dtype=dev.F64, device=dev.device()) if want_diag else None
        r0 = 0
        for ent, n, s0 in ents:
            dev.dot2_features_direct(ent, species, self.sigma, s0, want_diag, psi_out=P[r0:r0 + n],
                                     diag_out=None if diag is None else diag[r0:r0 + n],
                                     species_all=getattr(ent, "species_all", None))
            r0 += n
        return P, diag

    def _feature_path(self, data1, data2, zeta_ee):
        if not self.use_feature_map or float(zeta_ee) != 2.0:
            return None
        if self.feature_dim(data1) > 32 * 33 // 2 + 1:                 # d > 32: the feature kernels pack 32 columns
            return None
        species = self._species_of(data1, data2 if data2 is not None else data1)
        return species if 1 <= len(species) <= 4 else None

    def k_total_dev(self, data1, data2=None, tol=1e-12, transpose=False, comm=None):
        """Dot_mb.k_total (Dot_mb.py:87-119).  The reference calls kef_C / kff_C with zeta in the
        sigma0 slot, so those blocks silently use zeta = 2.0 while K_ee uses self.zeta
        (SURVEY.md Appendix A.13); reproduced here.  With self.zeta == 2 the whole matrix is a degree-2
        polynomial kernel and is assembled as sigma^2 Psi Psi'^T by one GEMM (csrc/features.hip)."""
        species = self._feature_path(data1, data2, self.zeta)
        if species is not None:
            P1 = self._psi(data1, species)
            P2 = P1 if data2 is None else self._psi(data2, species)
            if transpose:
                return dev.gemm(P2, P1, transb=True, alpha=self.sigma ** 2)
            return dev.gemm(P1, P2, transb=True, alpha=self.sigma ** 2)
        return self._assemble(data1, data2, self._kp(self.zeta), self._kp(2.0), self._kp(2.0), transpose=transpose,
                              comm=comm)

    def k_total_with_grad_dev(self, data1, comm=None):
        """(K, dK/dsigma, dK/dsigma0, nE) (Dot_mb.py:121-148; all blocks use self.zeta here)."""
        kp = self._kp(self.zeta)
        species = self._feature_path(data1, None, self.zeta)
        if species is not None:
            P = self._psi(data1, species)
            K = dev.gemm(P, P, transb=True, alpha=self.sigma ** 2)
        else:
            K = self._assemble(data1, None, kp, kp, kp, comm=comm)
        nE = n_points(data1["energy"]) if "energy" in data1 else 0
        K2 = torch.zeros_like(K)
        K2[:nE, :nE] = 0.8 * 2 * self.sigma ** 2 * self.sigma0          # dot_kernel.py:58 (sic)
        return K, K * (2.0 / self.sigma), K2, nE

    def diag_dev(self, data):
        """Dot_mb.diag (Dot_mb.py:45-85): both parts go through the numpy helpers K_ee / K_ff, i.e. the
        numpy-eps convention (Dot_mb.py:177-258), with self.zeta."""
        parts = []
        kp = self._kp(self.zeta)
        if "energy" in data and n_points(data["energy"]) > 0:
            parts.append(dev.diag_ee(kp, self.pack_energy(data["energy"], flags=_lib.SET_NUMPY_EPS)))
        if "force" in data and n_points(data["force"]) > 0:
            parts.append(dev.diag_ff(kp, self.pack_force(data["force"], flags=_lib.SET_NUMPY_EPS)))
        return torch.cat(parts) if parts else None

    # ------------------------------------------------------------------ reference protocol (numpy out)
    def diag(self, data):
        out = self.diag_dev(data)
        return None if out is None else out.cpu().numpy()

    def k_total(self, data1, data2=None, tol=1e-12):
        return self.k_total_dev(data1, data2, tol).cpu().numpy()

    def k_total_with_grad(self, data1):
        K, K1, K2, _ = self.k_total_with_grad_dev(data1)
        return K.cpu().numpy(), np.dstack((K1.cpu().numpy(), K2.cpu().numpy()))

    def k_total_with_stress(self, data1, data2, tol=1e-10):
        """Dot_mb.k_total_with_stress (Dot_mb.py:150-173)"""
        from .dot_kernel import kee_C, kef_C, kff_C
        from .base import build_covariance
        s, s0, z = self.sigma, self.sigma0, self.zeta
        C_ee = C_ef = C_fe = C_ff = C_se = C_sf = None
        for key1, d1 in data1.items():
            for key2, d2 in data2.items():
                if n_points(d1) > 0 and n_points(d2) > 0:
                    if key1 == 'energy' and key2 == 'energy':
                        C_ee = kee_C(d1, d2, s, s0, z)
                    elif key1 == 'energy' and key2 == 'force':
                        C_ef = kef_C(d1, d2, s, z)
                    elif key1 == 'force' and key2 == 'energy':
                        C_fe, C_se = kef_C(d2, d1, s, z, stress=True, transpose=True)
                    elif key1 == 'force' and key2 == 'force':
                        C_ff, C_sf = kff_C(d1, d2, s, z, stress=True)
        return build_covariance(C_ee, C_ef, C_fe, C_ff), build_covariance(None, None, C_se, C_sf)
