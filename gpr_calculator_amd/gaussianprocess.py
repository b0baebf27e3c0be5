"""GP: Gaussian-process regression of energies and forces with the API of the reference's
gpr_calc/gaussianprocess.py (class GP, :22-1161), computing on MI355X.

What runs where
  * covariance blocks, noise, Cholesky, solves, means and variances: HIP kernels of libgprhip.so
    (device-resident K, L, alpha; nothing of that is evaluated with numpy / scipy);
  * training-set bookkeeping, queue counters, selection policy of add_structure, L-BFGS-B driver:
    host Python, same semantics as the reference;
  * ranks: one process per GPU over torch.distributed (parallel.Comm) where the reference uses
    mpi4py; the covariance assembly of fit() is sharded by training point -- every rank computes the row slab of its
    points in place in K, the slabs are exchanged in place (distributed.assemble_sharded) -- large factorisations run
    block-cyclic over the ranks, the weight-space fit all-reduces its F x F system, and predictions of a batch of
    structures are dealt to the ranks by structure and returned through one tensor all-reduce.

Differences kept on purpose (documented in DESIGN.md): the predictive variance is evaluated as
diag - |L^-1 k|^2 (triangular solve) instead of through an explicit K^-1; with the Dot kernel at zeta = 2 the fit,
the log-marginal likelihood and the prediction run in weight space (explicit 466-per-species feature map, F x F system
instead of N x N); `_K_inv`, `L_` and `alpha_` keep the reference's dense definitions and are factored / downloaded
lazily on first access.  Batched prediction (`predict_structures`, and `predict_structure` without stress) keeps
neighbour search, descriptor and regrouping on the device; for the weight-space model it is one native call
(csrc/predict.hip).
"""
import json
import logging
import os
import time
from copy import deepcopy

import numpy as np
import torch
from scipy.optimize import minimize

from . import device as dev
from .kernels.Dot_mb import Dot_mb
from .kernels.RBF_mb import RBF_mb
from .kernels._common import n_points
from .neighbors import fixed_indices, has_fixed_atoms
try:                                     # host-side marshalling helper (CPython extension built next to libgprhip.so)
    from . import _pygather
except ImportError:
    _pygather = None
from .parallel import Comm, multi
from .utilities import (convert_train_data, list_to_tuple, metric_values, new_pt, species_numbers,
                        tuple_to_list)


class _SessionTicket:
    """a batch handed to predict_structures_submit"""
    __slots__ = ("ses", "ticket", "keep", "out", "done")

    def __init__(self, ses, ticket, keep, out, done=None):
        self.ses, self.ticket, self.keep, self.out, self.done = ses, ticket, keep, out, done


class GP():
    """
    Gaussian Process Regressor class to fit the interatomic potential from reference
    energies / forces (same constructor and main APIs as gpr_calc.gaussianprocess.GP:22-118):
        - fit(): fit the GPR model
        - predict_structure(struc): energy / forces (/ std) of one structure
        - predict_structures(strucs): the same for a batch in one pass (extension)
        - add_structure((struc, energy, forces)): add a structure to the training queue
    """

    def __init__(self, kernel, descriptor, base_potential=None, noise_e=0.005, noise_f=0.1, f_coef=10,
                 log_file="gpr.log"):
        self.log_file = log_file
        logging.getLogger().handlers.clear()
        logging.basicConfig(level=logging.INFO, format='%(asctime)s| %(message)s', filename=self.log_file)
        self.logging = logging

        self.comm = Comm()
        self.rank, self.size = self.comm.rank, self.comm.size
        # model definition.  A list for noise_e means "optimise the noise too": [start, lower, upper] (its tail is the
        # L-BFGS-B bound); the force noise then follows as f_coef x the energy noise (gaussianprocess.py:64-75, 268-283)
        tunable = isinstance(noise_e, list)
        self.noise_e, self.noise_f = (noise_e[0], noise_f[0]) if tunable else (noise_e, noise_f)
        self.noise_bounds = noise_e[1:] if tunable else None
        self.f_coef = f_coef
        self.kernel, self.descriptor, self.base_potential = kernel, descriptor, base_potential
        # training set: nothing yet.  train_x / train_y / train_db are filled by set_train_pts; the *_queue counters count
        # points added since the last fit (they are invisible to predictions until then, get_train_x)
        self.x = self.train_x = self.train_y = self.train_db = None
        for counter in ("N_energy", "N_forces", "N_energy_queue", "N_forces_queue", "N_queue"):
            setattr(self, counter, 0)
        self.error = None                # metrics of the last validate_data(show=True)

        # device-resident model state
        self._dense = {"L": None, "alpha": None, "dinv": None}   # Cholesky factor of K + noise [N, N], (K + noise)^-1 y
        self._perm_dev = None       # after an appended fit: row order of the factor (position -> reference row), else None
        self._dense_params = None   # (hyper-parameters, noise) the dense factor was built with
        self._dense_pending = False  # a weight-space fit leaves the N x N factorisation until somebody asks for it
        self._fitted = False
        self._K_inv_dev = None
        self._Linv_dev = None
        self._ws_cache = None
        self._host = {}             # lazily downloaded numpy mirrors

        self.fits = 0
        self.use_base = 0
        self.use_surrogate = 0
        if self.rank == 0:
            self.logging.info(self)

    # ------------------------------------------------------------------ dense factor state (lazy after a weight-space fit)
    def _ensure_dense(self):
        if self._dense_pending:
            self._dense_pending = False
            self._factor_dense(lazy=True)

    @property
    def _L_dev(self):
        self._ensure_dense()
        return self._dense["L"]

    @_L_dev.setter
    def _L_dev(self, v):
        self._dense["L"] = v

    @property
    def _alpha_dev(self):
        self._ensure_dense()
        return self._dense["alpha"]

    @_alpha_dev.setter
    def _alpha_dev(self, v):
        self._dense["alpha"] = v

    @property
    def _dinv_dev(self):
        self._ensure_dense()
        return self._dense["dinv"]

    @_dinv_dev.setter
    def _dinv_dev(self, v):
        self._dense["dinv"] = v

    # ------------------------------------------------------------------ reference attribute mirrors
    @property
    def alpha_(self):
        if not self._fitted:
            return None
        if "alpha" not in self._host:
            self._host["alpha"] = self._alpha_dev.cpu().numpy()
        return self._host["alpha"]

    @property
    def L_(self):
        if not self._fitted:
            return None
        if "L" not in self._host:
            if self._perm_dev is not None:        # the appended factor is the Cholesky factor of a row permutation of
                self._factor_dense(lazy=True)     # K + noise: the reference's L_ is refactored when somebody asks for it
            self._host["L"] = self._L_dev.cpu().numpy()
        return self._host["L"]

    @property
    def _K_inv(self):
        if not self._fitted:
            return None
        self.set_K_inv()
        if "K_inv" not in self._host:
            self._host["K_inv"] = self._K_inv_dev.cpu().numpy()
        return self._host["K_inv"]

    def __str__(self):
        s = f"------Gaussian Process Regression ({self.rank}/{self.size})------\n"
        s += "Kernel: {:s}".format(str(self.kernel))
        if hasattr(self, "train_x"):
            s += " {:d} energy ({:.5f})".format(self.N_energy, self.noise_e)
            s += " {:d} forces ({:.5f})\n".format(self.N_forces, self.noise_f)
        if self.use_base > 0:
            N1, N2, N3 = self.use_base, self.use_surrogate, self.fits
            s += "Total base/surrogate/gpr_fit calls: {}/{}/{}\n".format(N1, N2, N3)
        return s

    def __repr__(self):
        return str(self)

    def todict(self):
        """Added for ASE compatibility (gaussianprocess.py:114-122)"""
        return {}

    def set_K_inv(self):
        """K^-1 = L^-T L^-1 (gaussianprocess.py:128-131), formed on the device only when asked for."""
        if self._K_inv_dev is None and self._L_dev is not None:
            Linv = self._Linv()
            Kinv = dev.xtx_lower(Linv)
            if self._perm_dev is not None:        # back to the reference's row order
                full = torch.empty_like(Kinv)
                full[self._perm_dev[:, None], self._perm_dev[None, :]] = Kinv
                Kinv = full
            self._K_inv_dev = Kinv

    # ------------------------------------------------------------------ covariance of the training set
    def _n_energy_points(self, train_x):
        e = train_x.get('energy', [])
        return n_points(e) if len(e) > 0 else 0

    def _train_K(self, grad=False):
        """K(train, train) on the device.  Over several ranks the kernel object shards the point-pair tasks
        and sums the partial matrices with one all-reduce (kernels/_common.py:_assemble)."""
        if grad:
            return self.kernel.k_total_with_grad_dev(self.train_x, comm=self.comm)
        return self.kernel.k_total_dev(self.train_x, comm=self.comm)

    def log_marginal_likelihood(self, params, eval_gradient=False, clone_kernel=False):
        """Log marginal likelihood and its gradient (gaussianprocess.py:133-202) evaluated on the
        device; only the scalars come back to the host for scipy's L-BFGS-B."""
        if self.noise_bounds is None:
            noise_e, noise_f = self.noise_e, self.noise_f
            kernel_params = params
        else:
            noise_e = params[-1]
            noise_f = self.f_coef * noise_e
            kernel_params = params[:-1]
        kernel = self.kernel
        kernel.update(kernel_params)

        if self.weight_space_fit:
            res = self._lml_weight_space(noise_e, noise_f, eval_gradient)
            if res is not None:
                if not eval_gradient:
                    return res[0]
                llg = res[1] if self.noise_bounds is not None else res[1][:-1]
                return res[0], llg

        NE = self._n_energy_points(self.train_x)
        grads_of = {}

        def build():
            if eval_gradient:
                K, grads_of["dK0"], grads_of["dK1"], _ = self._train_K(grad=True)
            else:
                K = self._train_K()
            dev.add_diag_(K, NE, noise_e ** 2, noise_f ** 2)
            return K
        K, info, dinv = dev.potrf_factor(build, self.logging, self.comm)     # K now holds L
        N = K.shape[0]
        dK0, dK1 = grads_of.get("dK0"), grads_of.get("dK1")
        if int(info.item()) != 0:
            return (-np.inf, np.zeros_like(params)) if eval_gradient else -np.inf
        L = K
        y = dev.to_dev(self.y_train)
        alpha = y.clone()
        dev.trsm_lower_(L, alpha, dinv=dinv)
        dev.trsm_lower_(L, alpha, trans=True, dinv=dinv)
        MLL = -0.5 * dev.frob_dot(y, alpha).item() - dev.sum_log_diag(L).item() - N / 2 * np.log(2 * np.pi)
        if not eval_gradient:
            return MLL
        # gradient: 0.5 * tr((alpha alpha^T - K^-1) dK/dtheta)   (GPML eq. 5.9; gaussianprocess.py:188-200)
        Linv = dev.trtri_lower(L)
        Kinv = dev.xtx_lower(Linv)
        grads = []
        for dK in (dK0, dK1):
            quad = dev.frob_dot(alpha, dev.gemm(dK, alpha)).item()
            grads.append(0.5 * (quad - dev.frob_dot(Kinv, dK).item()))
        # noise column: d(K+noise)/d(noise) = diag(2 noise_e | 2 noise_f)
        a2 = (alpha * alpha).reshape(-1) - torch.diagonal(Kinv)
        g_noise = 0.5 * (2 * noise_e * a2[:NE].sum().item() + 2 * noise_f * a2[NE:].sum().item())
        llg = np.array(grads + [g_noise])
        if self.noise_bounds is None:
            llg = llg[:-1]
        return MLL, llg

    def _lml_weight_space(self, noise_e, noise_f, eval_gradient):
        """(LML, gradient[sigma, sigma0, noise]) of a kernel with an explicit feature map through the F x F matrix
        G = I + Psi'^T D^-1 Psi' (Psi' = sigma Psi, D the noise diagonal) instead of the N x N factorisation:
            y^T (K+D)^-1 y = y^T D^-1 y - b^T G^-1 b,   log det(K+D) = log det D + log det G,   b = Psi'^T D^-1 y,
            Psi'^T alpha = G^-1 b,   Psi'^T (K+D)^-1 Psi' = I - G^-1,   diag (K+D)^-1 = D^-1 - D^-1 diag(Psi' G^-1 Psi'^T) D^-1
        with the same dK/dtheta as the dense route (dK/dsigma = 2 K / sigma; the constant 1.6 sigma^2 sigma0 energy
        block of dot_kernel.py:58; the noise column).  None when the kernel has no feature map for this data."""
        kernel = self.kernel
        if not hasattr(kernel, "_feature_path"):
            return None
        train_x = self.train_x
        species = kernel._feature_path(train_x, None, kernel.zeta)
        if species is None:
            return None
        sig, sig0 = float(kernel.sigma), float(kernel.sigma0)
        S, yt, rs, NE_loc = self._ws_share(train_x, species, noise_e, noise_f)      # this rank's rows
        N, NE, F = len(self.y_train), self._n_energy_points(train_x), S.shape[1]
        zeros = lambda *shape: torch.zeros(shape, dtype=dev.F64, device=S.device)
        have = S.shape[0] > 0
        b = dev.gemm(S, yt.reshape(-1, 1), transa=True) if have else zeros(F, 1)
        yy = (yt * yt).sum().reshape(1) if have else zeros(1)
        self._ws_sum(b, yy)

        def build():
            G = dev.gemm(S, S, transa=True) if have else zeros(F, F)
            self._ws_sum(G)
            G.diagonal().add_(1.0)
            return G
        G, info, dinv = dev.potrf_factor(build, self.logging, self.comm)
        if int(info.item()) != 0:
            return (-np.inf, np.zeros(3)) if eval_gradient else (-np.inf,)
        u = b.clone()
        dev.trsm_lower_(G, u, dinv=dinv)
        dev.trsm_lower_(G, u, trans=True, dinv=dinv)                      # G^-1 b = Psi'^T alpha
        quad = float(yy.item()) - dev.frob_dot(b, u).item()
        logdet = 2.0 * (NE * np.log(noise_e) + (N - NE) * np.log(noise_f)) + 2.0 * dev.sum_log_diag(G).item()
        MLL = -0.5 * quad - 0.5 * logdet - N / 2 * np.log(2 * np.pi)
        if not eval_gradient:
            return (MLL,)
        # every G^-1 quadratic form goes through C^-1 (G = C C^T): x^T G^-1 x = |C^-1 x|^2 keeps the cancellations below
        # (N_E / noise_e^2 - ..., 1 - ...) at working precision, which an explicit G^-1 (cond ~ 1e10) does not
        Ci = dev.trtri_lower(G)                                             # C^-1
        tr_Ginv = dev.frob_dot(Ci, Ci).item()
        g_sigma = (dev.frob_dot(u, u).item() - (F - tr_Ginv)) / sig
        # row sums over this rank's rows, then over ranks: Psi'^T D^-1 1_E, sum of alpha over the energy rows, and the
        # noise-column sums of alpha^2 - diag (K+D)^-1 over energy / force rows
        if have:
            alpha = rs * (yt - dev.gemm(S, u).reshape(-1))                 # (K+D)^-1 y, this rank's rows
            uE = (S[:NE_loc] * rs[:NE_loc, None]).sum(dim=0).reshape(-1, 1)
            W = dev.gemm(S, Ci, transb=True)                                # rows C^-1 s_i
            a2 = alpha * alpha - rs * rs * (1.0 - dev.rowdot(W, W))        # alpha^2 - diag (K+D)^-1
            sums = torch.stack((alpha[:NE_loc].sum(), a2[:NE_loc].sum(), a2[NE_loc:].sum()))
        else:
            uE, sums = zeros(F, 1), zeros(3)
        self._ws_sum(uE, sums)
        zE = dev.gemm(Ci, uE)
        one_Kinv_one = NE / noise_e ** 2 - dev.frob_dot(zE, zE).item()
        sums = sums.cpu().numpy()
        c0 = 0.8 * 2 * sig ** 2 * sig0                                      # dot_kernel.py:58 (sic)
        g_sigma0 = 0.5 * c0 * (float(sums[0]) ** 2 - one_Kinv_one)
        g_noise = 0.5 * (2 * noise_e * float(sums[1]) + 2 * noise_f * float(sums[2]))
        return MLL, np.array([g_sigma, g_sigma0, g_noise])

    def optimize(self, fun, theta0, bounds, maxiter=10):
        """scipy L-BFGS-B on the host, exactly the reference's call (gaussianprocess.py:204-220)"""
        opt_res = minimize(fun, theta0, method="L-BFGS-B", bounds=bounds, jac=True,
                           options={'maxiter': maxiter, 'ftol': 1e-2})
        return opt_res.x, opt_res.fun

    def fit(self, TrainData=None, show=True, opt=True, maxiter=10):
        """Fit the model (gaussianprocess.py:222-317): optional hyper-parameter optimisation, then
        K + noise -> L -> alpha, all on the device."""
        if TrainData is None:
            self.update_y_train()
        else:
            self.set_train_pts(TrainData)
        if self.rank == 0 and show:
            print(self)

        def obj_func(params, eval_gradient=True):
            if eval_gradient:
                lml, grad = self.log_marginal_likelihood(params, eval_gradient=True, clone_kernel=False)
                if show:
                    strs = "Loss: {:12.3f} ".format(-lml)
                    for para in params:
                        strs += "{:6.3f} ".format(para)
                    if self.rank == 0:
                        print(strs)
                        self.logging.info(strs)
                return (-lml, -grad)
            return -self.log_marginal_likelihood(params, clone_kernel=False)

        hyper_params = list(self.kernel.parameters())
        hyper_bounds = list(self.kernel.bounds)          # copy: the reference mutates kernel.bounds (:268-271)
        if self.noise_bounds is not None:
            hyper_params += [self.noise_e]
            hyper_bounds += [self.noise_bounds]

        if opt:
            if self.rank == 0 and show:
                print(f"Update GP model => {self.N_queue}/{maxiter}")
            params, _ = self.optimize(obj_func, hyper_params, hyper_bounds, maxiter=maxiter)
            params = self.comm.bcast(params, root=0)
            if self.noise_bounds is not None:
                self.kernel.update(params[:-1])
                self.noise_e = params[-1]
                self.noise_f = self.f_coef * params[-1]
            else:
                self.kernel.update(params)

        if not opt and self._try_append_dense():
            self._fit_train_x, self._fit_y = dict(self.train_x), self.y_train
            self.logging.info("Cholesky factor extended by the queued points")
            self.comm.barrier()
            self.N_energy_queue = self.N_forces_queue = self.N_queue = 0
            self.fits += 1
            return
        self._dense = {"L": None, "alpha": None, "dinv": None}
        self._perm_dev = None
        self._dense_pending = False
        self._K_inv_dev = None
        self._Linv_dev = None
        self._ws_cache = None
        self._drop_session()
        self._host = {}
        # what this fit describes: points queued later (add_structure replaces train_x / y_train by longer arrays, the
        # old ones stay alive here) must not leak into a dense factorisation that is only formed on first access
        self._fit_train_x, self._fit_y = dict(self.train_x), self.y_train
        if self.weight_space_fit and self._fit_weight_space():
            self._dense_pending = True          # L_, alpha_, _K_inv are factored on first access
            self.logging.info("Weight-space fit is complete")
        else:
            self._factor_dense()
            self.logging.info("Cholesky Decomp is Complete")
        self._fitted = True
        self.comm.barrier()

        self.N_energy_queue = 0
        self.N_forces_queue = 0
        self.N_queue = 0
        self.fits += 1

    def _factor_dense(self, lazy=False):
        """K + noise -> L -> alpha on the device (gaussianprocess.py:286-317) for the training set of the last fit.
        lazy: called from an attribute access after a weight-space fit -- possibly on one rank only, possibly after
        more points were queued: no collective, and the set / targets as they were at fit time."""
        train_x, y_train = self._fit_train_x, self._fit_y
        NE = self._n_energy_points(train_x)

        def build():
            K = self.kernel.k_total_dev(train_x, comm=None if lazy else self.comm)
            assert K.shape[0] == len(y_train), (K.shape, len(y_train))
            dev.add_diag_(K, NE, self.noise_e ** 2, self.noise_f ** 2)
            return K
        if not lazy and multi(self.comm) and len(y_train) >= self.dist_chol_min_n:
            # rows of K block-cyclic over the ranks, panels exchanged over RCCL (distributed.py); L comes back replicated
            from .distributed import DeviceOps, potrf_distributed
            K, dinv, info = potrf_distributed(build(), self.comm, DeviceOps())
        else:
            K, info, dinv = dev.potrf_factor(build, self.logging, None if lazy else self.comm)
        if int(info.item()) != 0:
            raise np.linalg.LinAlgError(f"{int(info.item())}-th leading minor of K + noise is not positive definite")
        alpha = dev.to_dev(y_train).clone()
        dev.trsm_lower_(K, alpha, dinv=dinv)
        dev.trsm_lower_(K, alpha, trans=True, dinv=dinv)
        self._dense = {"L": K, "alpha": alpha, "dinv": dinv}
        self._perm_dev = None
        self._dense_params = (tuple(float(v) for v in self.kernel.parameters()), float(self.noise_e), float(self.noise_f))
        if lazy:
            self._Linv_dev = self._K_inv_dev = None
            self._host.pop("L", None)

    # fit(opt=False) after add_structure with unchanged hyper-parameters (gaussianprocess.py:381-425 appends the points,
    # :286-317 then rebuilds all of K and refactors it): K + noise only grows by the rows / columns of the queued points.
    # With the new rows ordered LAST,  K' = [[K, B], [B^T, C]]  has the factor  L' = [[L, 0], [X, L_s]],
    #   X^T = L^-1 B,   L_s L_s^T = C - X X^T,   L'^-1 = [[L^-1, 0], [-L_s^-1 X L^-1, L_s^-1]]:
    # k new columns cost N^2 k instead of N^3 / 3 + the N^2 covariance entries.  The reference orders rows as
    # [energies | forces], so new energy rows sit in the middle there: the factor is kept in "old rows, then new rows" order
    # with the permutation next to it (_perm_dev); alpha_ and _K_inv are handed out in the reference's order, L_ -- whose
    # definition depends on the order -- is refactored if somebody asks for it.
    incremental_fit = True

    def _try_append_dense(self):
        d = self._dense
        if (not self.incremental_fit or not self._fitted or self._dense_pending or d["L"] is None or self.N_queue <= 0
                or multi(self.comm) or getattr(self, "_fit_train_x", None) is None):
            return False
        if hasattr(self.kernel, "_feature_path") and self.kernel._feature_path(self.train_x, None, self.kernel.zeta) is not None:
            return False            # explicit feature map (Dot, zeta = 2): the F x F weight-space system is refitted anyway
        now = (tuple(float(v) for v in self.kernel.parameters()), float(self.noise_e), float(self.noise_f))
        if self._dense_params != now:
            return False
        old = self._fit_train_x
        nE0, nF0 = self._n_energy_points(old), (n_points(old["force"]) if len(old.get("force", [])) else 0)
        E, F = self.train_x["energy"], self.train_x["force"]
        nE1, nF1 = len(E[-1]), len(F[-1])
        if nE1 - nE0 != self.N_energy_queue or nF1 - nF0 != self.N_forces_queue or nE1 + nF1 == nE0 + nF0:
            return False
        from .distributed import slice_energy, slice_force
        new = {}
        if nE1 > nE0:
            new["energy"] = slice_energy(E, nE0, nE1)
        if nF1 > nF0:
            new["force"] = slice_force(F, nF0, nF1)
        old_view = {"energy": slice_energy(E, 0, nE0), "force": slice_force(F, 0, nF0)} if nE0 and nF0 else \
            ({"energy": slice_energy(E, 0, nE0)} if nE0 else {"force": slice_force(F, 0, nF0)})
        kern = self.kernel
        kt = (lambda a, b, **kw: kern.k_total_dev(a, b, 1e-10, **kw)) if kern.name == "RBF" else (lambda a, b, **kw: kern.k_total_dev(a, b, **kw))
        N0, k = nE0 + 3 * nF0, (nE1 - nE0) + 3 * (nF1 - nF0)
        D = d["L"].device
        B = kt(new, old_view, transpose=True)                            # K(old, new), old rows in reference order
        C = kt(new, None)
        dev.add_diag_(C, nE1 - nE0, self.noise_e ** 2, self.noise_f ** 2)
        perm0 = self._perm_dev if self._perm_dev is not None else torch.arange(N0, device=D)
        Xt = B.index_select(0, perm0).contiguous()                       # rows in the factor's order
        dev.trsm_lower_(d["L"], Xt, dinv=d["dinv"])                      # X^T = L^-1 B
        S, info, _ = dev.potrf_factor(lambda: C - dev.gemm(Xt, Xt, transa=True), self.logging)
        if int(info.item()) != 0:
            return False                                                 # not positive definite in finite precision: refit
        L1 = torch.zeros((N0 + k, N0 + k), dtype=dev.F64, device=D)
        L1[:N0, :N0] = d["L"]
        L1[N0:, :N0] = Xt.T
        L1[N0:, N0:] = S
        dinv = dev.tri_inv64(L1)
        # reference rows: [E_old | E_new | F_old | F_new]; old force rows move up by the new energies
        shift = nE1 - nE0
        p_old = torch.where(perm0 < nE0, perm0, perm0 + shift)
        p_new = torch.cat((torch.arange(nE0, nE1, device=D), torch.arange(nE1 + 3 * nF0, nE1 + 3 * nF1, device=D)))
        perm = torch.cat((p_old, p_new))
        y = dev.to_dev(self.y_train).index_select(0, perm).clone()
        dev.trsm_lower_(L1, y, dinv=dinv)
        dev.trsm_lower_(L1, y, trans=True, dinv=dinv)
        alpha = torch.empty_like(y)
        alpha[perm] = y
        Linv0 = self._Linv_dev
        self._dense = {"L": L1, "alpha": alpha, "dinv": dinv}
        self._perm_dev = perm
        self._K_inv_dev = None
        self._host = {}
        self._ws_cache = None
        self._drop_session()
        if Linv0 is not None:                                            # extend L^-1 too: -L_s^-1 X L^-1 | L_s^-1
            Ls_inv = torch.eye(k, dtype=dev.F64, device=D)
            dev.trsm_lower_(S, Ls_inv)
            Linv1 = torch.zeros_like(L1)
            Linv1[:N0, :N0] = Linv0
            Linv1[N0:, :N0] = -dev.gemm(Ls_inv, dev.gemm(Xt, Linv0, transa=True))
            Linv1[N0:, N0:] = Ls_inv
            self._Linv_dev = Linv1
        else:
            self._Linv_dev = None
        return True

    # kernels with an explicit feature map (Dot_mb at zeta = 2): K + D = Psi' Psi'^T + D with Psi' = sigma Psi [N, F],
    # F = 466 per species << N, so by the Woodbury identity everything a prediction needs comes from the F x F matrix
    #   G = I + Psi'^T D^-1 Psi':   w = sigma^2 Psi^T alpha = sigma G^-1 Psi'^T D^-1 y,
    #   M = sigma^4 Psi^T (K + D)^-1 Psi = sigma^2 (I - G^-1)
    # -- the same w and M the dense route derives from L and alpha (_weight_space), without forming K at all.
    weight_space_fit = True
    # several ranks: matrices from this size on are factored with the rows spread over them (GPR_DIST_CHOL_MIN_N overrides:
    # rehearsals of the block-cyclic route at small sizes)
    dist_chol_min_n = int(os.environ.get("GPR_DIST_CHOL_MIN_N", "16384"))

    def _ws_share(self, train_x, species, noise_e, noise_f):
        """This rank's share of the weight-space system: (S, y~, rs, NE_local) with S = D^-1/2 sigma Psi for the rows of
        the rank's energy points followed by those of its force points, y~ = D^-1/2 y, rs = D^-1/2.  One rank: all
        rows.  Everything the fit and the log-marginal likelihood need is a sum over rows (S^T S, S^T y~, ...), so the
        ranks exchange only F x F matrices, F-vectors and scalars (`_ws_sum`): the feature rows -- the expensive part,
        24.6 kflop per descriptor row -- are computed once, on the rank that owns the point."""
        kernel = self.kernel
        NE_all = self._n_energy_points(train_x)
        y_all = np.asarray(self.y_train).reshape(-1)
        if multi(self.comm):
            from .distributed import shard_points, slice_energy, slice_force
            E, Fc = train_x["energy"], train_x["force"]
            key = (id(E[0]), id(Fc[0]), self.comm.rank, self.comm.size)
            if getattr(self, "_ws_sub_key", None) != key:            # keep the slices alive: the pack cache keys on them
                e0, e1, f0, f1 = shard_points(E[-1], Fc[-1], self.comm.size)[self.comm.rank]
                sub = {}
                if e1 > e0:
                    sub["energy"] = slice_energy(E, e0, e1)
                if f1 > f0:
                    sub["force"] = slice_force(Fc, f0, f1)
                self._ws_sub_key, self._ws_sub = key, (sub, (e0, e1, f0, f1))
            sub, (e0, e1, f0, f1) = self._ws_sub
            y = np.concatenate((y_all[e0:e1], y_all[NE_all + 3 * f0:NE_all + 3 * f1]))
            NE = e1 - e0
        else:
            sub, y, NE = train_x, y_all, NE_all
        nfeat = len(species) * (kernel.feature_dim(train_x))
        if len(y) == 0:
            z = torch.zeros((0, nfeat), dtype=dev.F64, device=dev.device())
            return z, z[:, 0], z[:, 0], 0
        Psi = kernel._psi(sub, species)
        rs = torch.empty(Psi.shape[0], dtype=dev.F64, device=Psi.device)
        rs[:NE] = 1.0 / noise_e
        rs[NE:] = 1.0 / noise_f
        S = Psi * (float(kernel.sigma) * rs)[:, None]
        return S, dev.to_dev(y).reshape(-1) * rs, rs, NE

    def _ws_sum(self, *tensors):
        """sum over ranks, in place (one all-reduce per tensor; no-op with one rank)"""
        for t in tensors:
            self.comm.all_reduce_sum_(t)
        return tensors[0] if len(tensors) == 1 else tensors

    def _fit_weight_space(self):
        if not hasattr(self.kernel, "_feature_path"):
            return False
        train_x = self.train_x
        species = self.kernel._feature_path(train_x, None, self.kernel.zeta)
        if species is None:
            return False
        S, y, rs, NE = self._ws_share(train_x, species, self.noise_e, self.noise_f)    # rows of this rank
        F = S.shape[1]
        sig = float(self.kernel.sigma)
        b = (dev.gemv_t(S, y) if S.shape[0] else torch.zeros(F, dtype=dev.F64, device=S.device)).reshape(-1, 1)   # Psi'^T D^-1 y
        self._ws_sum(b)

        def build():
            G = dev.gemm(S, S, transa=True) if S.shape[0] else torch.zeros((F, F), dtype=dev.F64, device=S.device)
            self._ws_sum(G)
            G.diagonal().add_(1.0)
            return G
        G, info, dinv = dev.potrf_factor(build, self.logging, self.comm)
        if int(info.item()) != 0:
            return False
        dev.trsm_lower_(G, b, dinv=dinv)
        dev.trsm_lower_(G, b, trans=True, dinv=dinv)                              # G^-1 b
        Ci = dev.trtri_lower(G)                                                    # C^-1, G = C C^T
        # M = sigma^2 (I - G^-1) is formed from C^-1 only when the unfactored variance form asks for it (_ws_M)
        self._ws_cache = [species, None, b * sig, Ci]                     # Ci = C^-1 feeds the factored variance form
        return True

    def _ws_M(self, ws):
        """M = sigma^2 (I - G^-1) of the weight-space model; a weight-space fit leaves it to the first caller"""
        if ws[1] is None:
            sig, Ci = float(self.kernel.sigma), ws[3]
            M = dev.gemm(Ci, Ci, transa=True, alpha=-sig * sig)                   # -sigma^2 G^-1
            M.diagonal().add_(sig * sig)
            ws[1] = M
        return ws[1]

    # ------------------------------------------------------------------ prediction
    def _predict_dev(self, X, train_x, f_tol, return_std, want_flat=False):
        """mean [m] (and variance [m]) of the quantities in X on the device.
        K(train, X) = K_trans^T is built directly (structures streamed, training set resident)."""
        ws = self._weight_space()
        Kt = None
        if ws is not None and self.weight_space_mean and hasattr(self.kernel, "psi_diag_dev"):
            # Dot kernel at zeta = 2, batch form: features and prior variances straight from the descriptor
            # output; mean = Psi* w, var = diag - rowsum((Psi* M) o Psi*)
            pd = self.kernel.psi_diag_dev(X, ws[0], want_diag=return_std)
            if pd is not None:
                P, diag = pd
                out, flat = dev.ws_finish(P, ws[2], dev.gemm(P, self._ws_M(ws)) if return_std else None, diag, tail=1)
                res = (out[0], out[1] if return_std else None)
                return res + (flat,) if want_flat else res
        if ws is not None and self.weight_space_mean:
            # Dot kernel at zeta = 2: K_trans alpha = Psi* (sigma^2 Psi^T alpha) = Psi* w, w built once per fit
            mean = dev.gemm(self.kernel._psi(X, ws[0]), ws[2]).reshape(-1)
        else:
            if self.kernel.name == 'RBF':
                Kt = self.kernel.k_total_dev(X, train_x, f_tol, transpose=True)  # [N, m]
            else:
                Kt = self.kernel.k_total_dev(X, train_x, transpose=True)
            mean = dev.gemv_t(Kt, self._alpha_dev)                                # K_trans alpha
        if not return_std:
            return (mean, None, None) if want_flat else (mean, None)
        var = self.kernel.diag_dev(X)
        if ws is not None:
            # Dot kernel at zeta = 2: k = sigma^2 Psi psi, so |L^-1 k|^2 = psi^T M psi with the F x F matrix
            # M = (L^-1 sigma^2 Psi)^T (L^-1 sigma^2 Psi) built once per fit: no N-sized work per prediction
            species, M = ws[0], self._ws_M(ws)
            P = self.kernel._psi(X, species)
            var = var - dev.rowdot(dev.gemm(P, M), P)
        else:
            if Kt is None:
                Kt = self.kernel.k_total_dev(X, train_x, transpose=True)
            Ktp = Kt if self._perm_dev is None else Kt.index_select(0, self._perm_dev)
            var = var - dev.trmm_lower_sumsq(self._Linv(), Ktp)                   # |L^-1 K_trans^T|^2 per column
        var = torch.clamp_min(var, 0.0)
        return (mean, var, None) if want_flat else (mean, var)

    # mean of the Dot feature path evaluated in weight space (Psi* w) instead of K_trans alpha: same number,
    # summed over features instead of training points
    weight_space_mean = True

    def _weight_space(self):
        """(species, M, w) for kernels with an explicit feature map (Dot_mb at zeta = 2), else None"""
        if not hasattr(self.kernel, "_feature_path"):
            return None
        if self._ws_cache is None:
            train_x = self.get_train_x()
            species = self.kernel._feature_path(train_x, None, self.kernel.zeta)
            if species is None:
                self._ws_cache = False
            else:
                Psi = self.kernel._psi(train_x, species)
                w = dev.gemm(Psi, self._alpha_dev, transa=True, alpha=self.kernel.sigma ** 2)     # [F, 1]
                W = Psi * (self.kernel.sigma ** 2)
                if self._perm_dev is not None:
                    W = W.index_select(0, self._perm_dev)
                dev.trsm_lower_(self._L_dev, W, dinv=self._dinv_dev)
                self._ws_cache = (species, dev.gemm(W, W, transa=True), w)
        return self._ws_cache or None

    def _Linv(self):
        """L^-1 on the device (lower triangular), built once per fit: the variance of every later
        prediction is then one triangular matrix product instead of a substitution sweep"""
        if self._Linv_dev is None:
            self._Linv_dev = dev.trtri_lower(self._L_dev)
        return self._Linv_dev

    def predict(self, X, stress=False, total_E=False, return_std=False, return_cov=False):
        """Internal predict function (gaussianprocess.py:319-379)."""
        if stress:
            raise NotImplementedError("GP.predict(stress=True) calls kernel.k_total_with_stress(X, train_x, same=False) "
                                      "in the reference, a signature its kernels do not accept (gaussianprocess.py:333)")
        train_x = self.get_train_x()
        mean, var = self._predict_dev(X, train_x, 1e-10, return_std or return_cov)
        y_mean = mean.cpu().numpy()

        Npts = 0
        if 'energy' in X:
            Npts += n_points(X["energy"])
        if 'force' in X:
            Npts += 3 * n_points(X["force"])
        factors = np.ones(Npts)
        if total_E:
            if isinstance(X["energy"], tuple):
                N_atoms = np.array([x for x in X["energy"][-1]])
            else:
                N_atoms = np.array([len(x[0]) for x in X["energy"]])
            factors[:len(N_atoms)] = N_atoms
        y_mean *= factors

        if return_cov:
            K_trans = self.kernel.k_total_dev(X, train_x)
            v = K_trans.T.contiguous()
            dev.trsm_lower_(self._L_dev, v)
            y_cov = self.kernel.k_total_dev(X) - dev.gemm(v, v, transa=True)
            return y_mean, y_cov.cpu().numpy()
        elif return_std:
            return y_mean, np.sqrt(var.cpu().numpy()) * factors
        return y_mean

    def set_train_pts(self, data, mode="w"):
        """Set / append the training points (gaussianprocess.py:381-425)."""
        if mode == "w" or self.train_x is None:
            self.train_x = {'energy': [], 'force': []}
            self.train_y = {'energy': [], 'force': []}
            self.train_db = []

        N_E, N_F = 0, 0
        for d in data["db"]:
            (atoms, energy, force, energy_in, force_in) = d
            if energy_in:
                e_id = deepcopy(N_E + 1)
                N_E += 1
            else:
                e_id = None
            if len(force_in) > 0:
                f_ids = [N_F + i for i in range(len(force_in))]
                N_F += len(force_in)
            else:
                f_ids = []
            self.train_db.append((atoms, energy, force, energy_in, force_in, e_id, f_ids))

        for key in data.keys():
            if key == 'energy':
                if len(data[key]) > 0:
                    self.add_train_pts_energy(data[key])
            elif key == 'force':
                if len(data[key]) > 0:
                    self.add_train_pts_force(data[key])

        self.update_y_train()
        self.N_energy += N_E
        self.N_forces += N_F
        self.N_energy_queue += N_E
        self.N_forces_queue += N_F
        self.N_queue += N_E + N_F

    def remove_train_pts(self, e_ids, f_ids):
        """Delete training points and refit (gaussianprocess.py:427-464)."""
        data = {"energy": [], "force": [], "db": []}
        energy_data = tuple_to_list(self.train_x['energy'], mode='energy')
        force_data = tuple_to_list(self.train_x['force'])
        for id in range(len(energy_data)):
            if id not in e_ids:
                (X, ele) = energy_data[id]
                data['energy'].append((X, self.train_y['energy'][id], ele))
        for id in range(len(force_data)):
            if id not in f_ids:
                (X, dxdr, ele) = force_data[id]
                data["force"].append((X, dxdr, self.train_y['force'][id], ele))
        for d in self.train_db:
            (atoms, energy, force, energy_in, force_in, e_id, _f_ids) = d
            if e_id in e_ids:
                energy_in = False
            _force_in = [force_in[i] for i, f_id in enumerate(_f_ids) if f_id not in f_ids]
            if energy_in or len(_force_in) > 0:
                data['db'].append((atoms, energy, force, energy_in, _force_in))
        self.N_energy = self.N_forces = 0
        self.N_energy_queue = self.N_forces_queue = self.N_queue = 0
        self.set_train_pts(data)
        self.fit()

    def compute_base_potential(self, atoms):
        return self.base_potential.calculate(atoms)

    def update_y_train(self):
        """train_y -> [N, 1] column: energies then (fx, fy, fz) per force point (gaussianprocess.py:472-488)"""
        E = np.asarray(self.train_y["energy"], dtype=np.float64).reshape(-1)
        F = np.asarray(self.train_y["force"], dtype=np.float64).reshape(-1)
        self.y_train = np.concatenate((E, F)).reshape(-1, 1)

    def validate_data(self, test_data=None, total_E=False, return_std=False, show=False):
        """Validate a data set, by default the training set (gaussianprocess.py:490-535)."""
        if test_data is None:
            test_X_E = {"energy": self.train_x['energy']}
            test_X_F = {"force": self.train_x['force']}
            NE = len(test_X_E['energy'][-1])
            E = self.y_train[:NE].flatten()
            F = self.y_train[NE:].flatten()
        else:
            test_X_E = {"energy": [(data[0], data[2]) for data in test_data['energy']]}
            test_X_F = {"force": [(data[0], data[1], data[3]) for data in test_data['force']]}
            E = np.array([data[1] for data in test_data['energy']])
            F = np.array([data[2] for data in test_data['force']]).flatten()
        if total_E:
            for i in range(len(E)):
                E[i] *= len(test_X_E['energy'][i])

        E_Pred, E_std, F_Pred, F_std = None, None, None, None
        nE, nF = n_points(test_X_E['energy']), n_points(test_X_F['force'])
        if test_data is None and not return_std and not total_E:
            quick = self._train_predictions()
            if quick is not None:                # the training set itself, right after a fit: K alpha = y - D alpha
                E_Pred, F_Pred = quick[:nE] if nE > 0 else None, quick[nE:] if nF > 0 else None
                if show:
                    self.update_error(E, E_Pred, F, F_Pred)
                return E, E_Pred, F, F_Pred
        if return_std:
            if nE > 0:
                E_Pred, E_std = self.predict(test_X_E, total_E=total_E, return_std=True)
            if nF > 0:
                F_Pred, F_std = self.predict(test_X_F, return_std=True)
            if show:
                self.update_error(E, E_Pred, F, F_Pred)
            return E, E_Pred, E_std, F, F_Pred, F_std
        if nE > 0:
            E_Pred = self.predict(test_X_E, total_E=total_E)
        if nF > 0:
            F_Pred = self.predict(test_X_F)
        if show:
            self.update_error(E, E_Pred, F, F_Pred)
        return E, E_Pred, F, F_Pred

    def _train_predictions(self):
        """The model's mean at its own training points without building K(train, train) again: (K + D) alpha = y, so
        K alpha = y - D alpha with D = diag(noise_e^2 | noise_f^2) -- what `predict` would return for the training set
        (gaussianprocess.py:319-379 evaluated at X = train_x), up to the residual of the solve (~1e-12 of |y|).  The
        calculator validates on the training set after every refit (calculator.py:109): at a few thousand rows that was one
        more covariance assembly per refit.  None when the shortcut does not apply: points queued since the fit, a factor that
        has not been formed (weight-space fit: its own cheap route), hyper-parameters changed since the fit."""
        d = self._dense
        if not self._fitted or self.N_queue != 0 or self._dense_pending or d["alpha"] is None:
            return None
        now = (tuple(float(v) for v in self.kernel.parameters()), float(self.noise_e), float(self.noise_f))
        if self._dense_params != now or self._fit_y is not self.y_train:
            return None
        alpha = self.alpha_.reshape(-1)
        y = np.asarray(self.y_train, dtype=np.float64).reshape(-1)
        if alpha.shape != y.shape:
            return None
        NE = self._n_energy_points(self.train_x)
        out = y.copy()
        out[:NE] -= self.noise_e ** 2 * alpha[:NE]
        out[NE:] -= self.noise_f ** 2 * alpha[NE:]
        return out

    def update_error(self, E, E_Pred, F, F_Pred):
        e_r2, e_mae, e_rmse = metric_values(E, E_Pred)
        f_r2, f_mae, f_rmse = metric_values(F, F_Pred)
        self.error = {"energy_r2": e_r2, "energy_mae": e_mae, "energy_rmse": e_rmse,
                      "forces_r2": f_r2, "forces_mae": f_mae, "forces_rmse": f_rmse}
        if self.rank == 0:
            for key in self.error.keys():
                self.logging.info(f"{key:<12s}: {self.error[key]:.4f}")

    def get_train_x(self):
        """Training data the current factor was built from, i.e. without the queue
        (gaussianprocess.py:553-577)."""
        if self.N_queue > 0:
            train_x = {}
            (_X, _ELE, _indices) = self.train_x['energy']
            NE = self.N_energy - self.N_energy_queue
            if NE > 0:
                ids = sum(_indices[:NE])
                train_x['energy'] = (_X[:ids], _ELE[:ids], _indices[:NE])
            else:
                train_x['energy'] = (_X, _ELE, _indices)
            NF = self.N_forces - self.N_forces_queue
            (_X, _dXdR, _ELE, _indices) = self.train_x['force']
            if NF > 0:
                ids = sum(_indices[:NF])
                train_x['force'] = (_X[:ids], _dXdR[:ids], _ELE[:ids], _indices[:NF])
            else:
                train_x['force'] = (_X, _dXdR, _ELE, _indices)
            # keep the sliced views alive so that the kernel's pack cache can key on them
            key = (id(self.train_x['energy'][0]), id(self.train_x['force'][0]), NE, NF)
            if getattr(self, "_train_view_key", None) != key:
                self._train_view_key, self._train_view = key, train_x
            return self._train_view
        return self.train_x

    _grow = None                # backing arrays of the training rows: {key: array with spare capacity}

    def _grown(self, key, old, new):
        """`old` followed by `new` (rows), as a view of a backing array that grows geometrically: appending a structure to
        a cfg3-sized training set used to copy all 40 MB of it (16 ms per add_structure).  Earlier views stay valid -- rows
        are only ever written behind them -- and a training set that was replaced from outside simply starts a new array."""
        if self._grow is None:
            self._grow = {}
        buf = self._grow.get(key)
        n, k = len(old), len(new)
        leading = buf is not None and old.base is buf and (n == 0 or old.__array_interface__["data"][0] == buf.__array_interface__["data"][0])
        if not leading or n + k > len(buf) or buf.dtype != np.result_type(old, new) or buf.shape[1:] != old.shape[1:]:
            cap = max(int(1.5 * (n + k)) + 64, 1024)
            buf = np.empty((cap,) + tuple(old.shape[1:]), dtype=np.result_type(old, new))
            buf[:n] = old
            self._grow[key] = buf
        buf[n:n + k] = new
        return buf[:n + k]

    def add_train_pts_energy(self, energy_data):
        """append energy points (gaussianprocess.py:579-600)"""
        (X, ELE, indices, E) = list_to_tuple(energy_data, include_value=True, mode='energy')
        if len(self.train_x['energy']) == 3:
            (_X, _ELE, _indices) = self.train_x['energy']
            _X = self._grown("eX", np.asarray(_X), X)
            _indices = list(_indices) + list(indices)
            _ELE = self._grown("eELE", np.asarray(_ELE), ELE)
            self.train_x['energy'] = (_X, _ELE, _indices)
            self.train_y['energy'] = list(self.train_y['energy']) + list(E)
        else:
            self.train_x['energy'] = (X, ELE, indices)
            self.train_y['energy'] = E

    def add_train_pts_force(self, force_data):
        """append force points (gaussianprocess.py:602-629)"""
        (X, dXdR, ELE, indices, F) = list_to_tuple(force_data, include_value=True)
        if len(self.train_x['force']) == 4:
            (_X, _dXdR, _ELE, _indices) = self.train_x['force']
            _X = self._grown("fX", np.asarray(_X), X)
            _indices = list(_indices) + list(indices)
            _ELE = self._grown("fELE", np.asarray(_ELE), ELE)
            _dXdR = self._grown("fdXdR", np.asarray(_dXdR), dXdR)
            self.train_x['force'] = (_X, _dXdR, _ELE, _indices)
            self.train_y['force'] = list(self.train_y['force']) + list(F)
        else:
            self.train_x['force'] = (X, dXdR, ELE, indices)
            self.train_y['force'] = F

    # ------------------------------------------------------------------ structures
    def _get_fixed_atoms(self, struc):
        return list(fixed_indices(struc))

    def _structure_data(self, struc, d, stress=False):
        """the data dictionary predict_structure builds for one structure (gaussianprocess.py:846-870)"""
        ele = species_numbers(d['elements'])
        fix_ids = set(int(i) for i in self._get_fixed_atoms(struc))
        data = {"energy": list_to_tuple([(d['x'], ele)], mode='energy')}
        seq = d['seq']
        order = np.argsort(seq[:, 1], kind="stable")           # rows grouped by the atom that moves
        counts_all = np.bincount(seq[:, 1], minlength=len(struc))
        keep = np.array([i not in fix_ids for i in range(len(struc))])
        if stress:
            keep[:] = True
        row_keep = keep[seq[order, 1]]
        rows = order[row_keep]
        centres = seq[rows, 0]
        dxdr = d['dxdr'][rows]
        if stress:
            x_shape = d['x'].shape[1]
            rd = d['rdxdr'][rows].reshape(len(rows), x_shape, 9)[:, :, [0, 4, 8, 1, 2, 5]]
            dxdr = np.concatenate((dxdr, rd), axis=2)
        data["force"] = (d['x'][centres], dxdr, ele[centres], [int(c) for c in counts_all[keep]])
        free_ids = [i for i in range(len(struc)) if i not in fix_ids]
        return data, free_ids

    def predict_structure(self, struc, stress=True, return_std=False, f_tol=1e-8):
        """Energy, forces (, stress) and their standard deviations for one structure
        (gaussianprocess.py:834-918).  E is the total energy, E_std is per atom.

        Without stress (what the calculator asks for when the cell is fixed) a single structure takes the same
        device-resident route as predict_structures -- neighbour search, descriptor, regrouping and prediction
        without a host copy of x / dxdr; `device_single = False` selects the tuple route below, which is also the
        one for stress, for several ranks and for foreign descriptors."""
        if (not stress and self.device_single and not multi(self.comm)
                and hasattr(self.descriptor, "calculate_batch_dev")):
            Es, Fs, Es_std, Fs_std = self.predict_structures([struc], return_std=return_std, f_tol=f_tol)
            if return_std:
                return Es[0], Fs[0], None, Es_std[0], Fs_std[0]
            return Es[0], Fs[0], None
        d = self.descriptor.calculate(struc, use_mpi=True)
        data, free_ids = self._structure_data(struc, d, stress=stress)
        train_x = self.get_train_x()
        n = len(struc)
        S = None
        if stress:
            # force rows carry 9 columns; forces use the first three, stresses the other six
            force9 = data["force"]
            data_f3 = {"energy": data["energy"],
                       "force": (force9[0], np.ascontiguousarray(force9[1][:, :, :3]), force9[2], force9[3])}
            _, K_trans1 = self.kernel.k_total_with_stress(data, train_x, f_tol)
            S = dev.gemm(dev.to_dev(K_trans1), self._alpha_dev).reshape(n, 6).cpu().numpy()
            mean, var = self._predict_dev(data_f3, train_x, f_tol, return_std)
        else:
            mean, var = self._predict_dev(data, train_x, f_tol, return_std)
        y_mean = mean.cpu().numpy()
        y_mean[0] *= n
        E = y_mean[0]
        F = np.zeros((n, 3))
        F[free_ids] = y_mean[1:].reshape([len(free_ids), 3])

        if self.base_potential is not None:
            energy_off, force_off, stress_off = self.compute_base_potential(struc)
            E += energy_off
            F += force_off
            if stress:
                S += stress_off

        if return_std:
            y_std = np.sqrt(var.cpu().numpy())
            E_std = y_std[0]
            F_std = np.zeros((n, 3))
            F_std[free_ids] = y_std[1:].reshape([len(free_ids), 3])
            return E, F, S, E_std, F_std
        return E, F, S

    def _predict_batch_generic(self, mine_strucs, return_std, f_tol):
        """(mean, std, frees) of a batch through the kernel-object route (any kernel): device descriptor and
        regrouping, then K_trans / feature rows as the kernel decides"""
        X, frees = self._batch_data_dev(mine_strucs)
        mean, var, flat = self._predict_dev(X, self.get_train_x(), f_tol, return_std, want_flat=True)
        nbr = getattr(X["force"], "_nbr", None) if "force" in X else None
        if flat is not None:
            # [mean | var | status] share one buffer (weight-space path): one download, no extra synchronisation
            if nbr is not None:
                nbr.status_into(flat[-1:])
            host = flat.cpu().numpy()
            if nbr is not None and host[-1] != 0.0:
                raise RuntimeError(f"neighbour table is not symmetric: {int(host[-1])} force rows missing")
            m_ = mean.numel()
            mean, std = host[:m_], (np.sqrt(host[m_:2 * m_]) if return_std else None)
        else:
            if return_std:
                mean, std = mean.cpu().numpy(), np.sqrt(var.cpu().numpy())
            else:
                mean, std = mean.cpu().numpy(), None
            if nbr is not None:
                nbr.check()
        return mean, std, frees

    def _predict_batch_native(self, mine_strucs, return_std):
        """(mean, std, frees) through the one-call native pipeline gpr_predict_dot2_batch (csrc/predict.hip) when the
        model is a Dot kernel at zeta = 2 with its weight-space tables and the descriptor is the device SO3; None
        otherwise.  Same kernels as the generic route, queued from one host call."""
        if not self.native_batch or not hasattr(self.descriptor, "_device_tables"):
            return None
        tr = self._trace
        if tr is not None:
            tr.append(("native:enter", time.perf_counter_ns()))
        ws = self._weight_space()
        if ws is None or not self.weight_space_mean or getattr(self.descriptor, "_stress", False):
            return None
        from . import _lib
        import ctypes as C
        from .neighbors import batch_arrays
        des = self.descriptor
        if not des.derivative:
            return None
        # one pass over the structures in C when they expose plain arrays (csrc/pygather.c), the numpy route otherwise
        got = _pygather.gather(mine_strucs) if _pygather is not None else None
        if got is not None:
            a0 = np.frombuffer(got[0], dtype=np.int32)
            num = np.frombuffer(got[1], dtype=np.int32)
            pos = np.frombuffer(got[2], dtype=np.float64)
            cells = np.frombuffer(got[3], dtype=np.float64)
            pbc = np.frombuffer(got[4], dtype=np.int32)
            atom0, any_fixed = a0, got[5]
            cmask = got[6]
        else:
            atom0, numbers, pos, cells, pbc = batch_arrays(mine_strucs)
            a0 = atom0.astype(np.int32)
            num = numbers.astype(np.int32)
            any_fixed = any(has_fixed_atoms(s) for s in mine_strucs)
            cmask = None
        natoms = int(atom0[-1])
        mask, frees = None, None
        if any_fixed and cmask is not None:          # the gather pass has already marked the fixed atoms
            mask = np.frombuffer(cmask, dtype=np.uint8)
            if mask.all():
                mask = None
        elif any_fixed:
            mask = np.ones(natoms, dtype=np.uint8)
            for k, s in enumerate(mine_strucs):
                if has_fixed_atoms(s):
                    fix = self._get_fixed_atoms(s)
                    if len(fix):
                        mask[atom0[k] + np.asarray(fix, dtype=np.int64)] = 0
            if mask.all():
                mask = None
        if mask is None:
            frees = [None] * len(mine_strucs)
        nfree = natoms if mask is None else int(mask.sum())
        m = len(mine_strucs) + 3 * nfree
        table, nseg, ncheb = des._device_tables()[:3]
        species, w = ws[0], ws[2]
        Cinv = ws[3] if len(ws) > 3 else None
        M = None if (Cinv is not None or not return_std) else self._ws_M(ws)
        sp = np.ascontiguousarray(np.asarray(species, dtype=np.int32))
        ip = lambda a: a.ctypes.data_as(_lib.c_int_p)
        dp = lambda a: a.ctypes.data_as(_lib.c_double_p)
        nout = (2 if return_std else 1) * m
        # results land in page-locked host memory the device writes to directly (the tail kernel stores 2 m + 4 doubles):
        # after the stream has drained they are simply there -- no download call, no staging copy
        pin = self._pinned_out
        if pin is None or pin.numel() < nout + 4:
            pin = self._pinned_out = torch.empty(max(2 * (nout + 4), 4096), dtype=dev.F64, pin_memory=True)
        out = pin[:nout + 4]
        view = out.numpy()
        tailw = view[nout:nout + 4]
        # neighbour-table capacities guessed from the densest batch seen so far (pairs and seq rows per atom, 12 % spare):
        # with them the call queues everything without waiting for the device; too small a guess comes back flagged
        # and the batch is repeated with exact sizes
        hint = self._nbr_hint
        caps = (0, 0) if hint is None else (int(hint[0] * natoms * 1.125) + 64, int(hint[1] * natoms * 1.125) + 64)
        if tr is not None:
            tr.append(("native:marshalled", time.perf_counter_ns()))
        for attempt in range(2):
            tailw[:] = np.nan
            dev.check(_lib.lib().gpr_predict_dot2_batch(
                dev.stream_ptr(), des.nmax, des.lmax, float(des.rcut), float(des.alpha), nseg, ncheb, dev.ptr(table),
                len(mine_strucs), ip(a0), dp(pos), dp(cells), ip(pbc), ip(num), int(bool(des.weight_on)),
                None if mask is None else mask.ctypes.data_as(C.POINTER(C.c_ubyte)), len(sp), ip(sp),
                float(self.kernel.sigma), float(self.kernel.sigma0), dev.ptr(w), dev.ptr(M), dev.ptr(Cinv), int(bool(return_std)),
                caps[0], caps[1], dev.ptr(out), int(out.numel())), "gpr_predict_dot2_batch")
            # the status words are written by the last kernel of the call: poll them in host memory (a blocking stream
            # synchronisation sleeps and wakes tens of microseconds late), then confirm with a busy query of the stream
            if tr is not None:
                tr.append(("native:queued", time.perf_counter_ns()))
            polled = self._poll_results
            if polled:
                deadline = None
                while np.isnan(tailw).any():
                    if deadline is None:
                        deadline = time.perf_counter() + 2.0 + 1e-3 * len(mine_strucs)
                    elif time.perf_counter() > deadline:      # no answer in seconds: let the runtime report what happened
                        polled = False
                        if not GP._poll_warned:
                            GP._poll_warned = True
                            self.logging.warning("batched prediction: status words did not appear in host memory in time; "
                                                 "falling back to a stream synchronisation for this call")
                        break
            if not polled:
                torch.cuda.current_stream().synchronize()
            else:
                # the words say that the last kernel has issued all its stores; that every result row has LANDED in host
                # memory is what the queue's completion vouches for (seen once in a few hundred calls: a row behind the
                # words).  A busy query, no sleep.
                stream = torch.cuda.current_stream()
                while not stream.query():
                    pass
            if tr is not None:
                tr.append(("native:results", time.perf_counter_ns()))
            host = view.copy()                        # the pinned buffer is reused by the next call
            if host[nout + 1] == 0.0:
                break
            caps = (0, 0)
        if natoms:
            per_atom = (host[nout + 2] / natoms, host[nout + 3] / natoms)
            self._nbr_hint = per_atom if self._nbr_hint is None else (max(per_atom[0], self._nbr_hint[0]), max(per_atom[1], self._nbr_hint[1]))
        if host[nout] != 0.0:
            raise RuntimeError(f"neighbour table is not symmetric: {int(host[nout])} force rows missing")
        self._last_counts = (nfree == natoms, int(atom0[1] - atom0[0]) if len(mine_strucs) and natoms == len(mine_strucs) * int(atom0[1] - atom0[0])
                             and bool((np.diff(atom0) == atom0[1] - atom0[0]).all()) else -1)
        # for the caller's scatter of the force rows: atom offsets and the global indices of the moving atoms
        self._last_layout = (atom0, None if mask is None else np.flatnonzero(mask))
        res = host[:m], (np.sqrt(host[m:2 * m]) if return_std else None), frees
        if tr is not None:
            tr.append(("native:exit", time.perf_counter_ns()))
        return res

    def _session_submit(self, strucs, return_std, sync):
        """hand a batch to the prediction session (csrc/predict.hip): gather the structures' arrays in one C pass and
        either run the synchronous call (sync: returns the finished results) or queue it and return a ticket for
        _session_collect.  None when the model or the structures do not fit that route (the other routes take over)."""
        if (not self.native_batch or _pygather is None or self.base_potential is not None
                or not hasattr(self.descriptor, "_device_tables")):
            return None
        ws = self._weight_space()
        des = self.descriptor
        if ws is None or not self.weight_space_mean or getattr(des, "_stress", False) or not des.derivative:
            return None
        tr = self._trace
        if tr is not None:
            tr.append(("enter", time.perf_counter_ns()))
        got = _pygather.gather(strucs)
        if got is None:
            return None
        from . import _lib
        import ctypes as C
        # the session holds device pointers of ONE fitted model: it is tied to the weight-space cache object itself
        # (kept alive in the tuple, compared by identity -- an id() alone can be handed to the next model's list)
        key = (bool(return_std), float(self.kernel.sigma), float(self.kernel.sigma0))
        ses = self._session
        if ses is None or ses[3] is not ws or ses[0] != key:
            if ses is not None:
                _lib.lib().gpr_dot2_session_destroy(ses[1])
            table, nseg, ncheb = des._device_tables()[:3]
            species, w = ws[0], ws[2]
            Cinv = ws[3] if len(ws) > 3 else None
            M = None if (Cinv is not None or not return_std) else self._ws_M(ws)
            sp = np.ascontiguousarray(np.asarray(species, dtype=np.int32))
            h = C.c_void_p()
            dev.check(_lib.lib().gpr_dot2_session_create(
                C.byref(h), dev.stream_ptr(), des.nmax, des.lmax, float(des.rcut), float(des.alpha), nseg, ncheb, dev.ptr(table),
                int(bool(des.weight_on)), len(sp), sp.ctypes.data_as(_lib.c_int_p), float(self.kernel.sigma),
                float(self.kernel.sigma0), dev.ptr(w), dev.ptr(M), dev.ptr(Cinv), int(bool(return_std))), "gpr_dot2_session_create")
            ses = self._session = (key, h, (table, w, M, Cinv), ws)      # the tuple keeps the device arrays alive
            self._session_depth_set = None
        if self._session_depth_set != int(self.session_depth):           # follows the attribute, also on a live session
            dev.check(_lib.lib().gpr_dot2_session_set_depth(ses[1], int(self.session_depth)), "gpr_dot2_session_set_depth")
            self._session_depth_set = int(self.session_depth)
        # the gathered buffers go to the C-ABI by the addresses the gathering pass returns with them (five array views and their
        # .ctypes.data were a third of the host time before the call); `got` keeps them alive
        p_a0, p_num, p_pos, p_cells, p_pbc, natoms = got[7]
        a0 = np.frombuffer(got[0], dtype=np.int32)
        nS = len(strucs)
        fixed = None
        if got[5]:                                                       # some structure carries fixed atoms
            free = np.frombuffer(got[6], dtype=np.uint8)
            if not free.all():
                fixed = np.ascontiguousarray(free == 0).view(np.uint8)
        E, F = np.empty(nS), np.empty((natoms, 3))
        Es, Fs = (np.empty(nS), np.empty((natoms, 3))) if return_std else (None, None)
        vp = lambda a: None if a is None else a.__array_interface__["data"][0]
        if tr is not None:
            tr.append(("native call", time.perf_counter_ns()))
        if sync:
            dev.check(_lib.lib().gpr_dot2_session_predict(
                ses[1], nS, p_a0, p_pos, p_cells, p_pbc, p_num,
                vp(fixed), vp(E), vp(F), vp(Es), vp(Fs), None), "gpr_dot2_session_predict")
            if tr is not None:
                tr.append(("native done", time.perf_counter_ns()))
            return self._session_shape(a0, nS, natoms, return_std, E, F, Es, Fs)
        ticket = C.c_int(-1)
        dev.check(_lib.lib().gpr_dot2_session_submit(
            ses[1], nS, p_a0, p_pos, p_cells, p_pbc, p_num,
            vp(fixed), C.byref(ticket)), "gpr_dot2_session_submit")
        # the ticket keeps the gathered input buffers alive until the results are collected
        return _SessionTicket(ses, int(ticket.value), (got, fixed), (a0, nS, natoms, return_std, E, F, Es, Fs))

    @staticmethod
    def _session_shape(a0, nS, natoms, return_std, E, F, Es, Fs):
        n0 = int(a0[1]) if nS else 0
        none = [None] * nS
        if natoms == nS * n0:
            # same size everywhere (an NEB band): the arrays themselves, first axis = structure (64 views and 64 boxed
            # floats per array cost more host time than the unpacking in C)
            return E, F.reshape(nS, n0, 3), (Es if return_std else none), (Fs.reshape(nS, n0, 3) if return_std else none)
        Fl = [F[a0[k]:a0[k + 1]] for k in range(nS)]
        Fsl = [Fs[a0[k]:a0[k + 1]] for k in range(nS)] if return_std else none
        return list(E), Fl, (list(Es) if return_std else none), Fsl

    def _predict_batch_session(self, strucs, return_std):
        """(E, F, E_std, F_std) of a batch through the host-in / host-out session call gpr_dot2_session_predict"""
        return self._session_submit(strucs, return_std, sync=True)

    def predict_structures_submit(self, strucs, return_std=True):
        """Queue predict_structures(strucs) and return at once with a ticket; predict_structures_collect(ticket) hands out
        the results.  Up to GP.session_depth batches may be in flight: each has its own stream and result buffer
        (gpr_dot2_session_submit), so the device chains of consecutive batches overlap and the host prepares / unpacks one
        batch while the device works on the others -- the throughput form of the call (an image band evaluated batch
        after batch); same numbers as predict_structures, bit for bit.  Models or batches the session route does not take
        (other kernels, several ranks, a base potential) are evaluated on the spot and the ticket only carries the result."""
        if not multi(self.comm) and len(strucs) and self.session_batch:
            t = self._session_submit(strucs, return_std, sync=False)
            if t is not None:
                return t
        return _SessionTicket(None, -1, None, None, done=self.predict_structures(strucs, return_std=return_std))

    def predict_structures_collect(self, ticket):
        if ticket.done is not None:
            res, ticket.done = ticket.done, None
            return res
        ses, k, _, out = ticket.ses, ticket.ticket, ticket.keep, ticket.out
        if ses is None:
            raise RuntimeError("this ticket has been collected already")
        if self._session is not ses:
            raise RuntimeError("the model was refitted while this batch was in flight; its session is gone")
        from . import _lib
        a0, nS, natoms, return_std, E, F, Es, Fs = out
        vp = lambda a: None if a is None else a.__array_interface__["data"][0]
        dev.check(_lib.lib().gpr_dot2_session_wait(ses[1], k, vp(E), vp(F), vp(Es), vp(Fs), None), "gpr_dot2_session_wait")
        ticket.ses = ticket.keep = None
        return self._session_shape(a0, nS, natoms, return_std, E, F, Es, Fs)

    def _drop_session(self):
        """a (re)fit ends the prediction session of the previous model: its device pointers must never serve another"""
        ses, self._session = self._session, None
        if ses is not None:
            from . import _lib
            _lib.lib().gpr_dot2_session_destroy(ses[1])

    _session = None
    _trace = None               # a list: host timestamps (label, ns) of the native batch call are appended (tools/native_split.py)
    _nbr_hint = None
    _pinned_out = None
    _last_counts = None
    _last_layout = None
    _poll_results = True        # False: always synchronise the stream instead of polling the status words
    _poll_warned = False

    native_batch = True
    session_batch = True        # one rank: the host-in / host-out session call (False: the Python-side unpacking route)
    session_depth = 2           # batches predict_structures_submit may have in flight (1 .. 8)
    _session_depth_set = None   # the depth the live session was told
    device_single = True

    def predict_structures(self, strucs, return_std=True, f_tol=1e-8):
        """Batched predict_structure(stress=False): one neighbour search, one descriptor launch and one prediction
        (feature rows x weight-space tables for the Dot kernel at zeta = 2, K_trans + triangular product otherwise) for
        the whole batch; over ranks the batch is split by structure.
        Returns sequences (E, F, E_std, F_std) ordered like `strucs` (identical on every rank): lists, or -- a batch of
        equally sized structures on one rank -- arrays whose first axis is the structure."""
        nb = len(strucs)
        size, rank = self.comm.size, self.comm.rank
        many = multi(self.comm)          # several ranks (or the one-rank rehearsal of their code path: parallel.multi)
        if not many and nb and self.session_batch:
            fast = self._predict_batch_session(strucs, return_std)
            if fast is not None:
                return fast
        mine = list(range(rank, nb, size)) if size > 1 else list(range(nb))
        results = {}
        if mine:
            mine_strucs = strucs if size == 1 else [strucs[k] for k in mine]
            self._last_counts = None
            fast = self._predict_batch_native(mine_strucs, return_std)
            if fast is not None:
                mean, std, frees = fast
            else:
                mean, std, frees = self._predict_batch_generic(mine_strucs, return_std, f_tol)
            nS = len(mine)
            if self._last_counts is not None:       # the native call has already seen whether the batch is uniform
                all_free, n_uni = self._last_counts
                ns_atoms = [n_uni] * nS if n_uni >= 0 else [len(strucs[k]) for k in mine]
                uniform = all_free and n_uni >= 0 and self.base_potential is None
            else:
                ns_atoms = [len(strucs[k]) for k in mine]
                uniform = all(f is None for f in frees) and len(set(ns_atoms)) == 1 and self.base_potential is None
            if uniform:
                # same composition, nothing fixed, no base potential (an NEB band): slice the result block once
                n = ns_atoms[0]
                Eb = mean[:nS] * n
                Fb = mean[nS:].reshape(nS, n, 3)
                if not many:                     # one rank owns the whole batch: the slices are the answer
                    if return_std:
                        return list(Eb), list(Fb), list(std[:nS]), list(std[nS:].reshape(nS, n, 3))
                    return list(Eb), list(Fb), [None] * nS, [None] * nS
                if return_std:
                    Sb = std[nS:].reshape(nS, n, 3)
                    for s, k in enumerate(mine):
                        results[k] = (Eb[s], Fb[s], std[s], Sb[s])
                else:
                    for s, k in enumerate(mine):
                        results[k] = (Eb[s], Fb[s], None, None)
            layout = self._last_layout if (fast is not None and not uniform) else None
            if layout is not None:
                # ragged batch and / or fixed atoms: one scatter of all force rows into [natoms, 3], then per-structure views
                atom0, free_idx = layout
                natoms = int(atom0[-1])
                Fall = np.zeros((natoms, 3))
                Sall = np.zeros((natoms, 3)) if return_std else None
                if free_idx is None:
                    Fall[:] = mean[nS:].reshape(natoms, 3)
                    if return_std:
                        Sall[:] = std[nS:].reshape(natoms, 3)
                else:
                    Fall[free_idx] = mean[nS:].reshape(-1, 3)
                    if return_std:
                        Sall[free_idx] = std[nS:].reshape(-1, 3)
                Eall = mean[:nS] * np.diff(atom0)
                for s, k in enumerate(mine):
                    a0, a1 = int(atom0[s]), int(atom0[s + 1])
                    E, F = Eall[s], Fall[a0:a1]
                    if self.base_potential is not None:
                        e_off, f_off, _ = self.compute_base_potential(strucs[k])
                        E, F = E + e_off, F + f_off
                    results[k] = (E, F, std[s], Sall[a0:a1]) if return_std else (E, F, None, None)
            off = nS
            for s, k in enumerate(mine if (not uniform and layout is None) else ()):
                n, free_ids = len(strucs[k]), frees[s]
                nf = 3 * (n if free_ids is None else len(free_ids))
                E = mean[s] * n
                if free_ids is None:
                    F = mean[off:off + nf].reshape(n, 3).copy()
                else:
                    F = np.zeros((n, 3))
                    F[free_ids] = mean[off:off + nf].reshape(-1, 3)
                if self.base_potential is not None:
                    e_off, f_off, _ = self.compute_base_potential(strucs[k])
                    E, F = E + e_off, F + f_off
                if return_std:
                    if free_ids is None:
                        F_std = std[off:off + nf].reshape(n, 3).copy()
                    else:
                        F_std = np.zeros((n, 3))
                        F_std[free_ids] = std[off:off + nf].reshape(-1, 3)
                    results[k] = (E, F, std[s], F_std)
                else:
                    results[k] = (E, F, None, None)
                off += nf
        if many:
            # every rank knows the whole batch, hence where each structure's numbers go in one flat vector
            # [E, E_std, F (3n), F_std (3n)]: the ranks fill their own structures and ONE tensor all-reduce (zeros
            # elsewhere, so the sum is exact) leaves the complete vector everywhere -- nothing is pickled
            ns = np.array([len(s) for s in strucs], dtype=np.int64)
            per = (2 + 6 * ns) if return_std else (1 + 3 * ns)
            off = np.concatenate(([0], np.cumsum(per)))
            flat = np.zeros(int(off[-1]))
            for k, (E, F, es, fs) in results.items():
                o, n = int(off[k]), int(ns[k])
                if return_std:
                    flat[o], flat[o + 1] = E, es
                    flat[o + 2:o + 2 + 3 * n] = np.asarray(F).reshape(-1)
                    flat[o + 2 + 3 * n:o + 2 + 6 * n] = np.asarray(fs).reshape(-1)
                else:
                    flat[o] = E
                    flat[o + 1:o + 1 + 3 * n] = np.asarray(F).reshape(-1)
            t = torch.from_numpy(flat).to(getattr(self.comm, "tensor_device", "cpu"))
            flat = self.comm.all_reduce_sum_(t).cpu().numpy()
            results = {}
            for k in range(nb):
                o, n = int(off[k]), int(ns[k])
                if return_std:
                    results[k] = (float(flat[o]), flat[o + 2:o + 2 + 3 * n].reshape(n, 3), float(flat[o + 1]),
                                  flat[o + 2 + 3 * n:o + 2 + 6 * n].reshape(n, 3))
                else:
                    results[k] = (float(flat[o]), flat[o + 1:o + 1 + 3 * n].reshape(n, 3), None, None)
        Es, Fs, Es_std, Fs_std = [], [], [], []
        for k in range(nb):
            E, F, es, fs = results[k]
            Es.append(E); Fs.append(F); Es_std.append(es); Fs_std.append(fs)
        return Es, Fs, Es_std, Fs_std

    def _batch_data_dev(self, strucs):
        """the predict_structure data dictionary of a batch of structures, packed on the device straight
        from the descriptor kernel's output (no host round trip of x / dxdr): energy points = all atoms of
        each structure; force points = for every free atom a the seq rows with seq[:, 1] == a
        (gaussianprocess.py:852-870)."""
        if not hasattr(self.descriptor, "calculate_batch_dev"):
            datas, frees = [], []
            for s in strucs:
                data, free_ids = self._structure_data(s, self.descriptor.calculate(s))
                datas.append(data); frees.append(free_ids)
            return {"energy": _concat_energy([dt["energy"] for dt in datas]),
                    "force": _concat_force([dt["force"] for dt in datas])}, frees
        # order matters for the host: every upload happens before the descriptor kernel is queued (a pageable
        # upload waits for the work already on the stream)
        nbr = self.descriptor.neighbor_batch_dev(strucs)
        atom0 = nbr.atom0
        natoms = int(atom0[-1])
        nper = np.diff(atom0)
        keep, frees = None, []
        for k, s in enumerate(strucs):
            fix = self._get_fixed_atoms(s) if has_fixed_atoms(s) else ()
            if len(fix):
                if keep is None:
                    keep = np.ones(natoms, dtype=bool)
                keep[atom0[k] + np.asarray(fix, dtype=np.int64)] = False
                fixset = set(int(i) for i in fix)
                frees.append([i for i in range(int(nper[k])) if i not in fixset])
            else:
                frees.append(None)                                  # all atoms free
        # force point of atom a = the seq rows (i, a) of every centre i that feels a, regrouped on the device
        # (gpr_nbr_force_rows) and gathered inside the consumer kernels: x / dxdr never leave the descriptor's layout
        counts_f, row_index, x_index, f_row0 = nbr.force_rows(None if keep is None else np.flatnonzero(keep))
        e_row0, ele = nbr.atom0_dev(), nbr.numbers_dev()
        desc = self.descriptor.calculate_batch_dev(strucs, nbr=nbr)
        E = dev.SetPair(desc["x"], None, ele, nper)
        E._row0_dev = e_row0
        F = dev.SetPair(desc["x"], desc["dxdr"], ele, counts_f, x_index=x_index, dxdr_index=row_index)
        F._row0_dev = f_row0
        F._nbr = nbr                                                # keeps the index owner alive with the set
        E.species_all = F.species_all = tuple(int(z) for z in np.unique(nbr.numbers))
        return {"energy": E, "force": F}, frees

    # ---- add_structure: what the reference's routine (gaussianprocess.py:921-1002) amounts to
    #   labels      the base potential's share is taken off the labels (on copies; upstream edits the caller's arrays);
    #   judgement   a fitted model predicts the structure's energy per atom and every force component with their standard
    #               deviations; a model that has never been fitted counts as "twice the tolerance uncertain" everywhere;
    #   energy      always queued;
    #   forces      atom a is a CANDIDATE when its largest force std exceeds tol_f or the flat force vector's entry number a
    #               (sic -- upstream indexes the flattened [3n] arrays with the atom id; the trained sets depend on it) misses
    #               its label by more than 1.5 tol_f; candidates are taken in atom order, each only if its descriptor is new
    #               among those taken so far (utilities.new_pt), until N_max are in;
    #   queue       one db record (atoms, E, F, True, atom ids) and the points go to set_train_pts("a+").
    # tol_e = tol_e_var * noise_e, tol_f = tol_f_var * noise_f.
    def _labels_without_base(self, atoms, energy, force):
        if self.base_potential is None:
            return energy, force, 0, np.zeros((len(atoms), 3))
        e_off, f_off, _ = self.compute_base_potential(atoms)
        return energy - e_off, force - f_off, e_off, f_off

    def _judge_structure(self, my_data, natoms, energy, force, tol_e, tol_f):
        """(E, E_pred, E_std, F, F_pred, F_std[n, 3]): labels per atom / flat, what the model says, how sure it is"""
        if not self._fitted:
            label_e, label_f = [energy / natoms], force.flatten()
            return label_e, label_e, 2 * tol_e, label_f, label_f, np.full((natoms, 3), 2 * tol_f)
        E, E_pred, E_std, F, F_pred, F_std = self.validate_data(my_data, return_std=True)
        return E, E_pred, E_std[0], F, F_pred, F_std.reshape((natoms, 3))

    @staticmethod
    def _pick_force_points(x_rows, ele_rows, F, F_pred, F_std, tol_f, N_max):
        """atom ids whose force environments are queued, in atom order"""
        natoms = len(F_std)
        flat_label, flat_pred = np.asarray(F, dtype=np.float64).reshape(-1), np.asarray(F_pred, dtype=np.float64).reshape(-1)
        unsure = F_std.max(axis=1) > tol_f
        off = np.abs(flat_label[:natoms] - flat_pred[:natoms]) > 1.5 * tol_f        # entry a of the FLAT vectors (see above)
        candidates = np.flatnonzero(unsure | off)
        if N_max == 0 and (len(candidates) == 0 or candidates[0] != 0):
            return []                    # upstream tests "N_max points are in" at the END of every atom's turn, with ==:
                                         # N_max = 0 stops after atom 0 unless atom 0 got in, and is never met again if it did
        taken, seen = [], []
        for a in candidates:
            here = (x_rows[a], ele_rows[a])
            if not seen or new_pt(here, seen):
                taken.append(int(a))
                seen.append(here)
            if len(taken) == N_max:
                break
        return taken

    # A fitted model judges the structure in ONE device pass (device_judge): neighbour search, descriptor, every atom as a force
    # point, means and standard deviations -- and only the force points that are picked (at most N_max of, e.g., 220) are brought
    # to the host in the reference's tuple layout.  The route through convert_train_data / validate_data builds all of them on
    # the host first (gaussianprocess.py:950-954); it stays for unfitted models, several ranks and foreign descriptors, and
    # tests/test_gpu_edge.py holds the two to each other.
    device_judge = True

    def _judge_on_device(self, atoms, energy, force):
        """((E, E_pred, E_std, F, F_pred, F_std), energy tuple, force_point(a) -> tuple) or None when this route does not apply"""
        des = self.descriptor
        if (not self.device_judge or not self._fitted or self.comm.size != 1 or not hasattr(des, "calculate_batch_dev")
                or not getattr(des, "derivative", True) or getattr(des, "_stress", False)):
            return None
        from .neighbors import SimpleAtoms, atoms_arrays
        numbers, pos, cell, pbc = atoms_arrays(atoms)
        n = len(numbers)
        bare = SimpleAtoms(numbers, pos, cell, pbc)               # no constraints: every atom is a candidate (utilities.py:97-129)
        X, _ = self._batch_data_dev([bare])
        mean, var, _ = self._predict_dev(X, self.get_train_x(), 1e-10, True, want_flat=True)
        Fp = X["force"]
        Fp._nbr.check()
        mean_h, std_h = mean.cpu().numpy(), np.sqrt(var.cpu().numpy())
        x_host = X["energy"].x.cpu().numpy()
        ele = np.asarray(numbers, dtype=np.int64)
        xi, ri = Fp.x_index.cpu().numpy().astype(np.int64), Fp.dxdr_index
        row0 = np.concatenate(([0], np.cumsum(np.asarray(Fp.counts, dtype=np.int64))))
        force = np.asarray(force, dtype=np.float64)

        def force_point(a):
            sl = slice(int(row0[a]), int(row0[a + 1]))
            rows_x = xi[sl]
            return (x_host[rows_x], Fp.dxdr.index_select(0, ri[sl].long()).cpu().numpy(), force[a], ele[rows_x])
        judged = ([energy / n], [float(mean_h[0])], float(std_h[0]), force.flatten(), mean_h[1:], std_h[1:].reshape(n, 3))
        return judged, (x_host, energy / n, ele), force_point

    def add_structure(self, data, N_max=20, tol_e_var=1.2, tol_f_var=1.2, add_force=True):
        """Queue training points from a labelled structure `(atoms, energy, forces)`; returns (points, how many,
        (E, E_pred, E_std, F, F_pred, F_std)) with the base potential's share added back to the energies and forces."""
        atoms, energy, force = data
        natoms = len(atoms)
        tol_e, tol_f = tol_e_var * self.noise_e, tol_f_var * self.noise_f
        energy, force, e_off, f_off = self._labels_without_base(atoms, energy, force)
        on_device = self._judge_on_device(atoms, energy, force)
        if on_device is not None:
            (E, E_pred, E_std, F, F_pred, F_std), energy_point, force_point = on_device
            energy_points = [energy_point]
        else:
            my_data = convert_train_data([(atoms, energy, force)], self.descriptor)
            E, E_pred, E_std, F, F_pred, F_std = self._judge_structure(my_data, natoms, energy, force, tol_e, tol_f)
            energy_points, force_point = my_data["energy"], (lambda a: my_data["force"][a])
        F, F_pred = np.asarray(F, dtype=np.float64).reshape(-1), np.asarray(F_pred, dtype=np.float64).reshape(-1)

        x_rows, _, ele_rows = energy_points[0]
        atoms_in = self._pick_force_points(x_rows, ele_rows, F, F_pred, F_std, tol_f, N_max) if add_force else []
        points = {"energy": energy_points, "force": [force_point(a) for a in atoms_in],
                  "db": [(atoms, energy, force, True, atoms_in)]}
        self.set_train_pts(points, mode="a+")
        back = np.asarray(f_off).reshape(-1)
        return points, 1 + len(atoms_in), (E[0] + e_off, E_pred[0] + e_off, E_std, F + back, F_pred + back, F_std)

    # ------------------------------------------------------------------ persistence (json + ASE db)
    def save(self, filename, db_filename, verbose=True):
        """json + ASE db with the reference's schema (gaussianprocess.py:632-646, 689-724)"""
        dict0 = self.save_dict(db_filename)
        with open(filename, "w") as fp:
            json.dump(dict0, fp, indent=4)
        self.export_ase_db(db_filename, permission="w")
        if verbose:
            print(f"save model to {filename} and {db_filename}")

    @classmethod
    def load(cls, filename, N_max=None, device='gpu'):
        with open(filename, "r") as fp:
            dict0 = json.load(fp)
        instance = cls.load_from_dict(dict0, device=device)
        instance.extract_db(dict0["db_filename"], N_max)
        if instance.rank == 0:
            print(f"load GP model from {filename}")
            print(instance)
        return instance

    def save_dict(self, db_filename):
        noise = {"energy": self.noise_e, "force": self.noise_f, "f_coef": self.f_coef, "bounds": self.noise_bounds}
        dict0 = {"noise": noise, "kernel": self.kernel.save_dict(), "descriptor": self.descriptor.save_dict(),
                 "db_filename": db_filename}
        if self.error is not None:
            dict0["error"] = self.error
        if self.base_potential is not None:
            dict0["base_potential"] = self.base_potential.save_dict()
        return dict0

    def export_ase_db(self, db_filename, permission="w"):
        """Training structures and labels as an ASE sqlite db with the reference's row layout
        (gaussianprocess.py:689-724): data = {energy, force, energy_in, force_in}, key-value pairs dft_energy (per
        atom, base potential added back) and dft_fmax.  Written by asedb.write_db (standard library; the file opens in
        ase.db).  permission: "w" resets the file, "a+" appends."""
        from .asedb import write_db
        records = []
        for _data in self.train_db:
            (struc, energy, force, energy_in, force_in, _, _) = _data
            actual_energy = deepcopy(energy)
            actual_forces = force.copy()
            if self.base_potential is not None:
                energy_off, force_off, _ = self.compute_base_potential(struc)
                actual_energy += energy_off
                actual_forces += force_off
            data = {"energy": energy, "force": np.asarray(force, dtype=np.float64), "energy_in": bool(energy_in),
                    "force_in": [int(i) for i in force_in]}
            kvp = {"dft_energy": float(actual_energy / len(force)), "dft_fmax": float(np.max(np.abs(actual_forces.flatten())))}
            if hasattr(struc, "set_constraint"):
                struc.set_constraint()          # the reference drops constraints before writing (:722)
            records.append((struc, data, kvp))
        write_db(db_filename, records, append=(permission != "w"))

    def extract_db(self, db_filename, N_max=None):
        """structures of an ASE db -> descriptors -> training points (gaussianprocess.py:726-821).  Every rank reads
        the file and computes the descriptors itself (one batched device launch) where the reference scatters rows
        over MPI ranks and gathers the points."""
        from .asedb import read_db
        rows = []
        for row in read_db(db_filename, limit=N_max):
            d = row["data"]
            rows.append((row["atoms"], d["energy"], np.array(d["force"], dtype=np.float64), bool(d["energy_in"]),
                         [int(i) for i in np.asarray(d["force_in"]).reshape(-1)]))
        if self.rank == 0:
            print(f"Rank 0: Loaded {len(rows)} structures")
        self.set_train_pts(self.points_from_structures(rows), "w")

    def points_from_structures(self, rows):
        """(atoms, energy, force, energy_in, force_in) records -> training-point dictionary
        (the loop body of gaussianprocess.py:779-799)"""
        pts = {"energy": [], "force": [], "db": []}
        for (atoms, energy, force, energy_in, force_in) in rows:
            d = self.descriptor.calculate(atoms)
            ele = species_numbers(d['elements'])
            if energy_in:
                pts["energy"].append((d['x'], energy / len(atoms), ele))
            for id in force_in:
                ids = np.argwhere(d['seq'][:, 1] == id).flatten()
                _i = d['seq'][ids, 0]
                pts["force"].append((d['x'][_i, :], d['dxdr'][ids], force[id], ele[_i]))
            pts["db"].append((atoms, energy, force, energy_in, force_in))
        return pts

    def sparsify(self, e_tol=1e-10, f_tol=1e-10):
        """Remove training points the covariance matrix does not need (gaussianprocess.py:1004-1023): CUR leverage
        scores of the energy block and of the force block of K; a force point goes when all three of its rows are
        selected.  K comes from the device; the symmetric eigen-decomposition runs on the host with numpy like the
        reference's (an offline maintenance call, not on the prediction / fit path)."""
        K = self.kernel.k_total(self.train_x)
        N_e = len(self.train_x["energy"][-1])
        N_f = len(self.train_x["force"][-1])
        pts_e = CUR(K[:N_e, :N_e], e_tol)
        pts = CUR(K[N_e:, N_e:], f_tol)
        pts_f = []
        if N_f > 1:
            chosen = set(int(p) for p in pts)
            pts_f = [i for i in range(N_f) if all(3 * i + c in chosen for c in range(3))]
        print("{:d} energy and {:d} forces will be removed".format(len(pts_e), len(pts_f)))
        if len(pts_e) + len(pts_f) > 0:
            self.remove_train_pts([int(p) for p in pts_e], pts_f)
        return list(pts_e), pts_f

    # ---- bootstrap (gaussianprocess.py:1025-1116).  Two ways in:
    #   a saved model (json_file exists): load it -- descriptors are recomputed from its db -- optionally replace noise levels
    #   and kernel by the arguments (`overwrite`), fit with hyper-parameter optimisation, form K^-1;
    #   from scratch: default kernel of the requested family, SO3(nmax, lmax, rcut), then train_images.
    # Kernel families and their starting hyper-parameters as upstream: "Dot" -> Dot_mb([2, 2.0]), RBF_mb([1.0, 0.1]) otherwise;
    # the overwrite branch decides the other way round ("RBF" -> RBF_mb, anything else Dot_mb) and only when the family changes.
    @staticmethod
    def _starting_kernel(family, zeta):
        return Dot_mb(para=[2, 2.0], zeta=zeta) if family == "Dot" else RBF_mb(para=[1.0, 0.1], zeta=zeta)

    @classmethod
    def set_GPR(cls, images, base, kernel='RBF', zeta=2.0, noise_e=0.002, noise_f=0.1, lmax=4, nmax=3, rcut=5.0,
                json_file=None, overwrite=False):
        """Model for a set of images: restored from `json_file` when that exists, trained on the images otherwise."""
        from .SO3 import SO3
        restore = json_file is not None and os.path.exists(json_file)
        if not restore:
            gp = cls(kernel=cls._starting_kernel(kernel, zeta), descriptor=SO3(nmax=nmax, lmax=lmax, rcut=rcut),
                     base_potential=None)
            gp.noise_e, gp.noise_f = noise_e, noise_f
            gp.train_images(images, base)
            return gp
        gp = cls.load(json_file)
        if overwrite:
            gp.noise_e, gp.noise_f = noise_e, noise_f
            if gp.kernel.name != kernel:
                gp.kernel = cls._starting_kernel("RBF" if kernel == "RBF" else "Dot", zeta)
        gp.fit()
        gp.set_K_inv()
        return gp

    def _label_image(self, image, base, tag):
        """energy and forces of one image from the base calculator (rank 0 only; the others get them by broadcast).
        Calculators that write files (ase's Vasp has `set`) get a directory of their own per image."""
        eng, forces = 0.0, None
        if self.rank == 0:
            image.calc = base
            if hasattr(base, 'set'):
                try:
                    base.set(directory=f"GP/calc_{tag}")
                except Exception:
                    pass
            eng, forces = image.get_potential_energy(), image.get_forces()
            print(f"Calculate E/F for image {tag}: {eng:.6f}")
        image.calc = None
        self.comm.barrier()
        return self.comm.bcast(eng, root=0), self.comm.bcast(forces, root=0)

    def train_images(self, images, base):
        """Label every image with the base calculator, queue what add_structure selects, then fit (with optimisation),
        validate on the training set and form K^-1."""
        for tag, image in enumerate(images):
            eng, forces = self._label_image(image, base, tag)
            self.add_structure((image.copy(), eng, forces))
        self.fit()
        self.validate_data()
        self.set_K_inv()

    @classmethod
    def load_from_dict(cls, dict0, device='gpu'):
        """(gaussianprocess.py:1118-1161)"""
        from .SO3 import SO3
        instance = cls(kernel=None, descriptor=None, base_potential=None)
        if dict0["kernel"]["name"] in ["RBF", "RBF_mb"]:
            instance.kernel = RBF_mb()
        elif dict0["kernel"]["name"] in ["Dot", "Dot_mb"]:
            instance.kernel = Dot_mb()
        else:
            raise NotImplementedError("unknown kernel {:s}".format(dict0["kernel"]["name"]))
        instance.kernel.load_from_dict(dict0["kernel"])
        # the shipped model stores "RBF_mb" as the name; GP branches on 'RBF' / 'Dot'
        instance.kernel.name = 'RBF' if isinstance(instance.kernel, RBF_mb) else 'Dot'
        if dict0["descriptor"]["_type"] == "SO3":
            instance.descriptor = SO3()
            instance.descriptor.load_from_dict(dict0["descriptor"])
        else:
            raise NotImplementedError("unknown descriptors {:s}".format(dict0["descriptor"]["_type"]))
        if "base_potential" in dict0.keys():
            if dict0["base_potential"]["name"] == "LJ":
                from .calculator import LJ
                instance.base_potential = LJ()
                instance.base_potential.load_from_dict(dict0["base_potential"])
            else:
                raise NotImplementedError("unknown base potential {:s}".format(dict0["base_potential"]["name"]))
        instance.kernel.device = device
        instance.noise_e = dict0["noise"]["energy"]
        instance.noise_f = dict0["noise"]["force"]
        instance.f_coef = dict0["noise"]["f_coef"]
        instance.noise_bounds = dict0["noise"]["bounds"]
        return instance


def CUR(K, l_tol=1e-10):
    """rows of a covariance block ranked by their weight in the near-null eigenvectors (eigenvalue < l_tol), the first
    N_low of them (gaussianprocess.py:1165-1182; Appendix D of Jinnouchi et al., Phys. Rev. B 100, 014105 (2019))"""
    if K.shape[0] == 0:
        return np.zeros(0, dtype=np.int64)
    L, U = np.linalg.eigh(K)
    low = L < l_tol
    omega = (U[:, low] ** 2).sum(axis=1)
    return np.argsort(-omega, kind="stable")[:int(low.sum())]


def _concat_energy(tuples):
    X = np.concatenate([t[0] for t in tuples], axis=0)
    ELE = np.concatenate([t[1] for t in tuples], axis=0)
    counts = [c for t in tuples for c in t[2]]
    return (X, ELE, counts)


def _concat_force(tuples):
    X = np.concatenate([t[0] for t in tuples], axis=0)
    D = np.concatenate([t[1] for t in tuples], axis=0)
    ELE = np.concatenate([t[2] for t in tuples], axis=0)
    counts = [c for t in tuples for c in t[3]]
    return (X, D, ELE, counts)
