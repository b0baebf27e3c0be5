"""Full-size golden vectors for BASELINE.json configs 2 and 3, produced by the reference's own Python + C++
(imported through refshim.py, as make_golden_gp.py does).  Build container only; takes tens of minutes.

    python tests/golden/make_golden_cfg.py cfg3 [--procs 6] [--n-train 2000]
    python tests/golden/make_golden_cfg.py cfg2 [--n-train 500]

cfg3 (headline): Dot kernel (sigma 2, sigma0 2, zeta 2), the training set bench.py builds (63 seeded 32-atom fcc
Pd cells, 2000 force environments + 63 energies, N = 6063), noise 0.002 / 0.1.  The reference's k_total over the
56 k x 56 k force rows is ~3 core-hours of dot_kff_many, so the training covariance is computed by the reference's
OWN Dot_mb.k_total on row chunks of the training points in --procs worker processes (what its MPI mode does,
RBF_mb.py:257-269, 471-481) and stacked; GP.fit(opt=False) then runs unmodified on a kernel proxy that hands this
matrix back for k_total(train_x) and delegates everything else to the reference kernel object.  Predictions:
GP.predict on the data dictionary of predict_structure (list form, see make_golden_gp.py) for three images of
synthetic.neb_images(64), one of them with atoms 0-7 fixed.
cfg2: RBF kernel (sigma 2.5, l 1.3, zeta 2), 16 cells, 500 force environments + 16 energies (N = 1516), everything
through the unmodified single-process reference: fit(opt=False), predict_structure x2, LML + gradient at theta.

Outputs (data only): tests/golden/gp_cfg3_dot.npz, tests/golden/gp_cfg2_rbf.npz
"""
import argparse
import math
import multiprocessing as mp
import os
import sys
import time
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)


def training_rows(n_train):
    from gpr_calculator_amd.synthetic import fcc_cell, lj_energy_forces
    from refshim import RefAtoms
    rows, left = [], n_train
    for s in range(math.ceil(n_train / 32)):
        a = fcc_cell(seed=s)
        a = RefAtoms(a.numbers, a.positions, a.cell, a.pbc)
        E, F = lj_energy_forces(a)
        take = min(32, left)
        rows.append((a, E, F, list(range(take))))
        left -= take
    return rows


def build_gp(ns, kind, para, rows, kernel_wrap=None):
    from make_golden_gp import train_points
    des = ns.SO3.SO3(nmax=3, lmax=4, rcut=5.0, alpha=2.0, derivative=True, stress=False)
    kern = (ns.RBF_mb.RBF_mb if kind == "RBF" else ns.Dot_mb.Dot_mb)(para=list(para), zeta=2.0)
    k = kernel_wrap(kern) if kernel_wrap else kern
    gp = ns.gaussianprocess.GP(kernel=k, descriptor=des, noise_e=0.002, noise_f=0.1, log_file="/tmp/ref_gp_cfg.log")
    pts = train_points(ns, des, [(a, E, F) for (a, E, F, _) in rows], [f for (_, _, _, f) in rows])
    gp.set_train_pts(pts)
    return gp, kern, des


_W = {}


def _worker_init(n_train):
    warnings.simplefilter("ignore")
    import refshim
    ns = refshim.import_reference()
    gp, kern, des = build_gp(ns, "Dot", [2.0, 2.0], training_rows(n_train))
    _W.update(ns=ns, gp=gp, kern=kern)


def _worker_rows(job):
    """rows of K owned by a chunk of training points, from the reference's own Dot_mb.k_total"""
    what, lo, hi = job
    ns, gp, kern = _W["ns"], _W["gp"], _W["kern"]
    tx = gp.train_x
    if what == "energy":
        lst = ns.utilities.tuple_to_list(tx["energy"], mode="energy")[lo:hi]
        return kern.k_total({"energy": lst}, tx)
    lst = ns.utilities.tuple_to_list(tx["force"])[lo:hi]
    return kern.k_total({"force": lst}, tx)


class CachedTrainK:
    """kernel proxy: k_total(train_x) with no second argument returns the pre-computed matrix (a copy: GP.fit
    adds the noise in place); everything else goes to the reference kernel object.  The last cross-covariance
    k_total(data1, data2) the GP asked for (K_trans of a prediction) is remembered in `last_cross`."""

    def __init__(self, kernel):
        object.__setattr__(self, "_k", kernel)
        object.__setattr__(self, "K", None)
        object.__setattr__(self, "last_cross", None)

    def __getattr__(self, name):
        if name == "k_total":
            def k_total(data1, data2=None, *a, **kw):
                if data2 is None and self.K is not None:
                    return self.K.copy()
                out = self._k.k_total(data1, data2, *a, **kw)
                if data2 is not None:
                    object.__setattr__(self, "last_cross", (data1, out))
                return out
            return k_total
        return getattr(self._k, name)

    def __setattr__(self, name, value):
        if name in ("K", "last_cross"):
            object.__setattr__(self, name, value)
        else:
            setattr(self._k, name, value)

    def __str__(self):
        return str(self._k)


def stable_variance_longdouble(L, K_trans, diag):
    """diag - |L^-1 k|^2 (the quantity gaussianprocess.py:903-907 forms through an explicit K^-1) from the reference's
    own L, K_trans and diag, forward substitution and sums in np.longdouble: the variance free of the cond(K) * eps
    noise of the explicit inverse, for the tests to hold the device against at 1e-6 / 1e-5"""
    L = np.asarray(L)
    n, m = L.shape[0], K_trans.shape[0]
    V = np.array(K_trans.T, dtype=np.longdouble)                   # [n, m], overwritten by L^-1 K_trans^T row by row
    Ld = L.astype(np.longdouble)
    for i in range(n):
        if i:
            V[i] -= Ld[i, :i] @ V[:i]
        V[i] /= Ld[i, i]
    return np.asarray(np.asarray(diag, dtype=np.longdouble) - (V * V).sum(axis=0), dtype=np.float64)


def record_std_inputs(p, key, L, K_trans, diag, std):
    """what pins sigma at full size: the prior variances, the long-double stable variance, three probe rows of the
    reference's K_trans (first, second, last), and how far the reference's recorded sigma^2 is from the stable one"""
    var = stable_variance_longdouble(L, K_trans, diag)
    p[key + "_diag"], p[key + "_var_stable"] = np.asarray(diag), var
    rows = np.array([0, 1, K_trans.shape[0] - 1])
    p[key + "_K_trans_rows"], p[key + "_K_trans_probe"] = rows, np.asarray(K_trans)[rows]
    rec = np.asarray(std, dtype=np.float64).reshape(-1) ** 2
    print(f"    stable variance: min {var.min():.3e} max {var.max():.3e}; recorded sigma^2 differs by up to "
          f"{np.abs(rec - np.clip(var, 0, None)).max():.3e}", flush=True)


def cfg3(args):
    import refshim
    from make_golden_gp import predict_lists
    from gpr_calculator_amd.synthetic import neb_images
    t0 = time.time()
    ns = refshim.import_reference()
    rows = training_rows(args.n_train)
    gp, kern, des = build_gp(ns, "Dot", [2.0, 2.0], rows, kernel_wrap=CachedTrainK)
    nE, nF = len(gp.train_x["energy"][-1]), len(gp.train_x["force"][-1])
    print(f"descriptors done {time.time() - t0:.0f} s: nE {nE} nF {nF} force rows {len(gp.train_x['force'][0])}", flush=True)
    # row chunks balanced by row count (cost of a chunk ~ its rows x all rows)
    cnt = np.asarray(gp.train_x["force"][-1])
    n_chunks = args.procs * 6
    edges = np.searchsorted(np.cumsum(cnt), np.linspace(0, cnt.sum(), n_chunks + 1)[1:-1]).tolist()
    bounds = [0] + [e + 1 for e in edges] + [nF]
    jobs = [("energy", 0, nE)] + [("force", bounds[i], bounds[i + 1]) for i in range(n_chunks) if bounds[i + 1] > bounds[i]]
    with mp.get_context("fork").Pool(args.procs, initializer=_worker_init, initargs=(args.n_train,)) as pool:
        parts = []
        for k, part in enumerate(pool.imap(_worker_rows, jobs)):
            parts.append(part)
            print(f"  K rows chunk {k + 1}/{len(jobs)} {part.shape} at {time.time() - t0:.0f} s", flush=True)
    K = np.vstack(parts)
    assert K.shape == (nE + 3 * nF, nE + 3 * nF)
    print(f"K assembled {K.shape}, asymmetry {np.abs(K - K.T).max():.2e} of {np.abs(K).max():.2e}", flush=True)
    gp.kernel.K = K
    gp.fit(show=False, opt=False)
    lml = gp.log_marginal_likelihood(np.array([2.0, 2.0]), eval_gradient=False)
    print(f"fit done {time.time() - t0:.0f} s, lml {lml}", flush=True)
    p = dict(kind="Dot", para=np.array([2.0, 2.0]), zeta=2.0, noise_e=0.002, noise_f=0.1, n_train=args.n_train,
             N=len(gp.y_train), y_train=gp.y_train, alpha=gp.alpha_, lml=lml, L_diag=np.diag(gp.L_).copy(),
             K_probe_rows=np.array([0, nE - 1, nE, nE + 1, nE + 3 * (nF // 2) + 2, nE + 3 * nF - 1]))
    p["K_probe"] = K[p["K_probe_rows"]]
    images = neb_images(64)
    tests = [(0, []), (37, []), (63, list(range(8)))]
    for k, (img, fixed) in enumerate(tests):
        a = images[img]
        t = refshim.RefAtoms(a.numbers, a.positions, a.cell, a.pbc, fixed=fixed)
        data, E, F, E_std, F_std = predict_lists(ns, gp, des, t, fixed)
        d1, K_trans = gp.kernel.last_cross
        diag = kern.diag({"energy": data["energy"], "force": list(data["force"])})
        free = [i for i in range(len(t)) if i not in set(fixed)]
        record_std_inputs(p, f"test{k}", gp.L_, K_trans, diag, np.concatenate(([E_std], F_std[free].reshape(-1))))
        p[f"test{k}_image"], p[f"test{k}_fixed"] = img, np.asarray(fixed, int)
        p[f"test{k}_positions"] = a.positions
        p[f"test{k}_E"], p[f"test{k}_F"], p[f"test{k}_E_std"], p[f"test{k}_F_std"] = E, F, E_std, F_std
        print(f"  test {k}: image {img} E {E:.6f} E_std {E_std:.3e} |F|max {np.abs(F).max():.4f} at {time.time() - t0:.0f} s",
              flush=True)
    p["n_tests"] = len(tests)
    path = os.path.join(HERE, "gp_cfg3_dot.npz" if args.n_train == 2000 else f"gp_cfg3_dot_n{args.n_train}.npz")
    np.savez_compressed(path, **p)
    print(path, os.path.getsize(path) // 1024, "KiB", f"{time.time() - t0:.0f} s")


def cfg2(args):
    import refshim
    from gpr_calculator_amd.synthetic import neb_images
    t0 = time.time()
    ns = refshim.import_reference()
    rows = training_rows(args.n_train)
    gp, kern, des = build_gp(ns, "RBF", [2.5, 1.3], rows, kernel_wrap=CachedTrainK)
    print(f"descriptors done {time.time() - t0:.0f} s", flush=True)
    gp.fit(show=False, opt=False)
    print(f"fit done {time.time() - t0:.0f} s", flush=True)
    nE, nF = len(gp.train_x["energy"][-1]), len(gp.train_x["force"][-1])
    p = dict(kind="RBF", para=np.array([2.5, 1.3]), zeta=2.0, noise_e=0.002, noise_f=0.1, n_train=args.n_train,
             N=len(gp.y_train), y_train=gp.y_train, alpha=gp.alpha_, L_diag=np.diag(gp.L_).copy())
    images = neb_images(64)
    tests = [(5, []), (50, list(range(8)))]
    for k, (img, fixed) in enumerate(tests):
        a = images[img]
        t = refshim.RefAtoms(a.numbers, a.positions, a.cell, a.pbc, fixed=fixed)
        E, F, S, E_std, F_std = gp.predict_structure(t, stress=False, return_std=True, f_tol=1e-12)
        d1, K_trans = gp.kernel.last_cross
        diag = kern.diag({"energy": d1["energy"], "force": list(d1["force"])})
        free = [i for i in range(len(t)) if i not in set(fixed)]
        record_std_inputs(p, f"test{k}", gp.L_, K_trans, diag, np.concatenate(([E_std], F_std[free].reshape(-1))))
        p[f"test{k}_image"], p[f"test{k}_fixed"], p[f"test{k}_positions"] = img, np.asarray(fixed, int), a.positions
        p[f"test{k}_E"], p[f"test{k}_F"], p[f"test{k}_E_std"], p[f"test{k}_F_std"] = E, F, E_std, F_std
        print(f"  test {k}: image {img} E {E:.6f} E_std {E_std:.3e} at {time.time() - t0:.0f} s", flush=True)
    p["n_tests"] = len(tests)
    path = os.path.join(HERE, "gp_cfg2_rbf.npz" if args.n_train == 500 else f"gp_cfg2_rbf_n{args.n_train}.npz")
    np.savez_compressed(path, **p)                                   # saved before the long gradient evaluation
    for k, th in enumerate(([2.5, 1.3], [3.2, 1.0])):
        lml, g = gp.log_marginal_likelihood(np.array(th), eval_gradient=True)
        p[f"lml{k}_theta"], p[f"lml{k}"], p[f"lml{k}_grad"] = np.array(th), lml, g
        print(f"  lml {k}: {lml} {g} at {time.time() - t0:.0f} s", flush=True)
        np.savez_compressed(path, **p)
    print(path, os.path.getsize(path) // 1024, "KiB", f"{time.time() - t0:.0f} s")


if __name__ == "__main__":
    warnings.simplefilter("ignore")
    ap = argparse.ArgumentParser()
    ap.add_argument("which", choices=["cfg2", "cfg3"])
    ap.add_argument("--procs", type=int, default=6)
    ap.add_argument("--n-train", type=int, default=None)
    a = ap.parse_args()
    if a.n_train is None:
        a.n_train = 2000 if a.which == "cfg3" else 500
    (cfg3 if a.which == "cfg3" else cfg2)(a)
