"""oracle/cpu_baseline.py -- TEST INFRASTRUCTURE ONLY (bench.py's `cpu_baseline` leg; never imported by the product).

The reference's CPU path for ONE predicted structure of the bench workload, timed on the host cores of the box and
extrapolated, where a part is too long to run in full, by its unit count:

    python -m oracle.cpu_baseline --kernel Dot --n-train 2000 --procs 16 --budget 20      -> one JSON object on stdout

  descriptor   SO3.calculate of one 32-atom structure: the oracle's numpy restatement (oracle/so3_numpy.py, pinned to the
               reference's own SO3 by tests/golden/so3_*.npz), one core -- the reference computes it serially per rank
  K_trans      the natives of k_total(data, train_x): kff / kef (force rows and energy rows of the structure against the
               training force rows) and kee / kef against the training energy rows.  The UNMODIFIED reference C++
               (oracle/_ref, kind "reference") when present, else the C restatement (kind "port").  One process per core
               on disjoint chunks of the structure's rows = the reference's MPI row chunking (RBF_mb.py:471-481).  kff is
               run against a slice of the training rows sized for `--budget` seconds and extrapolated by pair count.
  solve        K_trans alpha and the explicit-inverse variance of gaussianprocess.py:880, 903-907 in numpy at the real N
               (random matrices of the right shape: dense products do not depend on the values)
  fit          scipy cholesky at N plus the K assembly extrapolated from the measured kff rate (reported, not part of
               structures/s)

Everything here runs in fresh Python processes that never load the HIP runtime; bench.py starts it before it touches
the GPU.  Training descriptors are computed here too (oracle SO3, spread over the worker processes).
"""
import argparse
import json
import math
import multiprocessing as mp
import os
import sys
import time

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _structure(seed):
    from gpr_calculator_amd.synthetic import fcc_cell
    return fcc_cell(seed=seed)


def _descriptor(atoms):
    from oracle.so3_numpy import so3_calculate
    return so3_calculate(atoms.numbers, atoms.positions, atoms.cell, atoms.pbc, 3, 4, 5.0, 2.0)


def _force_rows(d, numbers, atom_ids):
    """(x rows, dxdr rows, species of the centre) of the force points of the given atoms, stacked (App. A.2)"""
    xs, dxs, es = [], [], []
    for a in atom_ids:
        ids = np.flatnonzero(d["seq"][:, 1] == a)
        ci = d["seq"][ids, 0]
        xs.append(d["x"][ci]); dxs.append(d["dxdr"][ids]); es.append(np.asarray(numbers)[ci])
    return np.concatenate(xs), np.concatenate(dxs), np.concatenate(es).astype(np.int32)


def _train_job(job):
    seed, take = job
    t0 = time.perf_counter()
    a = _structure(seed)
    d = _descriptor(a)
    x, dx, e = _force_rows(d, a.numbers, range(take))
    return x, dx, e, d["x"], np.asarray(a.numbers, dtype=np.int32), time.perf_counter() - t0


def _native_job(job):
    kind, what, is_dot, x1, dx1, e1, x2, dx2, e2 = job
    from oracle.natives import NativeSet
    ns = NativeSet("ref" if kind == "reference" else "port")
    c = np.ascontiguousarray
    n1, n2, d = len(x1), len(x2), x1.shape[1]
    i1, i2 = np.zeros(n1, np.int32), np.zeros(n2, np.int32)
    t0 = time.perf_counter()
    if what == "ff":
        out = np.zeros((1, 3, 1, 3))
        if is_dot:
            ns.call("dot_kff_many", n1, n2, 0, n2, d, 1, 2.0, c(x1), c(dx1), c(e1), i1, c(x2), c(dx2), c(e2), i2, out)
        else:
            ns.call("rbf_kff_many", n1, n2, 0, n2, d, 1, 2.0, 6.25, 1.69, 1e-8, c(x1), c(dx1), c(e1), i1, c(x2), c(dx2), c(e2), i2, out)
    elif what == "ef":          # energy rows x1 against force rows x2
        out = np.zeros((1, 1, 3))
        if is_dot:
            ns.call("dot_kef_many", n1, n2, d, 1, 2.0, c(x1), c(e1), i1, c(x2), c(dx2), c(e2), i2, out)
        else:
            ns.call("rbf_kef_many", n1, n2, d, 1, 2.0, 6.25, 1.69, c(x1), c(e1), i1, c(x2), c(dx2), c(e2), i2, out)
    else:                       # ee
        out = np.zeros((1, 1))
        if is_dot:
            ns.call("dot_kee_many", n1, n2, d, 1, 2.0, 4.0, 4.0, c(x1), c(e1), i1, c(x2), c(e2), i2, out)
        else:
            ns.call("rbf_kee_many", n1, n2, d, 1, 2.0, 6.25, 1.69, c(x1), c(e1), i1, c(x2), c(e2), i2, out)
    return time.perf_counter() - t0, float(out.sum())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kernel", default="Dot", choices=["Dot", "RBF"])
    ap.add_argument("--n-train", type=int, default=2000)
    ap.add_argument("--procs", type=int, default=max(1, min(16, os.cpu_count() or 1)))
    ap.add_argument("--budget", type=float, default=20.0, help="seconds of wall time for the force-force sample")
    args = ap.parse_args()
    from oracle.natives import NativeSet
    kind = "reference" if NativeSet.available("ref") else "port"
    is_dot = args.kernel == "Dot"
    P = args.procs
    t_all = time.perf_counter()
    with mp.get_context("spawn").Pool(P) as pool:
        # ---- training descriptors (inputs; oracle SO3 on every core)
        n_struc = math.ceil(args.n_train / 32)
        jobs = [(s, min(32, args.n_train - 32 * s)) for s in range(n_struc)]
        t0 = time.perf_counter()
        parts = pool.map(_train_job, jobs)
        t_train_desc = time.perf_counter() - t0
        tX, tdX, tE = (np.concatenate([p[k] for p in parts]) for k in range(3))
        eX = np.concatenate([p[3] for p in parts]); eE = np.concatenate([p[4] for p in parts])
        desc_core_s = float(np.mean([p[5] for p in parts]))                 # one structure, one core
        # ---- the predicted structure: descriptor on one core (serial in the reference), rows of its force points
        from gpr_calculator_amd.synthetic import neb_images
        atoms = neb_images(1)[0]
        t0 = time.perf_counter()
        d = _descriptor(atoms)
        t_desc = time.perf_counter() - t0
        Xf, dXf, Ef = _force_rows(d, atoms.numbers, range(len(atoms)))
        Xe, Ee = d["x"], np.asarray(atoms.numbers, dtype=np.int32)
        # ---- natives of K_trans: ff on a slice of the training rows, chunks of the structure's rows over the cores
        rate_guess = 0.26e6                                                   # row pairs / s / core
        n2 = int(min(len(tX), max(256, args.budget * rate_guess * P / len(Xf))))
        chunks = [c for c in np.array_split(np.arange(len(Xf)), P) if len(c)]
        t0 = time.perf_counter()
        pool.map(_native_job, [(kind, "ff", is_dot, Xf[c], dXf[c], Ef[c], tX[:n2], tdX[:n2], tE[:n2]) for c in chunks])
        t_ff = time.perf_counter() - t0
        ff_pairs, ff_full = float(len(Xf)) * n2, float(len(Xf)) * len(tX)
        t_ff_full = t_ff * ff_full / ff_pairs
        # energy row of the structure against all training force rows (kef), force rows against training energies (kef
        # transposed), energy against energies (kee): run in full, they are cheap
        chunks2 = [c for c in np.array_split(np.arange(len(tX)), P) if len(c)]
        t0 = time.perf_counter()
        pool.map(_native_job, [(kind, "ef", is_dot, Xe, None, Ee, tX[c], tdX[c], tE[c]) for c in chunks2])
        pool.map(_native_job, [(kind, "ef", is_dot, eX, None, eE, Xf[c], dXf[c], Ef[c]) for c in chunks])
        pool.map(_native_job, [(kind, "ee", is_dot, Xe, None, Ee, eX, None, eE)])
        t_small = time.perf_counter() - t0
    # ---- mean and explicit-inverse variance at the real N (numpy, all cores through its BLAS)
    nE, nF = n_struc, args.n_train
    N, m = nE + 3 * nF, 1 + 3 * len(atoms)
    rng = np.random.default_rng(0)
    Kt, Ki, al = rng.random((m, N)), rng.random((N, N)), rng.random((N, 1))
    t0 = time.perf_counter()
    _ = Kt.dot(al)
    _ = np.einsum("ij,ij->i", np.dot(Kt, Ki), Kt)
    t_solve = time.perf_counter() - t0
    # ---- fit side: Cholesky at N (scipy, as the reference), K assembly extrapolated from the measured ff rate
    from scipy.linalg import cholesky
    A = Ki @ Ki.T if N <= 8192 else None
    t_chol = None
    if A is not None:
        A[np.diag_indices(N)] += N
        t0 = time.perf_counter()
        cholesky(A, lower=True)
        t_chol = time.perf_counter() - t0
    ff_rate = ff_pairs / t_ff                                               # pairs / s on P cores
    t_assembly = float(len(tX)) ** 2 / ff_rate
    per_struct = t_desc + t_ff_full + t_small + t_solve
    out = {
        "value": 1.0 / per_struct, "unit": "structures/s", "cores": P, "kind": kind,
        "sample": (f"one 32-atom structure against the cfg3 training set ({len(tX)} force rows, N = {N}): descriptor "
                   f"{t_desc:.2f} s (oracle numpy SO3, 1 core) + force-force natives {t_ff_full:.1f} s extrapolated by pair "
                   f"count from {ff_pairs:.3g} of {ff_full:.3g} row pairs run in {t_ff:.1f} s on {P} processes + kef / kee "
                   f"natives in full {t_small:.2f} s + K_trans alpha and the explicit-inverse variance at N = {N} in numpy "
                   f"{t_solve:.2f} s"),
        "seconds_per_structure": {"descriptor": t_desc, "ff_natives_extrapolated": t_ff_full, "ef_ee_natives": t_small,
                                  "mean_variance": t_solve},
        "ff_pairs_per_s": ff_rate, "ff_pairs_per_s_per_core": ff_rate / P,
        "fit": {"K_assembly_s_extrapolated": t_assembly, "cholesky_s": t_chol,
                "note": "reference fit = k_total over all training row pairs (ff block extrapolated from the measured rate; "
                        "the reference computes both triangles) + scipy cholesky at N"},
        "inputs": {"training_descriptors_s": t_train_desc, "descriptor_s_per_structure_core": desc_core_s},
        "wall_s": time.perf_counter() - t_all,
    }
    print(json.dumps(out))


if __name__ == "__main__":
    main()
