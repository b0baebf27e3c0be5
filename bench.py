#!/usr/bin/env python3
"""bench.py -- headline benchmark of the GPR hot path on MI355X.

Metric (BASELINE.json): GPR predict E+F+sigma structures/sec (+ fit ms) at N_train = 2000.
Workload (BASELINE.json configs[2], SURVEY.md section 8d "cfg3"): 2000 training force environments
(+ 63 energy points) of seeded synthetic 32-atom fcc Pd cells, SO3(nmax=3, lmax=4, rcut=5), Dot-product
kernel with force derivatives; one step = predict energy, forces and both standard deviations for a
batch of 64 synthetic NEB images: neighbour search -> descriptor -> feature rows -> weight-space mean and
variance for the Dot kernel at zeta = 2 (descriptor -> K_trans -> K_trans alpha -> L^-1 K_trans^T -> variance
with --kernel RBF), inputs resident on the host as ASE-like structures, model resident in HBM, results
back on the host.

    python bench.py --gpus N --steps K --warmup W

N > 1: one rank per GPU under torch.distributed.run.  Started that way (WORLD_SIZE set) the script is a rank; started
plainly with --gpus N > 1 it launches `python -m torch.distributed.run --nproc-per-node N bench.py ...` as a CHILD process
before anything here touches the GPU and relays the child's JSON line and exit code (the parent never holds GPU state
and never re-executes itself).  The timed step is the throughput form of the call: batches are handed to the prediction
session with up to --depth of them in flight (gpr_dot2_session_submit / _wait), K submissions and K collections inside
the timed region; the synchronous one-batch latency (what an NEB optimiser step sees) is measured right after and
reported as config.sync_ms_per_step.

One JSON line on stdout (rank 0).  Extra objects: "roofline" for the kernel with the largest summed HIP-event
time of the timed region (plus the same figures for every profiled kernel of the step under "per_kernel" and,
for the Dot run, the same step through the row-pair kernels under "pair_route") and "cpu_baseline" (one structure of
the same predict workload through the reference's CPU path -- oracle descriptor, the reference's C++ natives from
oracle/_ref if present (else the oracle's C port) on a bounded sample, numpy mean / variance -- on the host cores,
oracle/cpu_baseline.py, run in child processes before this process touches the GPU).
`--mode fit` times fit(opt=False) instead (strong scaling over ranks).
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch

FP64_MFMA_PEAK_TFLOPS = 78.6     # 256 CU x 4 SIMD x 2.4 GHz x (2048 flop / 64 clk per v_mfma_f64_16x16x4_f64)
FLOP_PER_PAIR_FF = 1.10e3        # force-force work per same-species row pair
N_TRAIN = 2000
BATCH = 64


def build_model(kernel_name, n_train, comm):
    from gpr_calculator_amd.gaussianprocess import GP
    from gpr_calculator_amd.SO3 import SO3
    from gpr_calculator_amd.kernels.Dot_mb import Dot_mb
    from gpr_calculator_amd.kernels.RBF_mb import RBF_mb
    from gpr_calculator_amd.synthetic import fcc_cell, lj_energy_forces
    kern = Dot_mb(para=[2.0, 2.0], zeta=2.0) if kernel_name == "Dot" else RBF_mb(para=[2.5, 1.3], zeta=2.0)
    gp = GP(kernel=kern, descriptor=SO3(nmax=3, lmax=4, rcut=5.0, alpha=2.0), noise_e=0.002, noise_f=0.1,
            log_file=os.path.join("/tmp", f"gpr_bench_{os.getpid()}.log"))
    n_struc = math.ceil(n_train / 32)
    rows, left = [], n_train
    for s in range(n_struc):
        atoms = fcc_cell(seed=s)
        E, F = lj_energy_forces(atoms)
        take = min(32, left)
        rows.append((atoms, E, F, True, list(range(take))))
        left -= take
    gp.set_train_pts(gp.points_from_structures(rows))
    return gp


def time_fit(gp, warm=2, reps=3):
    """(cold, warm) ms of fit(opt=False): the first call of this route in the process (feature rows / packed sets built,
    temporaries mapped, look-ahead context created), then the mean of `reps` calls after `warm` more untimed ones"""
    def once():
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        gp.fit(show=False, opt=False)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) * 1e3
    cold = once()
    for _ in range(warm):
        once()
    return cold, float(np.mean([once() for _ in range(reps)]))


def same_species_pairs(ele1, ele2):
    z1, c1 = np.unique(ele1, return_counts=True)
    z2, c2 = np.unique(ele2, return_counts=True)
    d2 = dict(zip(z2.tolist(), c2.tolist()))
    return float(sum(c * d2.get(z, 0) for z, c in zip(z1.tolist(), c1.tolist())))


def cpu_baseline(kernel, n_train, budget_s=20.0):
    """The reference's CPU path on this box's host cores for one predicted structure of the same workload: oracle
    descriptor + the reference's C++ natives of K_trans (force-force block on a bounded sample, extrapolated by pair
    count) + mean / explicit-inverse variance in numpy (oracle/cpu_baseline.py).  Runs as a child process that never
    loads HIP, started BEFORE this process touches the GPU."""
    import subprocess
    cores = max(1, min(16, os.cpu_count() or 1))
    cmd = [sys.executable, "-m", "oracle.cpu_baseline", "--kernel", kernel, "--n-train", str(n_train), "--procs", str(cores),
           "--budget", str(budget_s)]
    try:
        res = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600)
        line = [l for l in res.stdout.splitlines() if l.startswith("{")][-1]
        return json.loads(line)
    except Exception as exc:                                  # the baseline must never take the GPU measurement down
        return {"value": None, "unit": "structures/s", "cores": cores, "kind": "unavailable", "sample": repr(exc)}


def fit_mode(args, comm, gp, dev):
    """--mode fit: the fit itself as the timed step (strong scaling: the same training set on N ranks).  One step =
    fit(opt=False): covariance row slabs of the rank's point shard + all-gather + Cholesky (block-cyclic over the ranks
    from GP.dist_chol_min_n rows on) + alpha on the dense route; per-rank feature rows + all-reduce of the F x F system
    on the weight-space route of the Dot kernel at zeta = 2."""
    rank, world = comm.rank, comm.size
    from gpr_calculator_amd.parallel import multi
    many = multi(comm)          # several ranks -- or their code path rehearsed with one (GPR_DIST_FORCE=1, tools/rccl_one_rank.py)
    results = {}
    for route, ws in (("dense", False), ("weight_space", True)):
        if ws and args.kernel != "Dot":
            continue
        gp.weight_space_fit = ws
        for _ in range(args.warmup):
            gp.fit(show=False, opt=False)
        comm.barrier(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            gp.fit(show=False, opt=False)
        torch.cuda.synchronize(); comm.barrier()
        el = time.perf_counter() - t0
        if many:
            t = torch.tensor([el], dtype=torch.float64, device="cuda")
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            el = float(t.item())
        results[route] = el / args.steps * 1e3
    if rank == 0:
        N = len(gp.y_train)
        R = len(gp.train_x["force"][0])
        head = "weight_space" if "weight_space" in results else "dense"
        tE = gp.train_x["force"][2]
        pairs = 0.5 * same_species_pairs(tE, tE)
        F = gp.kernel.feature_dim(gp.train_x) if args.kernel == "Dot" else None
        # algorithmic flop of each route (per fit, all ranks together)
        flop = {"dense": (pairs * FLOP_PER_PAIR_FF if F is None else float(N) * N * F) + N ** 3 / 3.0}
        unit = {"dense": ("1.10 kflop per same-species row pair of the lower triangle" if F is None else
                          "N^2 F (lower triangle of Psi Psi^T on the matrix cores)") + " + N^3 / 3 (Cholesky), all ranks"}
        if "weight_space" in results:
            flop["weight_space"] = float(N) * F * (F + 1) + F ** 3 / 3.0 + F ** 3 / 3.0
            unit["weight_space"] = "N F (F + 1) (Gram matrix Psi^T D Psi and right-hand side) + F^3 / 3 (Cholesky) + F^3 / 3 (triangular inverse)"
        roof = {r: {"ms": results[r], "flop": flop[r], "achieved": flop[r] / (results[r] * 1e-3) / 1e12 / world,
                    "frac": flop[r] / (results[r] * 1e-3) / 1e12 / world / FP64_MFMA_PEAK_TFLOPS, "work_unit": unit[r]} for r in results}
        line = {"metric": "GPR fit ms at N_train=%d" % args.n_train, "value": results[head], "unit": "ms", "n_gpus": world,
                "steps": args.steps, "warmup": args.warmup, "ms_per_step": results[head], "higher_is_better": False,
                "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
                "config": {"workload": f"fit(opt=False) of the cfg3-shaped training set: N_train={args.n_train} force environments "
                                       f"(N={N}, {R} force rows), {args.kernel} kernel", "kernel": args.kernel, "N": N,
                           "route": head, "fit_ms": results,
                           "parallelism": f"training points sharded x{world}: row slabs + all-gather"
                                          f"{'; block-cyclic Cholesky' if world > 1 and N >= gp.dist_chol_min_n else ''}"
                                          + ("; weight space: feature rows sharded, all-reduce of the F x F system" if head == "weight_space" else "")},
                "roofline": {"bound": "mfma", "unit": "TFLOP/s", "peak": FP64_MFMA_PEAK_TFLOPS, "achieved": roof[head]["achieved"],
                             "frac": roof[head]["frac"], "traffic": None,
                             "kernel": f"whole fit on the {head} route (wall time of fit(), not one kernel: launches, host work "
                                       f"and collectives included)", "work_unit": unit[head], "routes": roof},
                "cpu_baseline": None}
        print(json.dumps(line))


def launch_ranks(n):
    """--gpus N > 1 without a torchrun environment: start the N ranks as a child job and relay its JSON line.  Runs
    before this process has made any GPU call; the parent stays GPU-free for its whole life."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    res = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, text=True)
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    for l in res.stdout.splitlines():
        if not l.startswith("{"):
            print(l, file=sys.stderr)
    if lines:
        print(lines[-1])
    sys.stdout.flush()
    return res.returncode if res.returncode else (0 if lines else 1)


def dry_run(args):
    """what the launch path must get right, without a GPU: N ranks meet over the process group (gloo on a CPU-only box),
    agree on who is there, and rank 0 prints the JSON line the parent relays"""
    from gpr_calculator_amd.parallel import init_from_env
    comm = init_from_env()
    seen = comm.allreduce(np.eye(comm.size)[comm.rank])          # one tensor all-reduce: 1 at every rank that took part
    comm.barrier()
    if comm.rank == 0:
        print(json.dumps({"metric": "GPR predict E+F+sigma structures/sec at N_train=%d" % args.n_train, "value": None, "dry_run": True,
                          "n_gpus": comm.size, "steps": args.steps, "warmup": args.warmup,
                          "config": {"ranks": comm.size, "ranks_seen": [int(v) for v in np.atleast_1d(seen)],
                                     "backend": torch.distributed.get_backend() if comm.size > 1 else None}}))
    if comm.size > 1:
        torch.distributed.destroy_process_group()


def run_pipelined(gp, batch, steps, depth):
    """`steps` submissions and `steps` collections, at most `depth` batches in flight; returns the last results"""
    from collections import deque
    q, out = deque(), None
    for _ in range(steps):
        if len(q) == depth:
            out = gp.predict_structures_collect(q.popleft())
        q.append(gp.predict_structures_submit(batch, return_std=True))
    while q:
        out = gp.predict_structures_collect(q.popleft())
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--kernel", default="Dot", choices=["Dot", "RBF"])
    ap.add_argument("--n-train", type=int, default=N_TRAIN)
    ap.add_argument("--batch", type=int, default=BATCH)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pair-route", action="store_true", help="skip the extra pair-kernel measurement of the Dot run")
    ap.add_argument("--mode", default="predict", choices=["predict", "fit"])
    # four chains side by side (each slot stream has a hardware queue of its own up to four): 0.1235 ms per step sustained against
    # 0.1405 with two; between two synchronisations only 20 steps apart (what the driver times) the fill and drain of the deeper
    # pipeline eat most of that: 0.150 against 0.156 (profiles/r04_depth_sweep.log)
    ap.add_argument("--depth", type=int, default=4, help="batches in flight in the timed region (1 = the synchronous call)")
    ap.add_argument("--sustain", type=float, default=2.0, help="seconds of the untimed sustained leg after the timed region")
    ap.add_argument("--dry-run", action="store_true", help="ranks only rendezvous and report (no GPU work): checks the launch path")
    args = ap.parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus))
    if args.dry_run:
        return dry_run(args)
    # the CPU baseline first, in child processes, before this process loads the HIP runtime (N = 1, rank 0 only)
    cpu = None
    if args.mode == "predict" and int(os.environ.get("WORLD_SIZE", "1")) == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args.kernel, args.n_train)

    from gpr_calculator_amd import _lib, device as dev
    from gpr_calculator_amd.parallel import init_from_env
    from gpr_calculator_amd.synthetic import neb_images
    comm = init_from_env()
    rank, world = comm.rank, comm.size
    from gpr_calculator_amd.parallel import multi
    many = multi(comm)          # several ranks -- or their code path rehearsed with one (GPR_DIST_FORCE=1)
    if world == 1:
        torch.cuda.set_device(0)
    _lib.require_gpu()

    gp = build_model(args.kernel, args.n_train, comm)
    if many:
        assert gp.comm.size == world, (gp.comm.size, world)
    if args.mode == "fit":
        fit_mode(args, comm, gp, dev)
        if many:
            torch.distributed.destroy_process_group()
        return
    # fit(opt=False): with the Dot kernel at zeta = 2 the model is fitted in weight space (F x F system); the dense
    # route (K assembly sharded over ranks + N x N Cholesky + alpha) is what every other kernel takes and what
    # L_ / alpha_ / _K_inv trigger lazily -- both are timed
    gp.weight_space_fit = False
    fit_dense_cold, fit_dense_ms = time_fit(gp)
    gp.weight_space_fit = True
    fit_cold, fit_ms = time_fit(gp)
    # weak scaling: every rank predicts its own batch of images with the replicated model
    batch = neb_images(args.batch, seed=100 + 10 * rank)
    real_comm = gp.comm
    gp.comm = _Serial()      # predict_structures shards a batch over gp.comm; here each rank owns a whole batch
    depth = max(1, min(8, args.depth))
    gp.session_depth = depth
    pipelined = depth > 1 and args.kernel == "Dot"

    def step():
        return gp.predict_structures(batch, return_std=True, f_tol=1e-8)

    def run(n):
        if pipelined:
            return run_pipelined(gp, batch, n, depth)
        out = None
        for _ in range(n):
            out = step()
        return out

    # untimed: a few priming steps before the W warm-up steps the caller asked for (first call sizes the neighbour
    # tables exactly and records the capacities later calls guess from; clocks and the stream-ordered pool settle)
    n_prime = max(0, 30 - args.warmup) if args.kernel == "Dot" else 0
    for _ in range(n_prime):
        step()
    run(args.warmup)
    comm.barrier()
    torch.cuda.synchronize()
    # HIP events on the launch stream around every 4th launch of each kernel kind of the timed region (around every one
    # costs 6 % of a 0.35 ms step in event records alone); RBF steps are 67 ms, every launch carries events there
    sample = 4 if (args.kernel == "Dot" and args.steps >= 16) else 1
    dev.profile_enable(sample)
    t0 = time.perf_counter()
    out = run(args.steps)
    torch.cuda.synchronize()
    comm.barrier()
    elapsed = time.perf_counter() - t0
    prof = dev.profile_fetch()
    dev.profile_enable(False)
    if many:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())
    # --- outside the metric: the synchronous one-batch call (latency an optimiser step sees), with its own kernel times
    nsync = min(args.steps, 100) if args.kernel == "Dot" else min(args.steps, 3)
    for _ in range(3 if args.kernel == "Dot" else 0):
        step()
    torch.cuda.synchronize()
    dev.profile_enable(sample if nsync >= 16 else 1)
    t1 = time.perf_counter()
    for _ in range(nsync):
        sync_out = step()
    torch.cuda.synchronize()
    sync_ms = (time.perf_counter() - t1) / max(nsync, 1) * 1e3
    prof_sync = dev.profile_fetch()
    dev.profile_enable(False)
    if pipelined:                                            # same batch, same model: the two forms agree to the bit
        for a, b in zip(out, sync_out):
            assert np.array_equal(np.asarray(a), np.asarray(b))
    # --- outside the metric: a sustained leg of the timed loop (seconds, not milliseconds: long enough for an external
    # GPU-busy sampler to see the device busy)
    sustained = None
    if args.sustain > 0:
        chunk = max(args.steps, 50)
        n_done, t2 = 0, time.perf_counter()
        while time.perf_counter() - t2 < args.sustain:
            run(chunk)
            n_done += chunk
        torch.cuda.synchronize()
        sustained = {"seconds": time.perf_counter() - t2, "steps": n_done,
                     "structures_per_s": n_done * args.batch / (time.perf_counter() - t2)}
    # --- outside the metric, N > 1: the SAME 64-structure batch dealt over the ranks by structure (strong scaling of one
    # prediction; results come back through one tensor all-reduce), next to the weak-scaled headline
    strong = None
    if many:
        gp.comm = real_comm
        shared = neb_images(args.batch, seed=100)
        for _ in range(5):
            gp.predict_structures(shared, return_std=True, f_tol=1e-8)
        comm.barrier(); torch.cuda.synchronize()
        t3 = time.perf_counter()
        nst = min(args.steps, 100)
        for _ in range(nst):
            gp.predict_structures(shared, return_std=True, f_tol=1e-8)
        torch.cuda.synchronize(); comm.barrier()
        t = torch.tensor([time.perf_counter() - t3], dtype=torch.float64, device="cuda")
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        strong = {"ms_per_step": float(t.item()) / nst * 1e3, "structures_per_s": args.batch * nst / float(t.item()),
                  "what": f"one {args.batch}-structure batch dealt over {world} ranks by structure, results all-reduced"}
        gp.comm = _Serial()
    assert all(np.isfinite(e) for e in out[0])

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = world * args.batch * args.steps / elapsed
        # roofline of the dominant kernel of the timed region (largest summed HIP-event time)
        desc = gp.descriptor.calculate_batch_dev(batch, host_index=True)
        # every seq row (i, j) is one row of atom j's force point; its species is that of centre i
        ele_rows = np.concatenate([np.asarray(nums)[sq[:, 0]] for nums, sq in zip(desc["numbers"], desc["seq"])])
        tX, _, tE, _ = gp.train_x["force"]
        tEe = gp.train_x["energy"][1] if gp.train_x.get("energy") is not None and len(gp.train_x["energy"][-1]) else np.zeros(0, dtype=int)
        ele_atoms = np.concatenate([np.asarray(nums) for nums in desc["numbers"]])      # energy rows of the batch: one per atom
        N = gp.alpha_.shape[0]
        m_rows = sum(1 + 3 * len(s) for s in batch)                    # predicted quantities per step
        n_nbr = int(gp.descriptor.neighbor_index_batch(batch)["w"].shape[0])
        F = (30 * 31 // 2 + 1) * len(set(np.concatenate(desc["numbers"]).tolist()))
        per_step = {k: (ms / args.steps, n / args.steps) for k, (ms, n) in prof.items() if n}
        # Dot at zeta = 2: the step is the native weight-space call (csrc/predict.hip), whose one GEMM is the variance product
        ws_native = args.kernel == "Dot" and gp.native_batch and getattr(gp.kernel, "use_feature_map", False) \
            and gp._weight_space() is not None
        work = {   # algorithmic FP64 operations of one step per kernel kind (SURVEY.md section 8d where it defines them)
            "ff": ("kblock_kernel<FF>: force-force block of K_trans", same_species_pairs(ele_rows, tE) * FLOP_PER_PAIR_FF,
                   "1.10 kflop per same-species row pair"),
            # energy rows of the batch against the training force rows + force rows of the batch against the training
            # energy rows (both through kblock_kernel<EF>), energy rows against energy rows (<EE>)
            "ef": ("kblock_kernel<EF>: energy-force blocks of K_trans",
                   (same_species_pairs(ele_atoms, tE) + same_species_pairs(ele_rows, tEe)) * 285.0, "285 flop per same-species row pair"),
            "ee": ("kblock_kernel<EE>: energy-energy block of K_trans", same_species_pairs(ele_atoms, tEe) * 95.0,
                   "95 flop per same-species row pair"),
            "trmm": ("gemm128_kernel via gpr_trmm_lower: V = L^-1 K_trans^T (predictive variance)", float(N) * N * m_rows,
                     "N^2 m (triangular-solve form of the variance)"),
            "gemm": (("ws_tail_rows_kernel (gpr_ws_tail): mean = Psi* w and the weight-space variance |psi|^2 - |C^-1 psi|^2 from "
                      "FP64 MFMA tiles of Psi* C^-T (C^-1 lower triangular: half of the m x F x F product; the product itself is "
                      "never stored; 32 rows of Psi* resident in LDS per workgroup, C^-1 streamed from L2)",
                      float(m_rows) * F * (F + 1) + 4.0 * m_rows * F,
                      "m F (F + 1) + 4 m F for the m = %d predicted rows and F = %d features" % (m_rows, F))
                     if ws_native else
                     ("gemm128_kernel via gpr_gemm: K_trans^T = sigma^2 Psi_train Psi*^T and K_trans alpha",
                      2.0 * N * m_rows * F + 2.0 * m_rows * N, "2 N m F + 2 m N")),
            "gemv": ("gemv_t_partial_kernel: K_trans alpha", 2.0 * m_rows * N, "2 m N"),
            # the descriptor is two launches from 512 centres on: the radial channels of every pair in a streaming
            # pass, then the centre kernel; each has its own HIP-event bracket.  The ~4 kflop per pair of the whole descriptor
            # (SURVEY 8d counts it as one unit) split by where they are done: 30 dot products of 16 coefficients with T_k(t),
            # the cutoff and the root (1.1 kflop) / Y_lm recurrences, c_tot, R / S contraction and 90 outputs per pair (2.9 kflop)
            "so3": ("so3_fixed_kernel<3, 4, true> (the centre kernel of the descriptor; below 512 centres so3_fixed_kernel<3, 4, false> "
                    "does the radial part too): SO3 power spectrum and derivatives (FP64 vector; no matrix shape)",
                    n_nbr * (2.9e3 if ("so3_radial" in per_step) else 4.0e3),
                    "~2.9 kflop per neighbour pair in the centre kernel (Y_lm recurrences, c_tot, R / S contraction and 90 outputs per "
                    "pair) + 1.1 in the radial pre-pass = the ~4 kflop of the descriptor; the reference's own formulation (80-node "
                    "Bessel quadrature per pair) would be ~21 kflop"),
            "so3_radial": ("so3_radial_pairs_kernel<3, 4>: radial channels A_nl, B_nl of every pair (FP64 vector, table in LDS)", n_nbr * 1.1e3,
                           "~1.1 kflop per neighbour pair (30 dot products of 16 Chebyshev coefficients with T_k(t), cutoff, root)"),
            "features": ("dot2_force_features_mfma_kernel (+ energy-point kernels): degree-2 feature rows and prior variances",
                         float(len(ele_rows)) * 2 * 3 * 2 * 32 * 32,
                         "two passes (native, numpy-eps) x 3 components x 2 x 32 x 32 flop per descriptor row: the symmetric "
                         "rank-2k update M_q = A~^T A'_q on the FP64 matrix cores"),
        }
        dom = max((k for k in per_step if k in work), key=lambda k: per_step[k][0])
        name, flops, unit_note = work[dom]
        avg_ms = per_step[dom][0] / max(per_step[dom][1], 1e-9)
        achieved = flops / max(per_step[dom][1], 1e-9) / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get(dom, {}).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        per_sync = {k: (ms / nsync, n / nsync) for k, (ms, n) in prof_sync.items() if n}
        roofline = {"bound": "fp64-vector" if dom in ("so3", "so3_radial", "ee") else "mfma", "achieved": achieved, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": achieved / FP64_MFMA_PEAK_TFLOPS, "traffic": traffic, "kernel": name,
                    "avg_launch_ms": avg_ms, "launches_per_step": per_step[dom][1], "event_sampling": sample, "flop_per_step": flops,
                    # the timed region keeps `batches_in_flight` device chains running side by side: a launch's duration there
                    # includes what it waits for the kernels of the other chains (rocprofv3 of this command shows the same).
                    # `solo`: the same kernel in the synchronous one-batch calls measured right after the timed region
                    "batches_in_flight": depth if pipelined else 1,
                    "solo": ({"avg_launch_ms": per_sync[dom][0] / max(per_sync[dom][1], 1e-9),
                              "achieved": flops / max(per_step[dom][1], 1e-9) / (per_sync[dom][0] / max(per_sync[dom][1], 1e-9) * 1e-3) / 1e12,
                              "frac": flops / max(per_step[dom][1], 1e-9) / (per_sync[dom][0] / max(per_sync[dom][1], 1e-9) * 1e-3) / 1e12 / FP64_MFMA_PEAK_TFLOPS}
                             if dom in per_sync and per_sync[dom][0] > 0 else None),
                    "work_unit": unit_note,
                    "kernel_ms_per_step": {k: round(v[0], 4) for k, v in sorted(per_step.items(), key=lambda kv: -kv[1][0])},
                    # the same figures for every profiled kernel kind of the step (the top two are often within a few
                    # per cent of each other, so which one is "dominant" can change from run to run)
                    "per_kernel": {k: {"ms_per_step": round(per_step[k][0], 4), "flop_per_step": work[k][1],
                                       "achieved": work[k][1] / (per_step[k][0] * 1e-3) / 1e12 if per_step[k][0] > 0 else 0.0,
                                       "frac": work[k][1] / (per_step[k][0] * 1e-3) / 1e12 / FP64_MFMA_PEAK_TFLOPS if per_step[k][0] > 0 else 0.0}
                                   for k in per_step if k in work},
                    # the same kernels timed alone on the device (synchronous one-batch calls after the timed region: no
                    # second batch shares the chip with them); in the timed region up to `depth` chains run side by side
                    # and a launch's duration includes what it waits for the others
                    "per_kernel_solo": {k: {"ms_per_step": round(per_sync[k][0], 4),
                                            "frac": work[k][1] / (per_sync[k][0] * 1e-3) / 1e12 / FP64_MFMA_PEAK_TFLOPS if per_sync[k][0] > 0 else 0.0}
                                        for k in per_sync if k in work},
                    "whole_step": {"flop_per_step": sum(work[k][1] for k in per_step if k in work),
                                   "frac": sum(work[k][1] for k in per_step if k in work) / (ms_per_step * 1e-3) / 1e12 / FP64_MFMA_PEAK_TFLOPS}}
        # the same step through the row-pair route (what every other kernel / zeta takes): the Dot kernel object with its
        # feature map switched off builds K_trans with kblock_kernel<FF> -- the FP64-MFMA kernel of SURVEY section 8d --
        # measured live here so that its roofline point sits next to the headline's
        if world == 1 and args.kernel == "Dot" and not args.no_pair_route:
            gp.kernel.use_feature_map = False
            gp._ws_cache = None
            gp.predict_structures(batch, return_std=True, f_tol=1e-8)
            dev.profile_enable(True)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(2):
                gp.predict_structures(batch, return_std=True, f_tol=1e-8)
            torch.cuda.synchronize()
            pr_ms = (time.perf_counter() - t1) / 2 * 1e3
            pprof = dev.profile_fetch()
            dev.profile_enable(False)
            gp.kernel.use_feature_map = True
            gp._ws_cache = None
            ff_ms, ff_n = pprof.get("ff", (0.0, 0))
            if ff_n:
                ff_flop = same_species_pairs(ele_rows, tE) * FLOP_PER_PAIR_FF
                ach = ff_flop / (ff_ms / ff_n * 1e-3) / 1e12
                roofline["pair_route"] = {"kernel": "kblock_kernel<FF>: force-force block of K_trans, Dot zeta=2 pair form",
                                          "bound": "mfma", "avg_launch_ms": ff_ms / ff_n, "achieved": ach,
                                          "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / FP64_MFMA_PEAK_TFLOPS,
                                          "ms_per_step": pr_ms, "structures_per_s": args.batch / (pr_ms * 1e-3),
                                          "work_unit": "1.10 kflop per same-species row pair (SURVEY.md section 8d)"}
        shape = {500: "cfg2", 2000: "cfg3", 8000: "cfg4"}.get(args.n_train, "cfg3-shaped")
        line = {"metric": "GPR predict E+F+sigma structures/sec at N_train=%d" % args.n_train, "value": value,
                "unit": "structures/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                # untimed steps actually run before the timed region: the caller's W plus the priming steps above
                "untimed_steps_before_timing": args.warmup + n_prime,
                # the same batch through the synchronous call (one batch at a time: what rounds 1-2 reported as `value` and
                # what an optimiser step sees); `value` is the throughput form with `batches_in_flight` batches queued
                "value_sync": world * args.batch / (sync_ms * 1e-3), "sync_ms_per_step": sync_ms,
                # N > 1: `value` is N replicas by construction; the informative figures are ONE batch over N ranks and the
                # N-rank fit
                "strong_predict": strong, "fit_ms_all_ranks": {"weight_space": fit_ms, "dense": fit_dense_ms},
                "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                "dtype": "f64", "data": "synthetic",
                "config": {"workload": f"{shape}: N_train={args.n_train} force environments + {len(gp.train_x['energy'][-1])} "
                                       f"energy points (N={N}), {args.kernel} kernel zeta=2, SO3(3,4,5.0), predict "
                                       f"E/F/sigma batch of {args.batch} 32-atom structures per GPU",
                           "kernel": args.kernel, "n_train": args.n_train, "batch_per_gpu": args.batch, "N": int(N),
                           "train_force_rows": int(len(tX)),
                           # fit(opt=False) on `world` ranks: warm = mean of 3 calls after 2 untimed ones; cold = the
                           # first call of that route in the process
                           "fit_ms": fit_ms, "fit_cold_ms": fit_cold, "fit_dense_ms": fit_dense_ms, "fit_dense_cold_ms": fit_dense_cold,
                           "batches_in_flight": depth if pipelined else 1,
                           "sync_ms_per_step": sync_ms, "sync_structures_per_s": args.batch / (sync_ms * 1e-3),
                           "sustained": sustained, "strong_predict": strong,
                           "ranks": world, "backend": (torch.distributed.get_backend() if many else None),
                           "parallelism": f"replicated model, structure-parallel predict x{world}; fit K sharded x{world}"},
                "roofline": roofline, "cpu_baseline": cpu}
        print(json.dumps(line))
    if many:
        torch.distributed.destroy_process_group()


class _Serial:
    """stand-in communicator: rank 0 of 1 (each rank predicts its own whole batch in the weak-scaling run)"""
    rank, size, active = 0, 1, False

    def barrier(self):
        pass

    def bcast(self, obj, root=0):
        return obj


if __name__ == "__main__":
    main()
