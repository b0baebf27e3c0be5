"""CPU, world size 2 over gloo: the multi-GPU fit schedule of gpr_calculator_amd/distributed.py -- point shards,
lower-trapezoid row slabs + all-gather + mirror, and the block-cyclic right-looking Cholesky (owner factors, broadcast,
panel solve, panel all-gather, trailing update, factor gather) -- driven through the real collectives with the local
arithmetic in numpy / scipy (dist_helpers.HostOps), checked against scipy.linalg.cholesky and the oracle's K."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from gpr_calculator_amd.distributed import (RowBlockCyclic, assemble_sharded, potrf_distributed, shard_points)
from gpr_calculator_amd.parallel import Comm


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def spd(n, seed=0):
    rng = np.random.default_rng(seed)
    B = rng.standard_normal((n, n // 2 + 3))
    return B @ B.T + np.eye(n) * (0.05 * n + 1.0)


def synthetic_train(seed=3):
    """random descriptor rows packed as a training dictionary: 5 energy points, 11 force points, two species"""
    rng = np.random.default_rng(seed)
    ce, cf = [4, 6, 5, 4, 7], [3, 5, 4, 6, 3, 4, 5, 3, 6, 4, 5]
    nE, nFr = sum(ce), sum(cf)
    E = (rng.random((nE, 12)), rng.choice([8, 46], nE).astype(np.int64), ce)
    F = (rng.random((nFr, 12)), rng.standard_normal((nFr, 12, 3)), rng.choice([8, 46], nFr).astype(np.int64), cf)
    return {"energy": E, "force": F}


def _worker(rank, world, port, q):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from dist_helpers import HostOps
        from oracle import gp_numpy as og
        from oracle.natives import NativeSet
        comm, out = Comm(), {}
        for (n, nb) in ((700, 256), (700, 128), (513, 256), (200, 256)):
            A = spd(n, seed=n)
            L, _, info = potrf_distributed(torch.from_numpy(A.copy()), comm, HostOps(), nb=nb)
            out[f"chol_{n}_{nb}"] = L.numpy()
            out[f"info_{n}_{nb}"] = int(info.item())
        # not positive definite, and the failing diagonal block (rows 256..511) belongs to rank 1: every rank must come
        # back with the same non-zero info (and none may be left waiting in a collective)
        A = spd(700, seed=1)
        A[300, 300] = -5.0
        _, _, info = potrf_distributed(torch.from_numpy(A.copy()), comm, HostOps(), nb=256)
        out["info_indefinite"] = int(info.item())
        kern = og.OracleRBF(para=(2.5, 1.3), zeta=2.0, natives=NativeSet("port"))
        tx = synthetic_train()

        def block_fn(d1, d2):
            K, dK = None, None
            full1 = {"energy": d1["energy"]} if "energy" in d1 else {"force": d1["force"]}
            return (torch.from_numpy(np.ascontiguousarray(kern.k_total(full1, d2, 1e-12))),)
        out["K"] = assemble_sharded(tx, comm, block_fn, HostOps())[0].numpy()
        q.put((rank, out))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as exc:                       # surface failures instead of hanging the parent
        q.put((rank, {"error": repr(exc)}))
        raise


def test_point_shards_and_row_distribution_are_consistent():
    ce, cf = [32] * 5, [28, 30, 27, 28, 29, 28, 31, 28] * 6
    for size in (1, 2, 3, 4, 8):
        sh = shard_points(ce, cf, size)
        assert sh[0][0] == 0 and sh[0][2] == 0 and sh[-1][1] == len(ce) and sh[-1][3] == len(cf)
        for a, b in zip(sh[:-1], sh[1:]):
            assert a[1] == b[0] and a[3] == b[2]                 # contiguous, disjoint, complete
        if size > 1:                                             # trapezoid areas within ~one point of each other
            area = [sum(cf[p] * (sum(ce) + sum(cf[:p + 1])) for p in range(s[2], s[3])) for s in sh]
            assert max(area) - min(area) <= 2 * max(cf) * (sum(ce) + sum(cf))
    for (n, size, nb) in ((700, 2, 256), (6063, 4, 256), (1000, 3, 128), (255, 2, 256)):
        d = RowBlockCyclic(n, size, nb)
        rows = np.concatenate([d.global_rows(r) for r in range(size)])
        assert sorted(rows.tolist()) == list(range(n))
        assert sum(d.local_nrows(r) for r in range(size)) == n
        for k in (-1, 0, d.nblk // 2, d.nblk - 2):
            if k >= d.nblk - 1 and k >= 0:
                continue
            # rank-major concatenation of the rows after block k, put back into global order
            cat = np.concatenate([d.global_rows(r)[d.rows_after(r, k)[0]:] for r in range(size)])
            assert np.array_equal(cat[d.perm_after(k)], np.arange((k + 1) * nb, n))


def test_slab_assembly_and_block_cyclic_cholesky_over_gloo(port_natives):
    from scipy.linalg import cholesky
    from oracle import gp_numpy as og
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
    for r in range(world):
        assert "error" not in res[r], res[r]
    for (n, nb) in ((700, 256), (700, 128), (513, 256), (200, 256)):
        ref = cholesky(spd(n, seed=n), lower=True)
        for r in range(world):
            got = res[r][f"chol_{n}_{nb}"]
            assert np.abs(got - ref).max() <= 1e-12 * np.abs(ref).max(), (n, nb, r)
            assert np.all(got[np.triu_indices(n, 1)] == 0.0)
            assert res[r][f"info_{n}_{nb}"] == 0
    assert res[0]["info_indefinite"] == res[1]["info_indefinite"] == 257       # first column of the failing block, 1-based
    kern = og.OracleRBF(para=(2.5, 1.3), zeta=2.0, natives=port_natives)
    K = kern.k_total(synthetic_train())
    for r in range(world):
        assert np.abs(res[r]["K"] - K).max() <= 1e-12 * np.abs(K).max()
        assert np.array_equal(res[r]["K"], res[r]["K"].T) and np.array_equal(res[r]["K"], res[0]["K"])
