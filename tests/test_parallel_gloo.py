"""CPU, world_size 2 over gloo: the communicator that stands where mpi4py's COMM_WORLD stands in the
reference, and the host-side partitioning used by the multi-GPU paths."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from gpr_calculator_amd.parallel import Comm, balanced_ranges, shard_range


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, q):
    try:
        _worker_body(rank, world, port, q)
    except Exception as exc:                       # surface failures instead of hanging the parent
        q.put((rank, {"error": repr(exc)}))
        raise


def _worker_body(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    comm = Comm()
    out = {}
    out["rank_size"] = (comm.rank, comm.size, comm.Get_rank(), comm.Get_size())
    out["bcast"] = comm.bcast({"params": [1.5, rank]} if rank == 0 else None, root=0)
    out["allreduce"] = comm.allreduce(np.array([1.0, 2.0]) * (rank + 1))
    # all-gather of ragged row slabs (K slabs of a point shard)
    counts = [3, 5]
    local = torch.full((counts[rank], 4), float(rank + 1), dtype=torch.float64)
    out["gather"] = comm.all_gather_rows(local, counts).numpy()
    # equal slabs take the single all_gather_into_tensor; a caller-provided landing zone is filled in place
    land = torch.zeros((8, 4), dtype=torch.float64)
    got = comm.all_gather_rows(torch.full((4, 4), float(rank + 1), dtype=torch.float64), [4, 4], out=land)
    out["gather_equal"] = (got.data_ptr() == land.data_ptr(), land.numpy().copy())
    # slabs computed in place in the full matrix, then exchanged in place (covariance assembly)
    full = torch.zeros((6, 3), dtype=torch.float64)
    row0, cnt, owners = [0, 2, 3, 5], [2, 1, 2, 1], [0, 1, 0, 1]
    for o, c, r in zip(row0, cnt, owners):
        if r == rank:
            full[o:o + c] = 10.0 * (r + 1) + torch.arange(o, o + c, dtype=torch.float64)[:, None]
    out["slabs"] = comm.gather_slabs_inplace(full, row0, cnt, owners).numpy().copy()
    # rounds of equal, adjacent slabs take ONE all_gather_into_tensor each, straight into their rows
    full = torch.zeros((10, 3), dtype=torch.float64)
    row0, cnt, owners = [0, 2, 4, 7], [2, 2, 3, 3], [0, 1, 0, 1]
    for o, c, r in zip(row0, cnt, owners):
        if r == rank:
            full[o:o + c] = 10.0 * (r + 1) + torch.arange(o, o + c, dtype=torch.float64)[:, None]
    out["slabs_equal"] = comm.gather_slabs_inplace(full, row0, cnt, owners).numpy().copy()
    out["max"] = comm.all_reduce_max_(torch.tensor([float(rank), 5.0 - rank], dtype=torch.float64)).numpy().copy()
    # all-reduce of a sharded matrix: every element owned by exactly one rank -> exact sum
    full = torch.arange(24, dtype=torch.float64).reshape(4, 6) * 0.1
    mine = torch.zeros_like(full)
    owner = (torch.arange(4)[:, None] + torch.arange(6)[None, :]) % world
    mine[owner == rank] = full[owner == rank]
    out["reduce_exact"] = bool(torch.equal(comm.all_reduce_sum_(mine), full))
    # structure-parallel split used by GP.predict_structures
    nb = 7
    mine_ids = list(range(rank, nb, world))
    box = [None] * world
    dist.all_gather_object(box, {k: k * k for k in mine_ids})
    merged = {}
    for part in box:
        merged.update(part)
    out["structures"] = merged
    comm.barrier()
    q.put((rank, out))
    dist.destroy_process_group()


def test_comm_world2_gloo():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in range(world):
        o = res[r]
        assert "error" not in o, o
        assert o["rank_size"] == (r, 2, r, 2)
        assert o["bcast"] == {"params": [1.5, 0]}
        assert np.allclose(o["allreduce"], [3.0, 6.0])
        assert o["gather"].shape == (8, 4) and (o["gather"][:3] == 1).all() and (o["gather"][3:] == 2).all()
        assert o["gather_equal"][0] and (o["gather_equal"][1][:4] == 1).all() and (o["gather_equal"][1][4:] == 2).all()
        want = np.array([10.0, 11.0, 22.0, 13.0, 14.0, 25.0])[:, None] * np.ones((1, 3))
        assert np.array_equal(o["slabs"], want)
        want = np.array([10.0, 11.0, 22.0, 23.0, 14.0, 15.0, 16.0, 27.0, 28.0, 29.0])[:, None] * np.ones((1, 3))
        assert np.array_equal(o["slabs_equal"], want)
        assert np.array_equal(o["max"], [1.0, 5.0])
        assert o["reduce_exact"]
        assert o["structures"] == {k: k * k for k in range(7)}


def test_serial_comm_is_a_noop():
    c = Comm()
    assert (c.rank, c.size) == (0, 1)
    assert c.bcast([1, 2]) == [1, 2] and c.allreduce(3.5) == 3.5
    t = torch.ones(2, 2)
    assert c.all_gather_rows(t, [2]) is t and c.all_reduce_sum_(t) is t
    c.barrier()


def test_partitions():
    # the reference's ceil split (RBF_mb.py:471-475)
    assert [shard_range(10, r, 4) for r in range(4)] == [(0, 3), (3, 6), (6, 9), (9, 10)]
    assert [shard_range(2, r, 4) for r in range(4)] == [(0, 1), (1, 2), (2, 2), (2, 2)]
    rng = np.random.default_rng(0)
    w = rng.integers(1, 60, 200)
    parts = balanced_ranges(w, 8)
    assert parts[0][0] == 0 and parts[-1][1] == 200 and all(a[1] == b[0] for a, b in zip(parts, parts[1:]))
    loads = [w[a:b].sum() for a, b in parts]
    assert max(loads) - min(loads) <= 2 * w.max()
