"""GPU: the multi-rank path in real processes (tests/rank_worker.py under torch.distributed.run).

  * test_rccl_ranks_one_per_gpu -- the first contact of this code with RCCL: as many ranks as the box has GPUs (at most
    four), backend "nccl", one rank per device.  The job is started by tests/conftest.py at the very beginning of the
    session, before this pytest process has made a single GPU call, and the test reads its verdict.  Skipped on boxes
    with fewer than two GPUs (every box this repository has been developed on).
  * test_two_ranks_share_the_one_gpu_over_gloo -- the same worker, same checks, two ranks on one device with gloo as the
    transport: what can be rehearsed on a one-GPU box is everything except the transport.
  * test_one_rank_through_rccl -- ... and the transport itself with a world of one: RCCL initialises, the collectives run on
    device tensors through backend "nccl".""" 
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def launch(nranks, backend, timeout=900, force=False):
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = env.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["GPR_DIST_BACKEND"] = backend
    if force:
        env["GPR_DIST_FORCE"] = "1"        # a world of one still goes through the backend
    env["GPR_DIST_LOOKAHEAD"] = "1"        # the two-lane schedule RCCL gets by default, also in the gloo rehearsal
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nranks}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "rank_worker.py")]
    res = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    return res.returncode, (json.loads(lines[-1]) if lines else None), res.stdout[-3000:] + res.stderr[-6000:]


def test_rccl_ranks_one_per_gpu(request):
    verdict = getattr(request.config, "_gpr_rccl_job", None)
    if verdict is None:
        pytest.skip("fewer than two GPUs on this box: RCCL has nobody to talk to (the gloo rehearsal below runs instead)")
    rc, line, tail = verdict
    assert rc == 0 and line is not None, tail
    assert line["rank_worker"] == "ok" and line["backend"] == "nccl" and line["ranks"] >= 2, line


def test_two_ranks_share_the_one_gpu_over_gloo():
    rc, line, tail = launch(2, "gloo")
    assert rc == 0 and line is not None, tail
    assert line["rank_worker"] == "ok" and line["backend"] == "gloo" and line["ranks"] == 2, line


def test_one_rank_through_rccl():
    """What a one-GPU box can check of the RCCL path: the library initialises here, and every tensor collective of the worker
    (equal and ragged all-gathers, in-place slab exchange, all-reduce, broadcast, object broadcast) runs through backend "nccl"
    on device tensors -- with a world of one (GPR_DIST_FORCE=1 removes the single-rank short cuts)."""
    rc, line, tail = launch(1, "nccl", force=True)
    assert rc == 0 and line is not None, tail
    assert line["rank_worker"] == "ok" and line["backend"] == "nccl" and line["ranks"] == 1, line
