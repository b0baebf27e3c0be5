import os
import sys
import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
HERE = os.path.dirname(os.path.abspath(__file__))
if HERE not in sys.path:
    sys.path.insert(0, HERE)                    # tests/gp_golden.py (fixture helpers)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """The RCCL job of tests/test_gpu_ranks.py runs HERE, before this process has made any GPU call (child processes
    that initialise their own devices must not descend from a parent that already holds one): one rank per GPU, at most
    four.  Only when GPU tests are selected and the box has at least two devices -- counting devices does not initialise
    the runtime.  The verdict is parked on the config for the test to read."""
    config = session.config
    markexpr = getattr(config.option, "markexpr", "") or ""
    if "gpu" not in markexpr or "not gpu" in markexpr:
        return
    try:
        import torch
        ndev = torch.cuda.device_count()
    except Exception:
        return
    if ndev < 2:
        return
    from test_gpu_ranks import launch
    config._gpr_rccl_job = launch(min(ndev, 4), "nccl")


@pytest.fixture(scope="session")
def port_natives():
    """The plain-C oracle restatement of the 13 natives (built on demand; gcc only)."""
    import subprocess
    from oracle.natives import NativeSet
    if not NativeSet.available("port"):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "port"])
    return NativeSet("port")
