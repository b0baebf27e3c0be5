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


@pytest.fixture(scope="session")
def port_natives():
    """The plain-C oracle restatement of the 13 natives (built on demand; gcc only)."""
    import subprocess
    from oracle.natives import NativeSet
    if not NativeSet.available("port"):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "port"])
    return NativeSet("port")
