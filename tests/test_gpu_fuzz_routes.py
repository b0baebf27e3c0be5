"""Randomised cross-check of the two prediction routes of the Dot (zeta = 2) model -- the session call (feature rows and
weight-space tail) against the kernel-object route (pair kernel, dense factor) -- over models with one to three species and
batches of 1 to 100 mixed structures with odd atom counts, strained cells and fixed atoms (tools/fuzz_session.py)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def test_session_route_agrees_with_kernel_object_route_on_random_cases():
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_session.py"), "36", "11"], cwd=ROOT,
                         stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert res.returncode == 0 and "ALL OK" in res.stdout, res.stdout[-2000:]
