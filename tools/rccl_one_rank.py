"""tests/rank_worker.py as ONE rank with backend "nccl": every collective of the fit and the prediction goes through RCCL
(trivially: a world of one), on a box with a single GPU -- what can be checked of the RCCL path without a second device is that
the library initialises here and that the device-tensor code paths the gloo rehearsal never takes run.
usage: python tools/rccl_one_rank.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from test_gpu_ranks import launch
rc, line, tail = launch(1, "nccl", timeout=500, force=True)
print("rc", rc)
print(line)
if rc != 0 or line is None:
    print(tail)
