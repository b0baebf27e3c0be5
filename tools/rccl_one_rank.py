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

# ... and bench.py itself as the driver starts it for N > 1 (one rank here): predict mode (weak-scaled headline + the strong-scaled
# batch through one all-reduce) and fit mode (sharded assembly, block-cyclic Cholesky from 1024 rows on)
import json, socket, subprocess
ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
for extra, envx in ((["--steps", "20", "--warmup", "3", "--no-pair-route", "--no-cpu-baseline"], {}),
                    (["--mode", "fit", "--kernel", "RBF", "--steps", "2", "--warmup", "1"], {"GPR_DIST_CHOL_MIN_N": "1024"})):
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update({"GPR_DIST_FORCE": "1", "GPR_DIST_BACKEND": "nccl", "HSA_ENABLE_IPC_MODE_LEGACY": env.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")}, **envx)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", "1"] + extra
    res = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    if res.returncode != 0 or not lines:
        print("bench.py", extra, "FAILED rc", res.returncode); print(res.stdout[-2000:]); print(res.stderr[-6000:])
        sys.exit(1)
    d = json.loads(lines[-1])
    print("bench.py", " ".join(extra), "->", d["metric"], round(d["value"], 3), d["unit"], "| ranks", d["config"].get("ranks"), "backend", d["config"].get("backend"),
          "| strong_predict", (d["config"].get("strong_predict") or {}).get("ms_per_step"))
