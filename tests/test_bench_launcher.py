"""CPU: `python bench.py --gpus N` started plainly (no torchrun environment) launches the N ranks as a child job before
touching any GPU, and relays rank 0's JSON line and the job's exit code.  --dry-run keeps the ranks off the GPU: they
rendezvous on 127.0.0.1 (gloo here), all-reduce who is there, and report."""
import json
import os
import subprocess
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _run(extra, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, cwd=ROOT, env=env, capture_output=True,
                          text=True, timeout=600)


def test_gpus_2_starts_two_ranks_and_relays_one_json_line():
    res = _run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-run"], {"GPR_DIST_BACKEND": "gloo"})
    assert res.returncode == 0, res.stdout + res.stderr
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["config"]["ranks"] == 2 and line["config"]["ranks_seen"] == [1, 1]
    assert line["config"]["backend"] == "gloo" and line["steps"] == 3 and line["warmup"] == 1


def test_a_failing_rank_fails_the_parent():
    res = _run(["--gpus", "2", "--dry-run", "--kernel", "nonsense"])
    assert res.returncode != 0
    assert not [l for l in res.stdout.splitlines() if l.startswith("{")]


def test_one_rank_needs_no_launcher():
    res = _run(["--gpus", "1", "--dry-run"])
    assert res.returncode == 0, res.stdout + res.stderr
    line = json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["config"]["ranks_seen"] == [1]
