"""bench.py's launcher contract on a box that cannot satisfy --gpus N: a parent that never touches a GPU, a clear message and a
non-zero exit code (here: no GPU at all; on the 1-GPU box the same path refuses --gpus 2)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_gpus_n_without_enough_gpus_exits_non_zero_with_a_message():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "64", "--steps", "1"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 2, (r.returncode, r.stderr[-500:])
    assert "needs 64 GPUs" in r.stderr and r.stdout.strip() == ""
