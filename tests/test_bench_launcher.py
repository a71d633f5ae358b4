"""bench.py --gpus N started without a launcher: the parent's watchdog.  Runs on the CPU: the rank that
is told to die does so before it imports anything, the parent must notice, stop the other rank and exit with that
rank's code within seconds -- not sit in communicate() until a collective times out."""
import os
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_parent_exits_with_the_failed_ranks_code():
    env = dict(os.environ, CASYNC_BENCH_FAIL_RANK="1")
    env.pop("RANK", None)
    env.pop("WORLD_SIZE", None)
    t0 = time.monotonic()
    res = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                          "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=120)
    took = time.monotonic() - t0
    assert res.returncode == 3, (res.returncode, res.stderr[-500:])
    assert "rank 1 exited with code 3" in res.stderr
    assert took < 30, took


def test_world8_parent_exits_with_the_failed_ranks_code():
    """The N = 8 launch of `bench.py --gpus 8` (the driver's scaling run; never started on the one-GPU box, whose process
    guard allows six GPU processes): eight children, rank 5 dies before importing anything, the parent reports THAT rank and
    its code within seconds and stops the seven others (on this CPU box they would otherwise fail later, for want of a GPU)."""
    env = dict(os.environ, CASYNC_BENCH_FAIL_RANK="5")
    env.pop("RANK", None)
    env.pop("WORLD_SIZE", None)
    t0 = time.monotonic()
    res = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "0",
                          "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=120)
    took = time.monotonic() - t0
    assert res.returncode == 3, (res.returncode, res.stderr[-500:])
    assert "rank 5 exited with code 3" in res.stderr
    assert took < 30, took
