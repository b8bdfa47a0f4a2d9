"""N > 1 path on CPU: two ranks (gloo), one independent stream each, landmark-map all-gather."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_rank_map_gather_gloo():
    env = dict(os.environ)
    env.setdefault("ARUCO_SLAM_LIB", os.path.join(ROOT, "tests", "hipemu", "_build", "libaruco_slam_emu.so"))
    env["OMP_NUM_THREADS"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29611", os.path.join(ROOT, "tests", "_dist_worker.py")]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "rank 0 ok" in r.stdout and "rank 1 ok" in r.stdout
