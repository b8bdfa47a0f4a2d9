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


def test_bench_launcher_starts_n_ranks():
    """`python bench.py --gpus 2` with no torchrun environment must start the two ranks itself (BASELINE config 4 is driven
    that way); rehearsed on the CPU emulation build with gloo through the hidden --emu-check mode of the very same launcher"""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["ARUCO_SLAM_LIB"] = os.path.join(ROOT, "tests", "hipemu", "_build", "libaruco_slam_emu.so")
    env["OMP_NUM_THREADS"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--emu-check"], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    js = json.loads(lines[0])
    assert js["n_gpus"] == 2 and len(js["landmarks_per_rank"]) == 2 and js["value"] > 0
