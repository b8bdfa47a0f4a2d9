"""The device promises no order among the workgroups of a launch.  The CPU emulation normally starts them in index order; with
HIPEMU_ORDER=reverse / shuffle (tests/hipemu/hip/hip_runtime.h) it starts them backwards / in a fixed random permutation.  A kernel whose
workgroups hand data to each other through global memory inside ONE launch would give different results (the round-2 race in the old
k_ekf_win_scan was of that kind): the window cases with the most hand-overs and the detection pipeline must not care."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("order", ["reverse", "shuffle"])
def test_results_do_not_depend_on_the_workgroup_order(order, on_emulation):
    if not on_emulation:
        pytest.skip("a property of the CPU emulation build")
    env = dict(os.environ, HIPEMU_ORDER=order)
    cmd = [sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider",
           os.path.join(ROOT, "tests", "test_ekf_window.py"), "-k", "two_groups or window_to_window or sliding_set or pieces",
           os.path.join(ROOT, "tests", "test_pipeline_emu.py")]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1500, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
