"""pytest configuration.

`-m gpu` tests are the parity tests proper: they drive the hipcc/gfx950 build of libaruco_slam_hip.so through
its C-ABI on a real MI355X and compare with the CPU oracle (oracle/).  Everything else runs without a GPU:
oracle known-answer tests, golden vectors, host logic, C-ABI symbol checks and — through the CPU *emulation*
build of the same kernel sources (tests/hipemu, test infrastructure) — small-size parity of the kernel logic.
"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

REAL_LIB = os.path.join(ROOT, "aruco_slam_amd", "libaruco_slam_hip.so")
EMU_LIB = os.path.join(ROOT, "tests", "hipemu", "_build", "libaruco_slam_emu.so")


def have_gpu():
    return os.path.exists("/dev/kfd") and os.path.exists(REAL_LIB)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run through gpurun / by the driver at round end)")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    if "ARUCO_SLAM_LIB" not in os.environ:
        if have_gpu():
            os.environ["ARUCO_SLAM_LIB"] = REAL_LIB
        else:
            subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "hipemu")])
            os.environ["ARUCO_SLAM_LIB"] = EMU_LIB


def pytest_collection_modifyitems(config, items):
    if have_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container (gpu tests run on the MI355X box)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def on_emulation():
    return os.environ.get("ARUCO_SLAM_LIB") == EMU_LIB
