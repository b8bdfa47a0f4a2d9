"""aruco_slam_amd — MI355X (gfx950) implementation of the ArUco EKF-SLAM hot path of gitAugust/Aruco_Slam.

`capi`   ctypes binding of the C-ABI shared library (hand-written HIP kernels, no CPU fallback)
`slam`   `ArucoSlam`: host-side mirror of the reference class surface (aruco_slam.h:101-193)
`synth`  deterministic synthetic scenes (inputs for tests and bench.py)
`dist`   one-stream-per-GPU sharding and the RCCL gather of the landmark map
"""
from . import capi  # noqa: F401

__all__ = ["capi"]
