"""aruco_slam_amd — MI355X (gfx950) implementation of the ArUco EKF-SLAM hot path of gitAugust/Aruco_Slam.

`capi`   ctypes binding of the C-ABI shared library (hand-written HIP kernels, no CPU fallback); `capi.Context` mirrors the
         reference's `ArucoSlam` surface (add_encoder / add_image / getters) plus the staged stream API
`synth`  deterministic synthetic scenes (inputs for tests and bench.py)
`dist`   one-stream-per-GPU sharding and the RCCL gather of the landmark map
(the C++ class surface itself is include/aruco_slam/aruco_slam.h on top of include/aruco_slam_hip.h)
"""
from . import capi  # noqa: F401

__all__ = ["capi"]
