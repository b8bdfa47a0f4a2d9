"""The C-ABI shared library (hipcc / gfx950 build) loads without a GPU and exports every symbol declared in
include/aruco_slam_hip.h; without a usable device the product path fails loudly (no CPU fallback)."""
import ctypes
import os
import re

import pytest

from aruco_slam_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REAL = os.path.join(ROOT, "aruco_slam_amd", "libaruco_slam_hip.so")
HEADER = os.path.join(ROOT, "include", "aruco_slam_hip.h")


def declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(aslam_[a-z_0-9]+)\s*\(", src)))


def test_header_and_binding_agree():
    assert set(declared_symbols()) == set(capi.EXPORTED_SYMBOLS)


@pytest.mark.skipif(not os.path.exists(REAL), reason="library not built (run __graft_entry__.build())")
def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(REAL)
    for name in declared_symbols():
        assert hasattr(lib, name), f"{name} missing from libaruco_slam_hip.so"


@pytest.mark.skipif(os.path.exists("/dev/kfd") or not os.path.exists(REAL), reason="only meaningful without a GPU")
def test_create_fails_loudly_without_a_device():
    lib = ctypes.CDLL(REAL)
    lib.aslam_default_init.argtypes = [ctypes.POINTER(capi.AslamInit)]
    lib.aslam_create.argtypes = [ctypes.POINTER(capi.AslamInit), ctypes.POINTER(ctypes.c_void_p)]
    init = capi.AslamInit()
    lib.aslam_default_init(ctypes.byref(init))
    h = ctypes.c_void_p()
    assert lib.aslam_create(ctypes.byref(init), ctypes.byref(h)) == -2        # ASLAM_E_NO_DEVICE
    assert not h.value


def test_default_init_matches_reference_parameters():       # parameters.yaml:5-17, aruco_slam.h:58
    i = capi.default_init()
    assert (i.Q_k, i.R_x, i.R_y, i.R_theta) == (0.01, 100.0, 100.0, 10.0)
    assert (i.kl, i.kr, i.b) == (0.05, 0.05, 0.09)
    assert i.marker_length == 0.27 and i.markers_dictionary == 16 and i.useful_distance_threshold == 3.0
