"""ctypes binding of libaruco_slam_hip.so (include/aruco_slam_hip.h).

The product path is the hipcc/gfx950 build next to this file.  There is no CPU fallback: if the library
is missing, or no HIP device is usable, loading / `Context()` raises.  (The parity tests may point
ARUCO_SLAM_LIB at the CPU *emulation* build of the same sources under tests/hipemu — test
infrastructure used only by `-m "not gpu"` tests to exercise kernel logic in the GPU-less container.)
"""
import ctypes as C
import importlib.util
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIB = os.path.join(_HERE, "libaruco_slam_hip.so")

ASLAM_OK = 0
E_NAMES = {-1: "INVALID", -2: "NO_DEVICE", -3: "HIP", -4: "CAPACITY", -5: "STATE"}
MAP_RECORD_BYTES = 104
MARKER_MAX = 128
CAND_MAX = 2048


class AslamInit(C.Structure):
    _fields_ = [
        ("Q_k", C.c_double), ("R_x", C.c_double), ("R_y", C.c_double), ("R_theta", C.c_double),
        ("kl", C.c_double), ("kr", C.c_double), ("b", C.c_double),
        ("marker_length", C.c_double),
        ("markers_dictionary", C.c_int),
        ("useful_distance_threshold", C.c_float),
        ("r2c_t", C.c_double * 3), ("r2c_q", C.c_double * 4),
        ("device_id", C.c_int),
        ("max_landmarks", C.c_int),
        ("max_rows", C.c_int), ("max_cols", C.c_int),
        ("max_batch", C.c_int),
        ("persistent_waves", C.c_int),
        ("max_updates_per_frame", C.c_int),
        ("cap_starts_per_frame", C.c_uint),
        ("cap_contours_per_frame", C.c_uint),
        ("cap_points_per_frame", C.c_uint),
        ("ekf_reserved_cus_per_xcd", C.c_int),
    ]


class DetectorParams(C.Structure):
    """mirror of aslam_detector_params (cv::aruco::DetectorParameters of OpenCV 3.2.0)"""
    _fields_ = [
        ("adaptiveThreshWinSizeMin", C.c_int), ("adaptiveThreshWinSizeMax", C.c_int), ("adaptiveThreshWinSizeStep", C.c_int),
        ("adaptiveThreshConstant", C.c_double),
        ("minMarkerPerimeterRate", C.c_double), ("maxMarkerPerimeterRate", C.c_double),
        ("polygonalApproxAccuracyRate", C.c_double),
        ("minCornerDistanceRate", C.c_double),
        ("minDistanceToBorder", C.c_int),
        ("minMarkerDistanceRate", C.c_double),
        ("doCornerRefinement", C.c_int),
        ("cornerRefinementWinSize", C.c_int), ("cornerRefinementMaxIterations", C.c_int),
        ("cornerRefinementMinAccuracy", C.c_double),
        ("markerBorderBits", C.c_int),
        ("perspectiveRemovePixelPerCell", C.c_int),
        ("perspectiveRemoveIgnoredMarginPerCell", C.c_double),
        ("maxErroneousBitsInBorderRate", C.c_double),
        ("minOtsuStdDev", C.c_double),
        ("errorCorrectionRate", C.c_double),
    ]


class PoseMsg(C.Structure):
    _fields_ = [("position", C.c_double * 3), ("orientation", C.c_double * 4), ("covariance", C.c_double * 36)]


class MarkerMsg(C.Structure):
    _fields_ = [("id", C.c_int), ("pad", C.c_int), ("scale", C.c_double * 3), ("color", C.c_float * 4), ("position", C.c_double * 3),
                ("orientation", C.c_double * 4), ("lifetime_sec", C.c_double)]


class AslamError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"aslam error {code} ({E_NAMES.get(code, '?')}): {msg}")
        self.code = code


_P = C.POINTER
_u8p, _ip, _fp, _dp, _llp = _P(C.c_uint8), _P(C.c_int), _P(C.c_float), _P(C.c_double), _P(C.c_longlong)

_SIGS = {
    "aslam_default_init": (None, [_P(AslamInit)]),
    "aslam_create": (C.c_int, [_P(AslamInit), _P(C.c_void_p)]),
    "aslam_destroy": (None, [C.c_void_p]),
    "aslam_last_error": (C.c_char_p, [C.c_void_p]),
    "aslam_set_camera": (C.c_int, [C.c_void_p, _dp, _dp, C.c_int]),
    "aslam_get_pose_msg": (C.c_int, [C.c_void_p, C.c_void_p]),
    "aslam_get_map_markers": (C.c_int, [C.c_void_p, C.c_int, _ip, C.c_void_p]),
    "aslam_get_detected_markers": (C.c_int, [C.c_void_p, C.c_int, _ip, C.c_void_p]),
    "aslam_export_map_async": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
    "aslam_export_wait": (C.c_int, [C.c_void_p, C.c_int]),
    "aslam_draw_detected_markers": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_size_t]),
    "aslam_load_map_txt": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int, _ip, C.c_void_p]),
    "aslam_save_state": (C.c_int, [C.c_void_p, C.c_char_p]),
    "aslam_load_state": (C.c_int, [C.c_void_p, C.c_char_p]),
    "aslam_stream_open": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]),
    "aslam_stream_push": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_double, C.c_double, C.c_double]),
    "aslam_stream_acquire": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]),
    "aslam_stream_commit": (C.c_int, [C.c_void_p, C.c_double, C.c_double, C.c_double]),
    "aslam_stream_flush": (C.c_int, [C.c_void_p]),
    "aslam_default_detector_params": (None, [C.c_void_p]),
    "aslam_set_detector_params": (C.c_int, [C.c_void_p, C.c_void_p]),
    "aslam_set_dictionary": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "aslam_set_dictionary_bytes": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "aslam_add_encoder": (C.c_int, [C.c_void_p, C.c_double, C.c_double, C.c_double]),
    "aslam_add_image": (C.c_int, [C.c_void_p, _u8p, C.c_int, C.c_int, C.c_int, C.c_size_t]),
    "aslam_get_state": (C.c_int, [C.c_void_p, _ip, _dp, _dp]),
    "aslam_set_state": (C.c_int, [C.c_void_p, C.c_int, _dp, _dp, _ip]),
    "aslam_get_detections": (C.c_int, [C.c_void_p, _ip, _ip, _fp, _dp, _dp]),
    "aslam_get_observations": (C.c_int, [C.c_void_p, _ip, _ip, _ip, _ip, _dp, _dp]),
    "aslam_get_landmark_ids": (C.c_int, [C.c_void_p, _ip, _ip]),
    "aslam_stage_frames": (C.c_int, [C.c_void_p, C.c_int, _u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_size_t, C.c_size_t]),
    "aslam_stage_encoders": (C.c_int, [C.c_void_p, C.c_int, C.c_int, _dp, _dp, _dp]),
    "aslam_run_staged": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "aslam_sync": (C.c_int, [C.c_void_p]),
    "aslam_get_slot_detections": (C.c_int, [C.c_void_p, C.c_int, _ip, _ip, _fp, _dp, _dp]),
    "aslam_get_slot_raw_observations": (C.c_int, [C.c_void_p, C.c_int, _ip, _ip, _ip, _dp, _dp]),
    "aslam_get_slot_ekf_stats": (C.c_int, [C.c_void_p, C.c_int, C.c_int, _ip]),
    "aslam_detect_batch": (C.c_int, [C.c_void_p, _u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_size_t, C.c_size_t,
                                     C.c_int, _ip, _ip, _fp, _dp, _dp]),
    "aslam_export_map": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
    "aslam_comm_get_unique_id": (C.c_int, [C.c_void_p]),
    "aslam_comm_create": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int]),
    "aslam_comm_gather_maps": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
    "aslam_comm_destroy": (C.c_int, [C.c_void_p]),
    "aslam_debug_get_nbr": (C.c_int, [C.c_void_p, C.c_int, C.c_int, _u8p]),
    "aslam_debug_get_frame_counts": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_uint)]),
    "aslam_debug_get_contours": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_longlong, _ip, _ip, _ip, _ip, _llp]),
    "aslam_debug_get_candidates": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, _ip, _fp, _ip, _ip]),
    "aslam_debug_inject_observations": (C.c_int, [C.c_void_p, C.c_int, C.c_int, _ip, _ip, _dp, _dp]),
    "aslam_profile_enable": (C.c_int, [C.c_void_p, C.c_int]),
    "aslam_profile_reset": (C.c_int, [C.c_void_p]),
    "aslam_get_plan_stats": (C.c_int, [C.c_void_p, _llp]),
    "aslam_get_last_timing": (C.c_int, [C.c_void_p, _dp]),
    "aslam_profile_get": (C.c_int, [C.c_void_p, C.c_int, _P(C.c_char_p), _ip, _dp]),
    "aslam_synth_render": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, _dp, C.c_int, _ip, _dp, C.c_double, C.c_int,
                                     C.c_int, C.c_uint, C.c_int, _u8p]),
}

EXPORTED_SYMBOLS = tuple(_SIGS)
_lib = None


def lib_path():
    return os.environ.get("ARUCO_SLAM_LIB", DEFAULT_LIB)


def load():
    """Load the shared library (once).  Raises if it is missing — there is no fallback implementation."""
    global _lib
    if _lib is None:
        path = lib_path()
        if not os.path.exists(path):
            raise OSError(f"{path} not found: build it with `make -C aruco_slam_amd/csrc` (hipcc, gfx950)")
        _share_hip_runtime(path)
        lib = C.CDLL(path)
        for name, (res, args) in _SIGS.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def _share_hip_runtime(path):
    """One HIP runtime per process, whatever the import order.  The library needs `libamdhip64.so.7` (found in /opt/rocm by its
    RUNPATH); a PyTorch wheel bundles its own copy under the file name `libamdhip64.so`, which torch's libraries ask for by that
    name - so with the library loaded first, a later `import torch` would map a SECOND runtime and find no devices.  If torch is
    installed but not yet imported, its copy is mapped here first: the library then binds to it by soname, and torch finds its
    own file already loaded.  (torch imported first: nothing to do, the soname is already satisfied.)"""
    if os.path.basename(path) != "libaruco_slam_hip.so" or "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is not None and spec.origin:
        cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
        if os.path.exists(cand):
            C.CDLL(cand, mode=C.RTLD_GLOBAL)


def _ptr(a, typ):
    return a.ctypes.data_as(typ) if a is not None else None


def default_init(**over):
    init = AslamInit()
    load().aslam_default_init(C.byref(init))
    for k, v in over.items():
        if k in ("r2c_t", "r2c_q"):
            for i, x in enumerate(v):
                getattr(init, k)[i] = x
        else:
            setattr(init, k, v)
    return init


class Context:
    """One camera stream: thin RAII wrapper over aslam_ctx (mirrors the `ArucoSlam` class surface)."""

    def __init__(self, init=None, **over):
        self.lib = load()
        self.init = init if init is not None else default_init(**over)
        h = C.c_void_p()
        rc = self.lib.aslam_create(C.byref(self.init), C.byref(h))
        if rc != ASLAM_OK:
            raise AslamError(rc, "aslam_create failed (no usable HIP device, bad init, or out of memory)")
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.lib.aslam_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        if rc != ASLAM_OK:
            raise AslamError(rc, self.lib.aslam_last_error(self.h).decode())

    # -- reference class surface ---------------------------------------------------------------
    def set_camera(self, K, D=None):
        K = np.ascontiguousarray(K, dtype=np.float64).reshape(9)
        D = np.zeros(0) if D is None else np.ascontiguousarray(D, dtype=np.float64).reshape(-1)
        self._ck(self.lib.aslam_set_camera(self.h, _ptr(K, _dp), _ptr(D, _dp) if D.size else None, int(D.size)))

    # -- what the node publishes / persistence --------------------------------------------------
    def pose_msg(self):
        m = PoseMsg()
        self._ck(self.lib.aslam_get_pose_msg(self.h, C.byref(m)))
        return np.array(m.position), np.array(m.orientation), np.array(m.covariance).reshape(6, 6)

    def _markers(self, fn):
        n = C.c_int(0)
        self._ck(fn(self.h, 0, C.byref(n), None))
        arr = (MarkerMsg * max(n.value, 1))()
        self._ck(fn(self.h, n.value, C.byref(n), arr))
        return [dict(id=a.id, scale=tuple(a.scale), color=tuple(a.color), position=np.array(a.position), orientation=np.array(a.orientation),
                     lifetime=a.lifetime_sec) for a in arr[:n.value]]

    def map_markers(self):
        return self._markers(self.lib.aslam_get_map_markers)

    def detected_markers(self):
        return self._markers(self.lib.aslam_get_detected_markers)

    def draw_detected_markers(self, bgr):
        """markered_img_ of getObservations (aruco_slam.cpp:318-319): returns a copy of the bgr8 frame with the last detections drawn"""
        out = np.ascontiguousarray(bgr, dtype=np.uint8).copy()
        assert out.ndim == 3 and out.shape[2] == 3
        self._ck(self.lib.aslam_draw_detected_markers(self.h, out.ctypes.data_as(C.c_void_p), out.shape[0], out.shape[1], out.strides[0]))
        return out

    def export_map_async(self, device_ptr, buffer):
        self._ck(self.lib.aslam_export_map_async(self.h, C.c_void_p(int(device_ptr)), int(buffer)))

    def export_wait(self, buffer):
        self._ck(self.lib.aslam_export_wait(self.h, int(buffer)))

    def load_map_txt(self, path):
        n = C.c_int(0)
        self._ck(self.lib.aslam_load_map_txt(self.h, str(path).encode(), 0, C.byref(n), None))
        arr = (MarkerMsg * max(n.value, 1))()
        self._ck(self.lib.aslam_load_map_txt(self.h, str(path).encode(), n.value, C.byref(n), arr))
        return [dict(id=a.id, scale=tuple(a.scale), color=tuple(a.color), position=np.array(a.position), orientation=np.array(a.orientation),
                     lifetime=a.lifetime_sec) for a in arr[:n.value]]

    def save_state(self, path):
        self._ck(self.lib.aslam_save_state(self.h, str(path).encode()))

    def load_state(self, path):
        self._ck(self.lib.aslam_load_state(self.h, str(path).encode()))

    # -- host-fed stream (pinned ring, asynchronous upload) -------------------------------------
    def stream_open(self, rows, cols, channels, frames_per_submit):
        self._ck(self.lib.aslam_stream_open(self.h, int(rows), int(cols), int(channels), int(frames_per_submit)))

    def stream_push(self, img, wl, wr, dt):
        img = np.ascontiguousarray(img, dtype=np.uint8)
        self._ck(self.lib.aslam_stream_push(self.h, img.ctypes.data_as(C.c_void_p), img.strides[0], float(wl), float(wr), float(dt)))

    def stream_slot(self, rows, cols, channels=1):
        """numpy view of the next pinned slot (fill it, then stream_commit)"""
        p, st = C.c_void_p(), C.c_size_t()
        self._ck(self.lib.aslam_stream_acquire(self.h, C.byref(p), C.byref(st)))
        buf = (C.c_uint8 * (rows * st.value)).from_address(p.value)
        a = np.frombuffer(buf, np.uint8).reshape(rows, st.value)
        return a[:, :cols * channels].reshape((rows, cols) if channels == 1 else (rows, cols, channels))

    def stream_commit(self, wl, wr, dt):
        self._ck(self.lib.aslam_stream_commit(self.h, float(wl), float(wr), float(dt)))

    def stream_flush(self):
        self._ck(self.lib.aslam_stream_flush(self.h))

    def set_detector_params(self, **kw):
        """cv::aruco::DetectorParameters fields by name; anything not given keeps its OpenCV 3.2.0 default"""
        p = DetectorParams()
        self.lib.aslam_default_detector_params(C.byref(p))
        for k, v in kw.items():
            if not hasattr(p, k):
                raise KeyError(k)
            setattr(p, k, v)
        self._ck(self.lib.aslam_set_detector_params(self.h, C.byref(p)))

    def set_dictionary(self, bits, max_correction_bits=0):
        """bits: n x ms x ms (1 = white); replaces the built-in DICT_ARUCO_ORIGINAL"""
        b = np.ascontiguousarray(bits, dtype=np.uint8)
        self._ck(self.lib.aslam_set_dictionary(self.h, int(b.shape[1]), int(b.shape[0]), int(max_correction_bits), b.ctypes.data_as(C.c_void_p)))

    def set_dictionary_bytes(self, bytes_list, marker_size, max_correction_bits=0):
        """bytes_list: cv::aruco::Dictionary::bytesList as an n x nbytes x 4 uint8 array"""
        b = np.ascontiguousarray(bytes_list, dtype=np.uint8)
        self._ck(self.lib.aslam_set_dictionary_bytes(self.h, int(marker_size), int(b.shape[0]), int(max_correction_bits), b.ctypes.data_as(C.c_void_p)))

    def add_encoder(self, wl, wr, t_now):
        self._ck(self.lib.aslam_add_encoder(self.h, float(wl), float(wr), float(t_now)))

    def add_image(self, img):
        img = np.ascontiguousarray(img, dtype=np.uint8)
        rows, cols = img.shape[:2]
        ch = 1 if img.ndim == 2 else img.shape[2]
        self._ck(self.lib.aslam_add_image(self.h, _ptr(img, _u8p), rows, cols, ch, cols * ch))

    def get_state(self):
        n = C.c_int()
        self._ck(self.lib.aslam_get_state(self.h, C.byref(n), None, None))
        N = n.value
        mu = np.zeros(N)
        sigma = np.zeros((N, N), order="F")
        self._ck(self.lib.aslam_get_state(self.h, C.byref(n), _ptr(mu, _dp), _ptr(sigma, _dp)))
        return mu, np.array(sigma)

    def set_state(self, mu, sigma, landmark_ids):
        mu = np.ascontiguousarray(mu, dtype=np.float64)
        sig = np.asfortranarray(sigma, dtype=np.float64)
        ids = np.ascontiguousarray(landmark_ids, dtype=np.int32)
        self._ck(self.lib.aslam_set_state(self.h, int(mu.size), _ptr(mu, _dp), _ptr(sig, _dp), _ptr(ids, _ip) if ids.size else None))

    def _detections(self, fn, *pre):
        m = C.c_int()
        ids = np.zeros(MARKER_MAX, np.int32)
        corners = np.zeros((MARKER_MAX, 4, 2), np.float32)
        rv = np.zeros((MARKER_MAX, 3))
        tv = np.zeros((MARKER_MAX, 3))
        self._ck(fn(self.h, *pre, C.byref(m), _ptr(ids, _ip), _ptr(corners, _fp), _ptr(rv, _dp), _ptr(tv, _dp)))
        M = m.value
        return ids[:M].copy(), corners[:M].copy(), rv[:M].copy(), tv[:M].copy()

    def get_detections(self):
        return self._detections(self.lib.aslam_get_detections)

    def get_slot_detections(self, slot):
        return self._detections(self.lib.aslam_get_slot_detections, int(slot))

    def get_observations(self):
        n = C.c_int()
        ids = np.zeros(MARKER_MAX, np.int32); idx = np.zeros(MARKER_MAX, np.int32); act = np.zeros(MARKER_MAX, np.int32)
        xyth = np.zeros((MARKER_MAX, 3)); R = np.zeros((MARKER_MAX, 3))
        self._ck(self.lib.aslam_get_observations(self.h, C.byref(n), _ptr(ids, _ip), _ptr(idx, _ip), _ptr(act, _ip), _ptr(xyth, _dp), _ptr(R, _dp)))
        k = n.value
        return ids[:k].copy(), idx[:k].copy(), act[:k].copy(), xyth[:k].copy(), R[:k].copy()

    def get_slot_raw_observations(self, slot):
        n = C.c_int()
        ids = np.zeros(MARKER_MAX, np.int32); valid = np.zeros(MARKER_MAX, np.int32)
        xyth = np.zeros((MARKER_MAX, 3)); R = np.zeros((MARKER_MAX, 3))
        self._ck(self.lib.aslam_get_slot_raw_observations(self.h, int(slot), C.byref(n), _ptr(ids, _ip), _ptr(valid, _ip), _ptr(xyth, _dp), _ptr(R, _dp)))
        k = n.value
        return ids[:k].copy(), valid[:k].copy(), xyth[:k].copy(), R[:k].copy()

    def get_slot_ekf_stats(self, first, count):
        """count x 4 ints: markers detected, landmarks appended, corrections fused, stationary no-ops of every slot's EKF step"""
        st = np.zeros((int(count), 4), np.int32)
        self._ck(self.lib.aslam_get_slot_ekf_stats(self.h, int(first), int(count), _ptr(st, _ip)))
        return st

    def get_landmark_ids(self):
        n = C.c_int()
        ids = np.zeros(max(int(self.init.max_landmarks), 1), np.int32)
        self._ck(self.lib.aslam_get_landmark_ids(self.h, C.byref(n), _ptr(ids, _ip)))
        return ids[: n.value].copy()

    # -- staged (device-resident) stream API ------------------------------------------------------
    def stage_frames(self, frames, slot0=0):
        frames = np.ascontiguousarray(frames, dtype=np.uint8)
        if frames.ndim == 2 or (frames.ndim == 3 and frames.shape[-1] == 3):
            frames = frames[None]                      # one gray (rows, cols) or one bgr8 (rows, cols, 3) frame
        n, rows, cols = frames.shape[:3]
        ch = 1 if frames.ndim == 3 else frames.shape[3]
        self._ck(self.lib.aslam_stage_frames(self.h, int(slot0), _ptr(frames, _u8p), n, rows, cols, ch, cols * ch, rows * cols * ch))

    def stage_encoders(self, wl, wr, dt, slot0=0):
        wl = np.ascontiguousarray(wl, dtype=np.float64); wr = np.ascontiguousarray(wr, dtype=np.float64)
        dt = np.ascontiguousarray(dt, dtype=np.float64)
        self._ck(self.lib.aslam_stage_encoders(self.h, int(slot0), int(wl.size), _ptr(wl, _dp), _ptr(wr, _dp), _ptr(dt, _dp)))

    def run_staged(self, first, count, with_ekf=True):
        """with_ekf: False/0 detection + pose only, True/1 full path, 2 EKF steps only (injected observations)"""
        self._ck(self.lib.aslam_run_staged(self.h, int(first), int(count), int(with_ekf)))

    def inject_observations(self, slot, ids, valid, xyth, Rdiag):
        ids = np.ascontiguousarray(ids, dtype=np.int32); valid = np.ascontiguousarray(valid, dtype=np.int32)
        xyth = np.ascontiguousarray(xyth, dtype=np.float64).reshape(-1, 3); Rdiag = np.ascontiguousarray(Rdiag, dtype=np.float64).reshape(-1, 3)
        self._ck(self.lib.aslam_debug_inject_observations(self.h, int(slot), int(ids.size), _ptr(ids, _ip), _ptr(valid, _ip), _ptr(xyth, _dp), _ptr(Rdiag, _dp)))

    def sync(self):
        self._ck(self.lib.aslam_sync(self.h))

    def detect_batch(self, frames, max_per_frame=64):
        frames = np.ascontiguousarray(frames, dtype=np.uint8)
        n, rows, cols = frames.shape[:3]
        ch = 1 if frames.ndim == 3 else frames.shape[3]
        counts = np.zeros(n, np.int32)
        ids = np.full((n, max_per_frame), -1, np.int32)
        corners = np.zeros((n, max_per_frame, 4, 2), np.float32)
        rv = np.zeros((n, max_per_frame, 3)); tv = np.zeros((n, max_per_frame, 3))
        self._ck(self.lib.aslam_detect_batch(self.h, _ptr(frames, _u8p), n, rows, cols, ch, cols * ch, rows * cols * ch,
                                             max_per_frame, _ptr(counts, _ip), _ptr(ids, _ip), _ptr(corners, _fp), _ptr(rv, _dp), _ptr(tv, _dp)))
        return counts, ids, corners, rv, tv

    def export_map(self):
        buf = np.zeros(int(self.init.max_landmarks) * MAP_RECORD_BYTES, np.uint8)
        self._ck(self.lib.aslam_export_map(self.h, buf.ctypes.data_as(C.c_void_p), 0))
        return buf

    # -- map gather over RCCL through the C-ABI (no torch) ----------------------------------------------------
    @staticmethod
    def comm_unique_id():
        buf = (C.c_uint8 * 128)()
        rc = load().aslam_comm_get_unique_id(buf)
        if rc != ASLAM_OK:
            raise AslamError(rc, "aslam_comm_get_unique_id failed (librccl.so not found?)")
        return bytes(buf)

    def comm_create(self, uid, world, rank):
        buf = (C.c_uint8 * 128).from_buffer_copy(uid)
        self._ck(self.lib.aslam_comm_create(self.h, buf, int(world), int(rank)))
        self._comm_world = int(world)

    def comm_gather_maps(self):
        out = np.zeros(self._comm_world * int(self.init.max_landmarks) * MAP_RECORD_BYTES, np.uint8)
        self._ck(self.lib.aslam_comm_gather_maps(self.h, out.ctypes.data_as(C.c_void_p), 0))
        return out

    def comm_destroy(self):
        self._ck(self.lib.aslam_comm_destroy(self.h))

    def export_map_to_device(self, device_ptr):
        self._ck(self.lib.aslam_export_map(self.h, C.c_void_p(int(device_ptr)), 1))

    # -- instrumentation ------------------------------------------------------------------------------
    def debug_nbr(self, slot, scale, rows, cols):
        out = np.zeros((rows, cols), np.uint8)
        self._ck(self.lib.aslam_debug_get_nbr(self.h, int(slot), int(scale), _ptr(out, _u8p)))
        return out

    def debug_frame_counts(self, slot):
        """list sizes of one slot after its last detection pass"""
        out = (C.c_uint * 6)()
        self._ck(self.lib.aslam_debug_get_frame_counts(self.h, int(slot), out))
        return dict(zip(("nodes", "contours", "points", "write_tickets", "quad_candidates", "serial_link"), [int(v) for v in out]))

    def debug_contours(self, slot, scale, max_contours=20000, max_points=4_000_000):
        n = C.c_int(); tot = C.c_longlong()
        sizes = np.zeros(max_contours, np.int32); keys = np.zeros(max_contours, np.int32)
        pts = np.zeros((max_points, 2), np.int32)
        self._ck(self.lib.aslam_debug_get_contours(self.h, int(slot), int(scale), max_contours, max_points, C.byref(n),
                                                   _ptr(sizes, _ip), _ptr(keys, _ip), _ptr(pts, _ip), C.byref(tot)))
        k = n.value
        return sizes[:k].copy(), keys[:k].copy(), pts[: tot.value].copy()

    def debug_candidates(self, slot, stage):
        n = C.c_int()
        corners = np.zeros((CAND_MAX, 4, 2), np.float32); sizes = np.zeros(CAND_MAX, np.int32); ids = np.zeros(CAND_MAX, np.int32)
        self._ck(self.lib.aslam_debug_get_candidates(self.h, int(slot), int(stage), CAND_MAX, C.byref(n), _ptr(corners, _fp), _ptr(sizes, _ip), _ptr(ids, _ip)))
        k = n.value
        return corners[:k].copy(), sizes[:k].copy(), ids[:k].copy()

    def profile_enable(self, on=True):
        self._ck(self.lib.aslam_profile_enable(self.h, 1 if on else 0))

    def profile_reset(self):
        self._ck(self.lib.aslam_profile_reset(self.h))

    def last_timing(self):
        """host-clock breakdown of the last add_image, microseconds"""
        out = np.zeros(6)
        self._ck(self.lib.aslam_get_last_timing(self.h, _ptr(out, _dp)))
        return dict(zip(("upload", "enqueue_detect", "enqueue_ekf", "wait", "readback", "total"), out.tolist()))

    def plan_stats(self):
        """frames fused inside windows / on the per-frame chain, windows formed, frames left to the device's own plan (since profile_reset)"""
        out = np.zeros(4, np.int64)
        self._ck(self.lib.aslam_get_plan_stats(self.h, _ptr(out, _llp)))
        return dict(frames_in_windows=int(out[0]), frames_per_frame_chain=int(out[1]), windows=int(out[2]), frames_device_planned=int(out[3]))

    def profile_get(self):
        names = (C.c_char_p * 32)(); calls = np.zeros(32, np.int32); ms = np.zeros(32)
        n = self.lib.aslam_profile_get(self.h, 32, names, _ptr(calls, _ip), _ptr(ms, _dp))
        return {names[i].decode(): (int(calls[i]), float(ms[i])) for i in range(n)}

    def synth_render(self, slot, rows, cols, K, ids, poses, marker_length=0.27, background=128, noise_amp=0, seed=0,
                     supersample=4, download=True):
        K = np.ascontiguousarray(K, dtype=np.float64).reshape(9)
        ids = np.ascontiguousarray(ids, dtype=np.int32)
        poses = np.ascontiguousarray(poses, dtype=np.float64).reshape(-1, 12)
        out = np.zeros((rows, cols), np.uint8) if download else None
        self._ck(self.lib.aslam_synth_render(self.h, int(slot), rows, cols, _ptr(K, _dp), int(ids.size), _ptr(ids, _ip) if ids.size else None,
                                             _ptr(poses, _dp) if ids.size else None, float(marker_length), int(background), int(noise_amp),
                                             int(seed), int(supersample), _ptr(out, _u8p)))
        return out
