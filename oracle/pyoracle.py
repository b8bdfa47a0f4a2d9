"""ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/oracle.h).  ctypes binding of oracle/_build/liboracle.so.

Imported only by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg — never by the product
package aruco_slam_amd.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(_HERE, "_build", "liboracle.so")
_P = C.POINTER
_u8p, _ip, _fp, _dp, _llp = _P(C.c_uint8), _P(C.c_int), _P(C.c_float), _P(C.c_double), _P(C.c_longlong)
_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            build()
        L = C.CDLL(LIB)
        L.orc_norm_angle.restype = C.c_double
        L.orc_norm_angle.argtypes = [C.c_double]
        L.orc_slam_create.restype = C.c_void_p
        L.orc_slam_create.argtypes = [_dp, C.c_float, C.c_int]
        for name in ("orc_slam_destroy", "orc_slam_set_camera", "orc_slam_add_encoder", "orc_slam_add_image", "orc_slam_add_poses",
                     "orc_slam_get_state", "orc_slam_set_state"):
            getattr(L, name).restype = None
        L.orc_slam_destroy.argtypes = [C.c_void_p]
        L.orc_slam_set_camera.argtypes = [C.c_void_p, _dp, _dp, C.c_int]
        L.orc_slam_add_encoder.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_double]
        L.orc_slam_set_dictionary.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.orc_set_dictionary.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.orc_set_detector_params.argtypes = [_dp]
        L.orc_corner_sub_pix.argtypes = [_u8p, C.c_int, C.c_int, _fp, C.c_int, C.c_int, C.c_int, C.c_double]
        L.orc_slam_set_detector_params.argtypes = [C.c_void_p, _dp]
        L.orc_slam_add_image.argtypes = [C.c_void_p, _u8p, C.c_int, C.c_int, C.c_int, C.c_size_t]
        L.orc_slam_add_poses.argtypes = [C.c_void_p, C.c_int, _ip, _fp, _dp, _dp]
        L.orc_slam_state_size.argtypes = [C.c_void_p]
        L.orc_slam_get_state.argtypes = [C.c_void_p, _dp, _dp]
        L.orc_slam_set_state.argtypes = [C.c_void_p, C.c_int, _dp, _dp, _ip]
        L.orc_slam_landmark_ids.argtypes = [C.c_void_p, _ip]
        L.orc_slam_log_detections.argtypes = [C.c_void_p, C.c_int, _ip, _fp, _dp, _dp]
        L.orc_slam_log_observations.argtypes = [C.c_void_p, C.c_int, _ip, _ip, _ip, _dp, _dp]
        L.orc_find_contours.argtypes = [_u8p, C.c_int, C.c_int, C.c_int, C.c_longlong, _ip, _ip, _ip, _ip, _llp]
        L.orc_threshold.argtypes = [_u8p, C.c_int, C.c_int, C.c_int, C.c_double, _u8p]
        L.orc_box_mean.argtypes = [_u8p, C.c_int, C.c_int, C.c_int, _u8p]
        L.orc_bgr2gray.argtypes = [_u8p, C.c_int, C.c_int, C.c_size_t, _u8p]
        L.orc_approx_poly.argtypes = [_ip, C.c_int, C.c_double, _ip, C.c_int]
        L.orc_candidates.argtypes = [_u8p, C.c_int, C.c_int, C.c_int, C.c_int, _fp, _ip, _ip, _ip]
        L.orc_extract_bits.argtypes = [_u8p, C.c_int, C.c_int, _fp, _u8p]
        L.orc_identify.argtypes = [_u8p, C.c_int, C.c_int, _fp, _ip]
        L.orc_detect.argtypes = [_u8p, C.c_int, C.c_int, C.c_int, C.c_size_t, C.c_int, _ip, _fp]
        L.orc_dict_bytes.argtypes = [_u8p]
        L.orc_dict_bits.argtypes = [C.c_int, _u8p]
        L.orc_rodrigues.argtypes = [_dp, _dp, _dp]
        L.orc_rodrigues_inv.argtypes = [_dp, _dp]
        L.orc_project_points.argtypes = [_dp, C.c_int, _dp, _dp, _dp, _dp, C.c_int, _dp, _dp, _dp]
        L.orc_solve_pnp.argtypes = [_fp, C.c_float, _dp, _dp, C.c_int, _dp, _dp, _ip]
        L.orc_heap_order.argtypes = [C.c_int, _ip, _ip]
        L.orc_perspective_transform.argtypes = [_fp, _fp, _dp]
        _lib = L
    return _lib


def _p(a, t):
    return a.ctypes.data_as(t)


def norm_angle(a):
    return load().orc_norm_angle(float(a))


def threshold(gray, k, C_=7.0):
    gray = np.ascontiguousarray(gray, np.uint8)
    out = np.zeros_like(gray)
    load().orc_threshold(_p(gray, _u8p), gray.shape[0], gray.shape[1], int(k), float(C_), _p(out, _u8p))
    return out


def box_mean(gray, k):
    gray = np.ascontiguousarray(gray, np.uint8)
    out = np.zeros_like(gray)
    load().orc_box_mean(_p(gray, _u8p), gray.shape[0], gray.shape[1], int(k), _p(out, _u8p))
    return out


def bgr2gray(bgr):
    bgr = np.ascontiguousarray(bgr, np.uint8)
    out = np.zeros(bgr.shape[:2], np.uint8)
    load().orc_bgr2gray(_p(bgr, _u8p), bgr.shape[0], bgr.shape[1], bgr.shape[1] * 3, _p(out, _u8p))
    return out


def find_contours(binimg, max_contours=400000, max_points=8_000_000):
    """contours in OpenCV output order: list of (sizes, keys, is_hole, points[n,2])"""
    binimg = np.ascontiguousarray(binimg, np.uint8)
    sizes = np.zeros(max_contours, np.int32); keys = np.zeros(max_contours, np.int32); hole = np.zeros(max_contours, np.int32)
    pts = np.zeros((max_points, 2), np.int32)
    tot = C.c_longlong()
    n = load().orc_find_contours(_p(binimg, _u8p), binimg.shape[0], binimg.shape[1], max_contours, max_points,
                                 _p(sizes, _ip), _p(keys, _ip), _p(hole, _ip), _p(pts, _ip), C.byref(tot))
    if n < 0:
        raise RuntimeError("oracle contour buffers too small")
    return sizes[:n].copy(), keys[:n].copy(), hole[:n].copy(), pts[: tot.value].copy()


def approx_poly(pts, eps):
    pts = np.ascontiguousarray(pts, np.int32)
    out = np.zeros((max(len(pts), 1), 2), np.int32)
    n = load().orc_approx_poly(_p(pts, _ip), len(pts), float(eps), _p(out, _ip), len(out))
    return out[:n].copy()


def candidates(gray, stage, maxn=8192):
    gray = np.ascontiguousarray(gray, np.uint8)
    corners = np.zeros((maxn, 4, 2), np.float32); sizes = np.zeros(maxn, np.int32); scales = np.zeros(maxn, np.int32); keys = np.zeros(maxn, np.int32)
    n = load().orc_candidates(_p(gray, _u8p), gray.shape[0], gray.shape[1], int(stage), maxn, _p(corners, _fp), _p(sizes, _ip), _p(scales, _ip), _p(keys, _ip))
    return corners[:n].copy(), sizes[:n].copy(), scales[:n].copy(), keys[:n].copy()


def extract_bits(gray, corners):
    gray = np.ascontiguousarray(gray, np.uint8)
    c = np.ascontiguousarray(corners, np.float32).reshape(8)
    bits = np.zeros(49, np.uint8)
    load().orc_extract_bits(_p(gray, _u8p), gray.shape[0], gray.shape[1], _p(c, _fp), _p(bits, _u8p))
    return bits.reshape(7, 7)


def identify(gray, corners):
    gray = np.ascontiguousarray(gray, np.uint8)
    c = np.ascontiguousarray(corners, np.float32).reshape(8).copy()
    idv = C.c_int(-1)
    ok = load().orc_identify(_p(gray, _u8p), gray.shape[0], gray.shape[1], _p(c, _fp), C.byref(idv))
    return bool(ok), idv.value, c.reshape(4, 2)


def detect(img, maxn=1024):
    img = np.ascontiguousarray(img, np.uint8)
    ch = 1 if img.ndim == 2 else img.shape[2]
    ids = np.zeros(maxn, np.int32); corners = np.zeros((maxn, 4, 2), np.float32)
    n = load().orc_detect(_p(img, _u8p), img.shape[0], img.shape[1], ch, img.shape[1] * ch, maxn, _p(ids, _ip), _p(corners, _fp))
    return ids[:n].copy(), corners[:n].copy()


PARAM_ORDER = ("adaptiveThreshWinSizeMin", "adaptiveThreshWinSizeMax", "adaptiveThreshWinSizeStep", "adaptiveThreshConstant",
               "minMarkerPerimeterRate", "maxMarkerPerimeterRate", "polygonalApproxAccuracyRate", "minCornerDistanceRate",
               "minDistanceToBorder", "minMarkerDistanceRate", "markerBorderBits", "perspectiveRemovePixelPerCell",
               "perspectiveRemoveIgnoredMarginPerCell", "maxErroneousBitsInBorderRate", "minOtsuStdDev", "errorCorrectionRate",
               "doCornerRefinement", "cornerRefinementWinSize", "cornerRefinementMaxIterations", "cornerRefinementMinAccuracy")
PARAM_DEFAULTS = (3, 23, 10, 7.0, 0.03, 4.0, 0.05, 0.05, 3, 0.05, 1, 8, 0.13, 0.35, 5.0, 0.6, 0, 5, 30, 0.1)


def _param_vector(kw):
    v = dict(zip(PARAM_ORDER, PARAM_DEFAULTS))
    for k in kw:
        if k not in v:
            raise KeyError(k)
    v.update(kw)
    return np.array([float(v[k]) for k in PARAM_ORDER])


def set_detector_params(**kw):
    """cv::aruco::DetectorParameters for the free functions; no arguments = OpenCV 3.2.0 defaults"""
    v = _param_vector(kw)
    load().orc_set_detector_params(_p(v, _dp))


def corner_sub_pix(gray, corners, win=5, max_iter=30, eps=0.1):
    gray = np.ascontiguousarray(gray, np.uint8)
    c = np.ascontiguousarray(corners, np.float32).reshape(-1, 2).copy()
    load().orc_corner_sub_pix(_p(gray, _u8p), gray.shape[0], gray.shape[1], _p(c, _fp), int(c.shape[0]), int(win), int(max_iter), float(eps))
    return c


def set_dictionary(bits, max_correction_bits=0):
    """bits: n x ms x ms array (1 = white) or None for the built-in DICT_ARUCO_ORIGINAL (free functions only)"""
    L = load()
    if bits is None:
        L.orc_set_dictionary(0, 0, 0, None)
        return
    b = np.ascontiguousarray(bits, np.uint8)
    L.orc_set_dictionary(int(b.shape[1]), int(b.shape[0]), int(max_correction_bits), b.ctypes.data_as(C.c_void_p))


def dict_bits(i):
    b = np.zeros(25, np.uint8)
    load().orc_dict_bits(int(i), _p(b, _u8p))
    return b.reshape(5, 5)


def dict_bytes():
    b = np.zeros(1024 * 16, np.uint8)
    load().orc_dict_bytes(_p(b, _u8p))
    return b.reshape(1024, 4, 4)


def rodrigues(r):
    r = np.ascontiguousarray(r, np.float64)
    R = np.zeros(9); J = np.zeros(27)
    load().orc_rodrigues(_p(r, _dp), _p(R, _dp), _p(J, _dp))
    return R.reshape(3, 3), J.reshape(3, 9)


def rodrigues_inv(R):
    R = np.ascontiguousarray(R, np.float64).reshape(9)
    r = np.zeros(3)
    load().orc_rodrigues_inv(_p(R, _dp), _p(r, _dp))
    return r


def project_points(obj, r, t, K, D):
    obj = np.ascontiguousarray(obj, np.float64); n = len(obj)
    r = np.ascontiguousarray(r, np.float64); t = np.ascontiguousarray(t, np.float64)
    K = np.ascontiguousarray(K, np.float64).reshape(9); D = np.ascontiguousarray(D, np.float64)
    out = np.zeros((n, 2)); dr = np.zeros((2 * n, 3)); dt = np.zeros((2 * n, 3))
    load().orc_project_points(_p(obj, _dp), n, _p(r, _dp), _p(t, _dp), _p(K, _dp), _p(D, _dp), int(D.size), _p(out, _dp), _p(dr, _dp), _p(dt, _dp))
    return out, dr, dt


def solve_pnp(corners, L, K, D):
    c = np.ascontiguousarray(corners, np.float32).reshape(8)
    K = np.ascontiguousarray(K, np.float64).reshape(9); D = np.ascontiguousarray(D, np.float64)
    rv = np.zeros(3); tv = np.zeros(3); it = C.c_int()
    load().orc_solve_pnp(_p(c, _fp), C.c_float(L), _p(K, _dp), _p(D, _dp), int(D.size), _p(rv, _dp), _p(tv, _dp), C.byref(it))
    return rv, tv, it.value


def heap_order(indices):
    idx = np.ascontiguousarray(indices, np.int32)
    out = np.zeros(len(idx), np.int32)
    load().orc_heap_order(len(idx), _p(idx, _ip), _p(out, _ip))
    return out


def perspective_transform(src, dst):
    s = np.ascontiguousarray(src, np.float32).reshape(8); d = np.ascontiguousarray(dst, np.float32).reshape(8)
    M = np.zeros(9)
    load().orc_perspective_transform(_p(s, _fp), _p(d, _fp), _p(M, _dp))
    return M.reshape(3, 3)


class Slam:
    """CPU restatement of the reference's ArucoSlam (oracle/ekf.cpp)."""

    def __init__(self, Q_k=0.01, R_x=100, R_y=100, R_theta=10, kl=0.05, kr=0.05, b=0.09, marker_length=0.27,
                 r2c_tx=0.0, r2c_ty=0.0, useful_distance_threshold=3.0, literal=True):
        p = np.array([Q_k, R_x, R_y, R_theta, kl, kr, b, marker_length, r2c_tx, r2c_ty], np.float64)
        self.L = load()
        self.h = C.c_void_p(self.L.orc_slam_create(_p(p, _dp), C.c_float(useful_distance_threshold), 1 if literal else 0))

    def __del__(self):
        if getattr(self, "h", None):
            self.L.orc_slam_destroy(self.h)
            self.h = None

    def set_camera(self, K, D):
        K = np.ascontiguousarray(K, np.float64).reshape(9); D = np.ascontiguousarray(D, np.float64)
        self.L.orc_slam_set_camera(self.h, _p(K, _dp), _p(D, _dp), int(D.size))

    def set_detector_params(self, **kw):
        v = _param_vector(kw)
        self.L.orc_slam_set_detector_params(self.h, _p(v, _dp))

    def set_dictionary(self, bits, max_correction_bits=0):
        b = np.ascontiguousarray(bits, np.uint8)
        self.L.orc_slam_set_dictionary(self.h, int(b.shape[1]), int(b.shape[0]), int(max_correction_bits), b.ctypes.data_as(C.c_void_p))

    def add_encoder(self, wl, wr, t):
        self.L.orc_slam_add_encoder(self.h, float(wl), float(wr), float(t))

    def add_image(self, img):
        img = np.ascontiguousarray(img, np.uint8)
        ch = 1 if img.ndim == 2 else img.shape[2]
        self.L.orc_slam_add_image(self.h, _p(img, _u8p), img.shape[0], img.shape[1], ch, img.shape[1] * ch)

    def add_poses(self, ids, corners, rvecs, tvecs):
        ids = np.ascontiguousarray(ids, np.int32); c = np.ascontiguousarray(corners, np.float32)
        rv = np.ascontiguousarray(rvecs, np.float64); tv = np.ascontiguousarray(tvecs, np.float64)
        self.L.orc_slam_add_poses(self.h, len(ids), _p(ids, _ip), _p(c, _fp), _p(rv, _dp), _p(tv, _dp))

    def get_state(self):
        N = self.L.orc_slam_state_size(self.h)
        mu = np.zeros(N); sig = np.zeros((N, N), order="F")
        self.L.orc_slam_get_state(self.h, _p(mu, _dp), _p(sig, _dp))
        return mu, np.array(sig)

    def set_state(self, mu, sigma, landmark_ids):
        mu = np.ascontiguousarray(mu, np.float64); sig = np.asfortranarray(sigma, np.float64)
        ids = np.ascontiguousarray(landmark_ids, np.int32)
        self.L.orc_slam_set_state(self.h, int(mu.size), _p(mu, _dp), _p(sig, _dp), _p(ids, _ip))

    def landmark_ids(self):
        N = self.L.orc_slam_state_size(self.h)
        ids = np.zeros(max((N - 3) // 3, 1), np.int32)
        n = self.L.orc_slam_landmark_ids(self.h, _p(ids, _ip))
        return ids[:n].copy()

    def log_detections(self, maxn=1024):
        ids = np.zeros(maxn, np.int32); c = np.zeros((maxn, 4, 2), np.float32); rv = np.zeros((maxn, 3)); tv = np.zeros((maxn, 3))
        n = self.L.orc_slam_log_detections(self.h, maxn, _p(ids, _ip), _p(c, _fp), _p(rv, _dp), _p(tv, _dp))
        return ids[:n].copy(), c[:n].copy(), rv[:n].copy(), tv[:n].copy()

    def log_observations(self, maxn=1024):
        ids = np.zeros(maxn, np.int32); idx = np.zeros(maxn, np.int32); act = np.zeros(maxn, np.int32)
        xyth = np.zeros((maxn, 3)); R = np.zeros((maxn, 9))
        n = self.L.orc_slam_log_observations(self.h, maxn, _p(ids, _ip), _p(idx, _ip), _p(act, _ip), _p(xyth, _dp), _p(R, _dp))
        return ids[:n].copy(), idx[:n].copy(), act[:n].copy(), xyth[:n].copy(), R[:n].reshape(-1, 3, 3).copy()
