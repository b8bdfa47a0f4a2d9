"""Deterministic synthetic scenes for the ArUco EKF-SLAM hot path (inputs for tests and bench.py).

The reference ships no data; its only scenario is a Gazebo world from packages that are not in
the repository (launch/slam.launch:11-41).  This module builds the measurement scene of SURVEY.md §8(d):
a world of L planar markers arranged in panels of M, a differential-drive robot whose noise-free wheel
speeds reproduce the true trajectory through the reference's own motion model (aruco_slam.cpp:35-52), and per
frame exactly M markers in view, each < 3 m from the camera (the range gate, aruco_slam.cpp:327-333).

Conventions (reference observation model, aruco_slam.cpp:359-361): robot frame x forward / y left; camera
optical frame x right / y down / z forward, mounted at r2c = (tx, ty) in the robot frame;
observation = (t_z + r2c.x, -t_x + r2c.y, atan2(-R02, R22)).  A marker yawed by psi about the vertical and
facing the camera has R = [[c,0,-s],[0,-1,0],[-s,0,-c]] and is observed at theta = pi - psi.
"""
import math
from dataclasses import dataclass, field

import numpy as np

_WORDS = (0x10, 0x17, 0x09, 0x0E)


def aruco_original_bits(marker_id):
    """5x5 bit matrix (1 = white) of DICT_ARUCO_ORIGINAL id: each row carries two id bits, MSB first."""
    b = np.zeros((5, 5), np.uint8)
    for y in range(5):
        val = _WORDS[(marker_id >> (2 * (4 - y))) & 3]
        for x in range(5):
            b[y, x] = (val >> (4 - x)) & 1
    return b


def _code(b):
    v = 0
    for bit in b.reshape(-1):
        v = (v << 1) | int(bit)
    return v


def unambiguous_ids(count, start=1):
    """ids whose four rotations are pairwise distinct and collide with no rotation of a smaller id, so that
    Dictionary::identify (first match in id order) returns exactly the rendered id and rotation."""
    seen = set()
    out = []
    for m in range(1024):
        b = aruco_original_bits(m)
        rots = [_code(np.rot90(b, k)) for k in range(4)]
        ok = len(set(rots)) == 4 and not any(r in seen for r in rots)
        seen.update(rots)
        if ok and m >= start:
            out.append(m)
            if len(out) == count:
                break
    if len(out) < count:
        raise ValueError("dictionary too small for the requested number of ids")
    return out


def camera_matrix(rows, cols, f):
    return np.array([[f, 0, (cols - 1) / 2.0], [0, f, (rows - 1) / 2.0], [0, 0, 1.0]])


def marker_pose(t_cam, psi):
    """(R row-major 3x3, t) of a marker at camera-frame position t_cam, yawed by psi about the vertical."""
    c, s = math.cos(psi), math.sin(psi)
    R = np.array([[c, 0.0, -s], [0.0, -1.0, 0.0], [-s, 0.0, -c]])
    return R, np.asarray(t_cam, dtype=np.float64)


def norm_angle(a):
    if a >= math.pi:
        a -= 2 * math.pi
    if a < -math.pi:
        a += 2 * math.pi
    return a


@dataclass
class Frame:
    ids: np.ndarray            # marker ids in view
    poses: np.ndarray          # n x 12: R (row-major) then t, marker -> camera
    wl: float                  # encoder sample that precedes this frame
    wr: float
    dt: float
    true_pose: tuple           # (x, y, theta) of the robot
    landmark_index: np.ndarray = field(default=None)  # world landmark index of every visible marker


@dataclass
class SceneConfig:
    rows: int = 720
    cols: int = 1280
    f: float = 900.0
    grid: tuple = (5, 4)           # columns x rows of markers per panel (M = 20)
    n_panels: int = 10             # L = n_panels * M landmarks
    marker_length: float = 0.27    # parameters.yaml:17
    col_spacing: float = 0.45
    row_spacing: float = 0.345
    tz_far: float = 2.65
    tz_near: float = 1.87
    step: float = 0.02             # robot advance per frame (> 1 cm so that no update is "stationary")
    dt: float = 1.0 / 30.0
    kl: float = 0.05
    kr: float = 0.05
    b: float = 0.09
    r2c: tuple = (0.0, 0.0)
    max_yaw_deg: float = 25.0
    seed: int = 1
    # cv::aruco::DetectorParameters fields that differ from the OpenCV 3.2.0 defaults for this scene (applied to the library
    # AND to the oracle by whoever drives the scene: apply_detector())
    detector: dict = field(default_factory=dict)
    # "panel": PanelWorld (the visible set is constant over a panel approach); "ring": RingWorld (the visible set slides)
    kind: str = "panel"
    ring_radius: float = 2.6       # RingWorld: radius of the landmark ring
    ring_robot_radius: float = 0.3  # ... and of the robot's own circle inside it
    ring_lap_frames: int = 500     # frames per revolution (heading advances 2 pi / ring_lap_frames per frame)
    ring_yaw_jitter_deg: float = 10.0


CONFIGS = {
    # BASELINE.json configs[0..2]; cfg1 uses 3 panels of 4 (12 landmarks) — see DESIGN.md
    "cfg1": SceneConfig(rows=480, cols=640, f=450.0, grid=(2, 2), n_panels=3, col_spacing=0.9, row_spacing=0.7),
    # The benchmark scenes cfg2 / cfg3 must fuse EXACTLY M corrections in every frame ("the generator must be tuned
    # (and the count asserted) or the EKF silently sees fewer than M updates").  With the detector at its 3.2.0 defaults that is
    # unreachable, measured on the oracle over whole laps: cfg2 fuses 20 of 20 in only 183 of 400 frames (mean
    # 19.2), cfg3 50 of 50 in 66 of 320.  Two DetectorParameters fields are therefore set for these scenes (everything else, and
    # every other test, stays at the defaults):
    #  * polygonalApproxAccuracyRate 0.03 (the OpenCV >= 3.3 default): with 0.05 the INNER contour of the black border of
    #    ~8 % of the DICT_ARUCO_ORIGINAL ids approximates to a quad that is longer than the outer contour, and
    #    _filterTooCloseCandidates then keeps it instead of the marker (13 % of the cfg2 frames and 29 % of the cfg3 frames
    #    find fewer markers than were rendered; cfg3 needs 1000 of the 1023 usable ids, so the ids cannot be hand-picked);
    #  * doCornerRefinement (the "corner sub-pixel refine" stage of BASELINE.json's north_star): with integer corners 1..11 %
    #    of the observations (near .. far end of a panel approach) exceed the reference's covariance gate ||R||_F <= 1
    #    (aruco_slam.cpp:367 with R_x = R_y = 100, parameters.yaml:6-7), whatever the poses: e ~ err^2 f / s^2.
    # What is left (a white bar inside a marker read as id 1023 in ~1.5 % of the frames) depends on the pixel noise; bench.py
    # re-renders such a frame with another noise seed.
    # step 0.025 m per frame: at 0.02 m two consecutive observations of one marker can differ by < 0.01 (estimation noise against
    # the robot's advance) and take the reference's "stationary" no-op branch (aruco_slam.cpp:192-198)
    "cfg2": SceneConfig(step=0.025, detector={"polygonalApproxAccuracyRate": 0.03, "doCornerRefinement": 1}),
    "cfg3": SceneConfig(rows=1080, cols=1920, f=1000.0, grid=(10, 5), n_panels=20, col_spacing=0.36, row_spacing=0.34,
                        tz_far=2.3, tz_near=2.0, max_yaw_deg=20.0, seed=2, step=0.025,
                        detector={"polygonalApproxAccuracyRate": 0.03, "doCornerRefinement": 1}),
    # the headline sizes (1280x720, 20 markers in view, 200 landmarks) on a world whose visible set SLIDES: 50 columns x 4 rows
    # of markers on a ring around the robot, rows staggered, so that one marker leaves and another enters every 2.5 frames
    "cfg2_sliding": SceneConfig(kind="ring", grid=(5, 4), n_panels=10, detector={"polygonalApproxAccuracyRate": 0.03, "doCornerRefinement": 1}),
}


def apply_detector(cfg, ctx=None, oracle=None):
    """install cfg.detector on a capi.Context and / or an oracle (module pyoracle or a pyoracle.Slam)"""
    if cfg.detector:
        if ctx is not None:
            ctx.set_detector_params(**cfg.detector)
        if oracle is not None:
            oracle.set_detector_params(**cfg.detector)


class PanelWorld:
    """Closed polygonal tour past `n_panels` panels; one lap = n_panels * frames_per_panel frames."""

    def __init__(self, cfg: SceneConfig):
        self.cfg = cfg
        gc, gr = cfg.grid
        self.M = gc * gr
        self.L = self.M * cfg.n_panels
        self.K = camera_matrix(cfg.rows, cfg.cols, cfg.f)
        self.frames_per_panel = int(math.floor((cfg.tz_far - cfg.tz_near) / cfg.step + 1e-9)) + 1
        rng = np.random.RandomState(cfg.seed)
        self.ids = np.array(unambiguous_ids(self.L), np.int32)
        # per landmark: lateral offset (left +), height (up +), yaw
        cols_off = (np.arange(gc) - (gc - 1) / 2.0) * cfg.col_spacing
        rows_off = ((gr - 1) / 2.0 - np.arange(gr)) * cfg.row_spacing
        self.lat = np.tile(np.repeat(cols_off[None, :], gr, axis=0).reshape(-1), cfg.n_panels) * -1.0   # image left->right = robot left->right
        self.height = np.tile(np.repeat(rows_off[:, None], gc, axis=1).reshape(-1), cfg.n_panels)
        self.psi = np.deg2rad(rng.uniform(-cfg.max_yaw_deg, cfg.max_yaw_deg, self.L))
        # polygon vertices / headings and landmark world poses
        dphi = 2 * math.pi / cfg.n_panels
        self.vertex = np.zeros((cfg.n_panels + 1, 2))
        self.heading = np.arange(cfg.n_panels + 1) * dphi
        side = (self.frames_per_panel - 1) * cfg.step
        for k in range(cfg.n_panels):
            u = np.array([math.cos(self.heading[k]), math.sin(self.heading[k])])
            self.vertex[k + 1] = self.vertex[k] + side * u
        self.world = np.zeros((self.L, 3))
        for k in range(cfg.n_panels):
            phi = self.heading[k]
            u = np.array([math.cos(phi), math.sin(phi)])
            n = np.array([-math.sin(phi), math.cos(phi)])
            d0 = cfg.tz_far + cfg.r2c[0]
            for i in range(self.M):
                li = k * self.M + i
                p = self.vertex[k] + d0 * u + (self.lat[li] + cfg.r2c[1]) * n
                self.world[li] = (p[0], p[1], norm_angle(norm_angle(phi + math.pi - self.psi[li])))

    def lap_length(self):
        return self.cfg.n_panels * self.frames_per_panel

    def frame(self, index):
        """frame `index` of an endless sequence of laps"""
        cfg = self.cfg
        fpp = self.frames_per_panel
        lap_pos = index % self.lap_length()
        k, j = divmod(lap_pos, fpp)
        # the tour is a closed regular polygon (equal sides, equal turns), so every lap sees identical geometry
        phi = self.heading[k]
        u = np.array([math.cos(phi), math.sin(phi)])
        p = self.vertex[k] + j * cfg.step * u
        if index == 0:
            wl = wr = 0.0                                    # first sample only arms the filter (aruco_slam.cpp:24-29)
        elif j == 0:
            dphi = 2 * math.pi / cfg.n_panels                # turn in place: s_r = -s_l = b * dphi
            wr = cfg.b * dphi / (cfg.kr * cfg.dt)
            wl = -cfg.b * dphi / (cfg.kl * cfg.dt)
        else:
            wl = cfg.step / (cfg.kl * cfg.dt)
            wr = cfg.step / (cfg.kr * cfg.dt)
        sel = np.arange(k * self.M, (k + 1) * self.M)
        poses = np.zeros((self.M, 12))
        n = np.array([-math.sin(phi), math.cos(phi)])
        for a, li in enumerate(sel):
            d = self.world[li, :2] - p
            zx, zy = float(d @ u), float(d @ n)
            t = (-(zy - cfg.r2c[1]), -self.height[li], zx - cfg.r2c[0])
            R, t = marker_pose(t, self.psi[li])
            poses[a, :9] = R.reshape(-1)
            poses[a, 9:] = t
        theta = norm_angle(phi) if phi < 2 * math.pi else 0.0
        return Frame(ids=self.ids[sel].copy(), poses=poses, wl=wl, wr=wr, dt=cfg.dt, true_pose=(p[0], p[1], theta),
                     landmark_index=sel)


class RingWorld:
    """L = n_cols x n_rows markers on a ring of radius `ring_radius` that face its centre; the robot drives a small circle
    around the centre with CONSTANT wheel speeds ("constant wheel speeds ... camera pose advanced with the same
    midpoint model"), camera looking along its heading.  A marker is in view while its ring angle is within half a window
    (grid[0] column spacings wide) of the heading; the rows are staggered by 1 / n_rows of a column spacing, so every frame sees
    exactly M = grid[0] * grid[1] markers and the visible SET changes by one marker every (column spacing / n_rows) of
    heading: the sliding-visibility stream the windowed EKF must not depend on being absent."""

    def __init__(self, cfg: SceneConfig):
        self.cfg = cfg
        gc, gr = cfg.grid
        self.M = gc * gr
        self.L = self.M * cfg.n_panels
        self.n_cols, self.n_rows = self.L // gr, gr
        self.K = camera_matrix(cfg.rows, cfg.cols, cfg.f)
        rng = np.random.RandomState(cfg.seed + 77)
        self.ids = np.array(unambiguous_ids(self.L), np.int32)
        self.dalpha = 2 * math.pi / self.n_cols
        self.half_window = 0.5 * gc * self.dalpha
        col = np.arange(self.L) // gr
        row = np.arange(self.L) % gr
        # + 0.37 of a stagger step: no marker ever sits exactly on the window's edge at a frame's heading
        self.alpha = (col + (row + 0.37) / gr) * self.dalpha
        self.height = ((gr - 1) / 2.0 - row) * cfg.row_spacing
        jitter = np.deg2rad(rng.uniform(-cfg.ring_yaw_jitter_deg, cfg.ring_yaw_jitter_deg, self.L))
        self.theta_w = np.array([norm_angle(norm_angle(a + math.pi + j)) for a, j in zip(self.alpha, jitter)])
        self.pos = cfg.ring_radius * np.stack([np.cos(self.alpha), np.sin(self.alpha)], 1)
        self.world = np.concatenate([self.pos, self.theta_w[:, None]], 1)
        # the robot's trajectory = the reference's own motion model (aruco_slam.cpp:35-52) integrated with constant wheel speeds
        self.dtheta = 2 * math.pi / cfg.ring_lap_frames
        ds = cfg.ring_robot_radius * self.dtheta
        ds_r, ds_l = ds + cfg.b * self.dtheta, ds - cfg.b * self.dtheta
        self.wr, self.wl = ds_r / (cfg.kr * cfg.dt), ds_l / (cfg.kl * cfg.dt)
        pose = np.zeros((cfg.ring_lap_frames, 3))
        x, y, th = cfg.ring_robot_radius * math.sin(0.0), -cfg.ring_robot_radius * math.cos(0.0), 0.0
        for i in range(cfg.ring_lap_frames):
            pose[i] = (x, y, th)
            x += ds * math.cos(th + 0.5 * self.dtheta)
            y += ds * math.sin(th + 0.5 * self.dtheta)
            th = (i + 1) * self.dtheta
        self.pose = pose

    def lap_length(self):
        return self.cfg.ring_lap_frames

    def frame(self, index):
        cfg = self.cfg
        i = index % self.lap_length()
        px, py, phi = self.pose[i]
        u = np.array([math.cos(phi), math.sin(phi)])
        n = np.array([-math.sin(phi), math.cos(phi)])
        rel = (self.alpha - phi + math.pi) % (2 * math.pi) - math.pi
        sel = np.nonzero(np.abs(rel) < self.half_window)[0]
        poses = np.zeros((len(sel), 12))
        for a, li in enumerate(sel):
            d = self.pos[li] - np.array([px, py])
            zx, zy = float(d @ u), float(d @ n)
            t = (-(zy - cfg.r2c[1]), -self.height[li], zx - cfg.r2c[0])
            psi = (phi + math.pi - self.theta_w[li] + math.pi) % (2 * math.pi) - math.pi       # observed at theta = pi - psi
            R, t = marker_pose(t, psi)
            poses[a, :9] = R.reshape(-1)
            poses[a, 9:] = t
        wl, wr = (0.0, 0.0) if index == 0 else (self.wl, self.wr)    # the first sample only arms the filter (aruco_slam.cpp:24-29)
        return Frame(ids=self.ids[sel].copy(), poses=poses, wl=wl, wr=wr, dt=cfg.dt, true_pose=(px, py, norm_angle(phi)),
                     landmark_index=sel)


def make_world(cfg: SceneConfig):
    return RingWorld(cfg) if cfg.kind == "ring" else PanelWorld(cfg)


def simple_scene(rows, cols, f, n_markers, seed=0, tz=(1.2, 2.6), marker_length=0.27, max_yaw_deg=30.0):
    """n markers scattered on a jittered grid in front of the camera (for detector/pose tests)."""
    rng = np.random.RandomState(seed)
    gc = int(math.ceil(math.sqrt(n_markers * cols / rows)))
    gr = int(math.ceil(n_markers / gc))
    ids = unambiguous_ids(max(n_markers, 1), start=1 + 7 * seed % 300)
    poses = np.zeros((n_markers, 12))
    cellw, cellh = cols / gc, rows / gr
    for i in range(n_markers):
        r, c = divmod(i, gc)
        z = rng.uniform(*tz)
        size_px = marker_length * 1.45 * f / z           # marker + quiet zone, with slack for yaw
        z = max(z, marker_length * 1.45 * f / (0.9 * min(cellw, cellh)))
        size_px = marker_length * 1.45 * f / z
        u = (c + 0.5) * cellw + rng.uniform(-1, 1) * max(0.0, (cellw - size_px) / 2 - 4)
        v = (r + 0.5) * cellh + rng.uniform(-1, 1) * max(0.0, (cellh - size_px) / 2 - 4)
        t = ((u - (cols - 1) / 2.0) * z / f, (v - (rows - 1) / 2.0) * z / f, z)
        R, t = marker_pose(t, math.radians(rng.uniform(-max_yaw_deg, max_yaw_deg)))
        # small roll / pitch so quads are general
        ax, az = math.radians(rng.uniform(-8, 8)), math.radians(rng.uniform(-10, 10))
        Rx = np.array([[1, 0, 0], [0, math.cos(ax), -math.sin(ax)], [0, math.sin(ax), math.cos(ax)]])
        Rz = np.array([[math.cos(az), -math.sin(az), 0], [math.sin(az), math.cos(az), 0], [0, 0, 1]])
        R = Rz @ Rx @ R
        poses[i, :9] = R.reshape(-1)
        poses[i, 9:] = t
    return np.array(ids[:n_markers], np.int32), poses, camera_matrix(rows, cols, f)


def random_dictionary(marker_size, count, min_distance, seed=0):
    """bits (count x ms x ms, 1 = white) of a random dictionary whose markers keep a Hamming distance >= min_distance to
    every rotation of every other marker and to their own rotations (how cv::aruco::generateCustomDictionary selects codes);
    maxCorrectionBits = (min_distance - 1) // 2."""
    rng = np.random.RandomState(seed)
    chosen, rots = [], []
    while len(chosen) < count:
        b = rng.randint(0, 2, (marker_size, marker_size)).astype(np.uint8)
        r = [np.rot90(b, k) for k in range(4)]
        if any((r[0] != r[k]).sum() < min_distance for k in range(1, 4)):
            continue
        if any((b != q).sum() < min_distance for q in rots):
            continue
        chosen.append(b)
        rots.extend(r)
    return np.stack(chosen), (min_distance - 1) // 2


def opencv_bytes_list(bits):
    """cv::aruco::Dictionary::getByteListFromBits for every marker: n x nbytes x 4 (rotation channels) uint8"""
    n, ms, _ = bits.shape
    nbytes = (ms * ms + 7) // 8
    out = np.zeros((n, nbytes, 4), np.uint8)
    for i in range(n):
        for r in range(4):
            # OpenCV's rotations index the source as (col, n-1-row), (n-1-row, n-1-col), (n-1-col, row): successive rot90(k=1)
            flat = np.rot90(bits[i], r).reshape(-1)
            cur, byte = 0, 0
            for bit in flat:
                out[i, byte, r] = ((int(out[i, byte, r]) << 1) | int(bit)) & 0xFF
                cur += 1
                if cur == 8:
                    cur, byte = 0, byte + 1
    return out
