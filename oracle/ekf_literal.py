"""ORACLE — TEST INFRASTRUCTURE ONLY.  Second, independently written restatement of the reference's EKF-SLAM
arithmetic in numpy: a line-by-line transcription of /root/reference/src/aruco_slam.cpp:21-74 (addEncoder),
:88-263 (addImage queue loop), :325-374 (observation assembly), :412-435, :437-471, with dense matrices exactly
as the reference forms them (F, Gx = Gxm*F, (I - K*Gx)*sigma, tmp_sigma growth).  It exists to pin the C++
oracle (oracle/ekf.cpp): oracle/make_golden.py runs it on seeded inputs and commits inputs + outputs under
tests/golden/; tests/test_oracle_golden.py replays them through the C++ oracle.
"""
import math

import numpy as np

PI = 3.14159265358979323846


def norm_angle(a):                                  # aruco_slam.cpp:412-421 (wraps once)
    if a >= PI:
        a -= 2.0 * PI
    if a < -PI:
        a += 2.0 * PI
    return a


def rodrigues(r):                                   # cv::Rodrigues, vector -> matrix
    r = np.asarray(r, float)
    th = float(np.linalg.norm(r))
    if th < 2.220446049250313e-16:
        return np.eye(3)
    k = r / th
    Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return math.cos(th) * np.eye(3) + (1 - math.cos(th)) * np.outer(k, k) + math.sin(th) * Kx


def project(obj, rvec, tvec, K, D):                 # cv::projectPoints, plumb_bob
    R = rodrigues(rvec)
    k1, k2, p1, p2, k3 = (list(D) + [0] * 5)[:5]
    out = []
    for X in obj:
        p = R @ np.asarray(X, float) + np.asarray(tvec, float)
        x, y = p[0] / p[2], p[1] / p[2]
        r2 = x * x + y * y
        cd = 1 + k1 * r2 + k2 * r2 * r2 + k3 * r2 * r2 * r2
        xd = x * cd + 2 * p1 * x * y + p2 * (r2 + 2 * x * x)
        yd = y * cd + p1 * (r2 + 2 * y * y) + 2 * p2 * x * y
        out.append((xd * K[0][0] + K[0][2], yd * K[1][1] + K[1][2]))
    return np.array(out)


class _Heap:
    """std::priority_queue<ArucoMarker> of libstdc++ (push_heap / pop_heap), comparator a < b <=> a.index > b.index
    (aruco_slam.h:85-88)."""

    def __init__(self):
        self.c = []

    @staticmethod
    def _less(a, b):
        return a["index"] > b["index"]

    def push(self, v):
        c = self.c
        c.append(v)
        hole = len(c) - 1
        parent = (hole - 1) // 2
        while hole > 0 and self._less(c[parent], v):
            c[hole] = c[parent]
            hole = parent
            parent = (hole - 1) // 2
        c[hole] = v

    def pop(self):
        c = self.c
        top = c[0]
        if len(c) > 1:
            value = c[-1]
            c[-1] = c[0]
            n = len(c) - 1
            hole = 0
            second = 0
            while second < (n - 1) // 2:
                second = 2 * (second + 1)
                if self._less(c[second], c[second - 1]):
                    second -= 1
                c[hole] = c[second]
                hole = second
            if (n & 1) == 0 and second == (n - 2) // 2:
                second = 2 * (second + 1)
                c[hole] = c[second - 1]
                hole = second - 1
            parent = (hole - 1) // 2
            while hole > 0 and self._less(c[parent], value):
                c[hole] = c[parent]
                hole = parent
                parent = (hole - 1) // 2
            c[hole] = value
        c.pop()
        return top


class LiteralSlam:
    def __init__(self, Q_k=0.01, R_x=100.0, R_y=100.0, R_theta=10.0, kl=0.05, kr=0.05, b=0.09, marker_length=0.27,
                 r2c=(0.0, 0.0), useful_distance_threshold=3.0):
        self.Q_k, self.R_x, self.R_y, self.R_theta = Q_k, R_x, R_y, R_theta
        self.kl, self.kr, self.b, self.L = kl, kr, b, marker_length
        self.r2c = r2c
        self.thr = np.float32(useful_distance_threshold)
        self.mu = np.zeros(3)
        self.sigma = np.zeros((3, 3))
        self.is_init = False
        self.last_time = 0.0
        self.id_map = {}
        self.last_observed = []
        self.K = np.eye(3)
        self.D = np.zeros(5)
        self.log = []

    # aruco_slam.cpp:21-74
    def add_encoder(self, wl, wr, t_now):
        if not self.is_init:
            self.last_time = t_now
            self.is_init = True
            return
        dt = t_now - self.last_time
        self.last_time = t_now
        delta_sl = self.kl * (dt * wl)
        delta_sr = self.kr * (dt * wr)
        delta_theta = (delta_sr - delta_sl) / (2 * self.b)
        delta_s = 0.5 * (delta_sr + delta_sl)
        tmp_th = self.mu[2] + 0.5 * delta_theta
        c, s = math.cos(tmp_th), math.sin(tmp_th)
        self.mu[0] += delta_s * c
        self.mu[1] += delta_s * s
        self.mu[2] = norm_angle(self.mu[2] + delta_theta)
        H_xi = np.array([[1.0, 0.0, -delta_s * s], [0.0, 1.0, delta_s * c], [0.0, 0.0, 1.0]])
        wkh = (0.5 * self.kl * dt) * np.array([[c, c], [s, s], [1 / self.b, -1 / self.b]])
        N = self.mu.size
        F = np.zeros((N, 3)); F[:3, :3] = np.eye(3)
        Hx = np.eye(N); Hx[:3, :3] = H_xi
        sigma_u = np.diag([self.Q_k * abs(wl), self.Q_k * abs(wr)])
        Qk = wkh @ sigma_u @ wkh.T
        self.sigma = Hx @ self.sigma @ Hx.T + F @ Qk @ F.T

    # aruco_slam.cpp:325-374 (one detection) + :437-471
    def make_observation(self, marker_id, corners, rvec, tvec):
        tvec = np.asarray(tvec, float)
        dist = np.float32(np.linalg.norm(tvec))
        if dist > self.thr:
            return None
        R = rodrigues(rvec)
        x = tvec[2] + self.r2c[0]
        y = -tvec[0] + self.r2c[1]
        theta = norm_angle(math.atan2(-R[0, 2], R[2, 2]))
        hl = float(np.float32(self.L / 2.0))
        obj = [(-hl, hl, 0), (hl, hl, 0), (hl, -hl, 0), (-hl, -hl, 0)]
        proj = project(obj, rvec, tvec, self.K, self.D).astype(np.float32).astype(float)
        c = np.asarray(corners, np.float32).astype(float).reshape(4, 2)
        total = float(sum(np.hypot(*(c[i] - proj[i])) ** 2 for i in range(4)))
        rms = total / 4.0
        object_error = (rms / float(np.hypot(*(c[0] - c[2])))) * (float(np.linalg.norm(tvec)) / self.L)
        cov = np.diag([object_error * self.R_x + 1e-2, object_error * self.R_y + 1e-2, object_error * self.R_theta + 1e-3])
        if np.linalg.norm(cov) > 1:
            return None
        return dict(id=int(marker_id), index=self.id_map.get(int(marker_id), -1), z=np.array([x, y, theta]), R=cov,
                    last=np.full(3, np.nan))

    # aruco_slam.cpp:76-263 given the detections' poses
    def add_poses(self, ids, corners, rvecs, tvecs):
        if not self.is_init:
            return
        q = _Heap()
        for i in range(len(ids)):
            ob = self.make_observation(ids[i], corners[i], rvecs[i], tvecs[i])
            if ob is not None:
                q.push(ob)
        mu = self.mu.copy()
        observed = []
        self.log = []
        while q.c:
            ob = q.pop()
            Rk = ob["R"]
            if ob["index"] >= 0:
                N = self.mu.size
                i3 = 3 + 3 * ob["index"]
                F = np.zeros((6, N)); F[:3, :3] = np.eye(3); F[3:, i3:i3 + 3] = np.eye(3)
                mx, my, mth = mu[i3], mu[i3 + 1], mu[i3 + 2]
                x, y, th = mu[0], mu[1], mu[2]
                s, c = math.sin(th), math.cos(th)
                gdx, gdy = mx - x, my - y
                gdth = norm_angle(mth - th)
                z_hat = np.array([gdx * c + gdy * s, -gdx * s + gdy * c, gdth])
                z = ob["z"].copy()
                ze = z - z_hat
                ze[2] = norm_angle(ze[2])
                Gxm = np.array([[-c, -s, -gdx * s + gdy * c, c, s, 0],
                                [s, -c, -gdx * c - gdy * s, -s, c, 0],
                                [0, 0, -1, 0, 0, 1]], float)
                Gx = Gxm @ F
                Kg = self.sigma @ Gx.T @ np.linalg.inv(Gx @ self.sigma @ Gx.T + Rk)
                last = next((o for o in self.last_observed if o["id"] == ob["id"]), None)
                if last is not None and np.linalg.norm(last["last"] - z) < 0.01:
                    action = 2                                  # 3x0 block: nothing happens
                else:
                    action = 1
                    ob["last"] = z
                    self.mu = self.mu + Kg @ ze
                    self.sigma = (np.eye(N) - Kg @ Gx) @ self.sigma
            else:
                action = 0
                sinth = float(np.float32(math.sin(mu[2])))
                costh = float(np.float32(math.cos(mu[2])))
                N = self.mu.size
                map_x = mu[0] + costh * ob["z"][0] - sinth * ob["z"][1]
                map_y = mu[1] + sinth * ob["z"][0] + costh * ob["z"][1]
                map_th = norm_angle(mu[2] + ob["z"][2])
                dx, dy = map_x - mu[0], map_y - mu[1]
                sigma_s = self.sigma[:3, :3]
                Gsk = np.array([[-costh, -sinth, -sinth * dx + costh * dy], [sinth, -costh, -dx * costh - dy * sinth], [0, 0, -1]], float)
                Gmi = np.array([[costh, sinth, 0], [-sinth, costh, 0], [0, 0, 1]], float)
                sigma_mm = Gmi @ (Gsk @ sigma_s @ Gsk.T + Rk).T @ Gmi.T
                sigma_mx = -Gmi @ Gsk @ self.sigma[:3, :]
                tmp = np.zeros((N + 3, N + 3))
                tmp[:N, :N] = self.sigma
                tmp[:N, N:] = sigma_mx.T
                tmp[N:, :N] = sigma_mx
                tmp[N:, N:] = sigma_mm
                self.sigma = tmp
                self.mu = np.concatenate([self.mu, [map_x, map_y, map_th]])
                self.id_map.setdefault(ob["id"], (self.mu.size - 3) // 3 - 1)
            observed.append(ob)
            self.log.append((ob["id"], ob["index"], action))
        self.last_observed = observed
