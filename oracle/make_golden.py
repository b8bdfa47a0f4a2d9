"""ORACLE — TEST INFRASTRUCTURE ONLY.  Generates tests/golden/ekf_literal_*.npz with the numpy literal
transcription (oracle/ekf_literal.py): seeded inputs (encoder samples, per-frame marker ids / corners / poses)
and the expected mu, Sigma, pop order and actions after every frame.  Run: python oracle/make_golden.py"""
import math
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.ekf_literal import LiteralSlam, project, rodrigues  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def rvec_from_R(R):
    th = math.acos(max(-1.0, min(1.0, (np.trace(R) - 1) / 2)))
    v = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    return v / (2 * math.sin(th)) * th


def make_case(seed, n_landmarks, n_frames, per_frame, r2c, D, dup_frame=None):
    rng = np.random.RandomState(seed)
    K = np.array([[450.0, 0, 319.5], [0, 450.0, 239.5], [0, 0, 1]])
    s = LiteralSlam(r2c=r2c)
    s.K, s.D = K, np.asarray(D, float)
    ids_all = rng.permutation(np.arange(1, 200))[:n_landmarks]
    frames = []
    t = 0.0
    hl = 0.135
    obj = [(-hl, hl, 0), (hl, hl, 0), (hl, -hl, 0), (-hl, -hl, 0)]
    exp = []
    for f in range(n_frames):
        t += 0.1
        wl, wr = rng.uniform(1, 6), rng.uniform(1, 6)
        vis = rng.permutation(n_landmarks)[:per_frame]
        if f % 5 == 4:
            vis = frames[-1]["vis"]                     # same markers again -> exercises the "stationary" branch
        ids, corners, rvs, tvs = [], [], [], []
        for li in vis:
            rs = np.random.RandomState(1000 * seed + 17 * int(li) + (f - 1 if f % 5 == 4 else f))
            psi = rs.uniform(-0.5, 0.5)
            c_, s_ = math.cos(psi), math.sin(psi)
            R = np.array([[c_, 0, -s_], [0, -1, 0], [-s_, 0, -c_]])
            tz = rs.uniform(0.7, 2.2) if rs.rand() < 0.9 else rs.uniform(2.9, 3.3)      # a few beyond the 3 m range gate
            tv = np.array([rs.uniform(-0.3, 0.3) * tz, rs.uniform(-0.2, 0.2) * tz, tz])
            rv = rvec_from_R(R)
            pr = project(obj, rv, tv, K, D)
            cr = np.round(pr + rs.uniform(-0.15, 0.15, pr.shape)).astype(np.float32)      # integer corners (Q15)
            ids.append(int(ids_all[li])); corners.append(cr); rvs.append(rv); tvs.append(tv)
        if dup_frame is not None and f == dup_frame and ids:
            ids.append(ids[0]); corners.append(corners[0]); rvs.append(rvs[0]); tvs.append(tvs[0])   # duplicate id (Q10)
        s.add_encoder(wl, wr, t)
        s.add_poses(ids, corners, rvs, tvs)
        frames.append(dict(vis=vis, wl=wl, wr=wr, t=t, ids=np.array(ids, np.int32), corners=np.array(corners, np.float32).reshape(-1, 4, 2),
                           rvecs=np.array(rvs).reshape(-1, 3), tvecs=np.array(tvs).reshape(-1, 3)))
        exp.append(dict(mu=s.mu.copy(), sigma=s.sigma.copy(), log=np.array(s.log, np.int32).reshape(-1, 3)))
    return K, frames, exp


def main():
    os.makedirs(OUT, exist_ok=True)
    cases = [dict(seed=1, n_landmarks=6, n_frames=12, per_frame=3, r2c=(0.18, -0.1), D=[0, 0, 0, 0, 0]),
             dict(seed=2, n_landmarks=10, n_frames=16, per_frame=5, r2c=(0.0, 0.0), D=[0.0416, -0.0477, -0.00326, -0.00399, 0.0111], dup_frame=1),
             dict(seed=3, n_landmarks=20, n_frames=10, per_frame=20, r2c=(0.05, 0.02), D=[0, 0, 0, 0, 0])]
    for i, c in enumerate(cases):
        K, frames, exp = make_case(**c)
        d = dict(K=K, D=np.asarray(c["D"], float), r2c=np.asarray(c["r2c"], float), n_frames=len(frames))
        for f, (fr, ex) in enumerate(zip(frames, exp)):
            for k in ("wl", "wr", "t", "ids", "corners", "rvecs", "tvecs"):
                d[f"in{f}_{k}"] = fr[k]
            for k in ("mu", "sigma", "log"):
                d[f"out{f}_{k}"] = ex[k]
        path = os.path.join(OUT, f"ekf_literal_{i + 1}.npz")
        np.savez_compressed(path, **d)
        print(path, "frames", len(frames), "final N", exp[-1]["mu"].size, "actions", np.bincount(np.concatenate([e["log"][:, 2] for e in exp]), minlength=3))


if __name__ == "__main__":
    main()
