"""The device EKF (plan / gather / small / T / update kernels) replayed on the committed golden vectors of the numpy
literal transcription (oracle/ekf_literal.py): observations are formed in Python exactly as the reference does
(aruco_slam.cpp:325-374), injected into the slots, and only the EKF steps run.  Covers what the rendered scenes cannot:
duplicate ids in one frame (Q10), the "stationary" branch (Q2), range/covariance gates, many new landmarks per frame
(libstdc++ heap order, Q9).  Runs on the emulation build without a GPU and on the real library on the MI355X box."""
import glob
import os

import numpy as np
import pytest

from aruco_slam_amd import capi
from oracle.ekf_literal import LiteralSlam

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ekf_literal_*.npz")))


def replay(path, batch, max_updates=24):
    g = np.load(path) if isinstance(path, str) else path
    nfr = int(g["n_frames"])
    r2c = g["r2c"]
    ctx = capi.Context(max_rows=64, max_cols=64, max_batch=nfr, persistent_waves=4, max_landmarks=32, max_updates_per_frame=max_updates,
                       r2c_t=(float(r2c[0]), float(r2c[1]), 0.0))
    ctx.set_camera(g["K"], g["D"])
    # observation assembly (gates, covariance) in Python, independent of the device pose kernel
    s = LiteralSlam(r2c=(float(r2c[0]), float(r2c[1])))
    s.K, s.D = g["K"], g["D"]
    wl = [float(g[f"in{f}_wl"]) for f in range(nfr)]
    wr = [float(g[f"in{f}_wr"]) for f in range(nfr)]
    t = [float(g[f"in{f}_t"]) for f in range(nfr)]
    dt = [0.0] + [t[f] - t[f - 1] for f in range(1, nfr)]
    ctx.stage_encoders(wl, wr, dt)
    for f in range(nfr):
        ids = g[f"in{f}_ids"]
        obs = [s.make_observation(ids[i], g[f"in{f}_corners"][i], g[f"in{f}_rvecs"][i], g[f"in{f}_tvecs"][i]) for i in range(len(ids))]
        valid = [0 if o is None else 1 for o in obs]
        xyth = [np.zeros(3) if o is None else o["z"] for o in obs]
        Rd = [np.ones(3) if o is None else np.diag(o["R"]) for o in obs]
        ctx.inject_observations(f, ids, valid, np.array(xyth).reshape(-1, 3), np.array(Rd).reshape(-1, 3))
    seen = np.zeros(3, int)
    for f0 in range(0, nfr, batch):
        nb = min(batch, nfr - f0)
        ctx.run_staged(f0, nb, with_ekf=2)
        ctx.sync()
        f = f0 + nb - 1
        ids, idx, act, xyth, R = ctx.get_observations()
        assert np.array_equal(np.stack([ids, idx, act], 1).reshape(-1, 3), g[f"out{f}_log"])     # pop order, indices, branch
        mu, S = ctx.get_state()
        assert mu.shape == g[f"out{f}_mu"].shape
        assert np.allclose(mu, g[f"out{f}_mu"], rtol=1e-9, atol=1e-11)
        assert np.abs(S - g[f"out{f}_sigma"]).max() <= 1e-9 * np.abs(S).max()
        seen += np.bincount(act, minlength=3)[:3]
    return seen


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p) for p in GOLDEN])
def test_device_ekf_replays_golden_per_frame(path):
    seen = replay(path, batch=1)
    assert seen[0] > 0 and seen[1] > 0


@pytest.mark.parametrize("path", GOLDEN[:1], ids=[os.path.basename(p) for p in GOLDEN[:1]])
def test_device_ekf_replays_golden_batched(path):
    replay(path, batch=5)


def test_device_ekf_general_chain():
    """max_updates_per_frame > 24 selects the general 5-kernel chain (used by 50-marker frames)"""
    replay(GOLDEN[2], batch=1, max_updates=64)


def test_device_ekf_medium_chain():
    """25 .. 64 updates per frame: register-block Gauss-Jordan with 2 x 2 blocks per thread + the f64 MFMA covariance update"""
    replay(GOLDEN[2], batch=1, max_updates=48)


def test_device_ekf_general_chain_above_64():
    replay(GOLDEN[1], batch=1, max_updates=100)


def test_too_many_updates_is_reported():
    with pytest.raises(capi.AslamError) as e:
        replay(GOLDEN[2], batch=1, max_updates=4)
    assert e.value.code == -4


@pytest.mark.gpu
@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p) for p in GOLDEN])
def test_device_ekf_replays_golden_on_gpu(path):
    replay(path, batch=1)
    replay(path, batch=4)
    replay(path, batch=3, max_updates=64)
    replay(path, batch=2, max_updates=100)


def _fresh_case(seed):
    """cases drawn at test time by the golden generator (oracle/make_golden.py:make_case) with other seeds and shapes: duplicate
    ids, repeated frames, gated markers, distortion; every frame against the numpy literal transcription"""
    from oracle.make_golden import make_case
    rng = np.random.RandomState(seed)
    c = dict(seed=seed, n_landmarks=int(rng.randint(4, 16)), n_frames=int(rng.randint(6, 14)), per_frame=0, r2c=(float(rng.uniform(-0.2, 0.2)), float(rng.uniform(-0.2, 0.2))),
             D=[0, 0, 0, 0, 0] if seed % 2 else [0.05, -0.04, 0.002, -0.003, 0.01], dup_frame=int(rng.randint(0, 5)) if seed % 3 else None)
    c["per_frame"] = int(rng.randint(2, c["n_landmarks"] + 1))
    K, frames, exp = make_case(**c)
    g = dict(K=K, D=np.asarray(c["D"], float), r2c=np.asarray(c["r2c"], float), n_frames=len(frames))
    for f, (fr, ex) in enumerate(zip(frames, exp)):
        for k in ("wl", "wr", "t", "ids", "corners", "rvecs", "tvecs"):
            g[f"in{f}_{k}"] = fr[k]
        for k in ("mu", "sigma", "log"):
            g[f"out{f}_{k}"] = ex[k]
    return g


@pytest.mark.parametrize("seed", [11, 12, 13, 14])
def test_device_ekf_on_fresh_random_cases(seed):
    assert replay(_fresh_case(seed), batch=1 + seed % 3).sum() > 0


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(21, 33))
def test_device_ekf_on_fresh_random_cases_gpu(seed):
    g = _fresh_case(seed)
    assert replay(g, batch=1 + seed % 3).sum() > 0
    replay(g, batch=2, max_updates=48)
    replay(g, batch=1, max_updates=100)
