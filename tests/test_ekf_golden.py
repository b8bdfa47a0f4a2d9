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
    g = np.load(path)
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
