"""What the node publishes, as plain data (toRosPose / detected_map_ / detected_markers_, aruco_slam.cpp:265-281,325-347,
378-410), and state persistence.  Host arithmetic only; checked against independent numpy formulas."""
import math
import numpy as np
import pytest

from aruco_slam_amd import capi, synth
from oracle import pyoracle as orc


def rpy_matrix(r, p, y):
    Rx = np.array([[1, 0, 0], [0, math.cos(r), -math.sin(r)], [0, math.sin(r), math.cos(r)]])
    Ry = np.array([[math.cos(p), 0, math.sin(p)], [0, 1, 0], [-math.sin(p), 0, math.cos(p)]])
    Rz = np.array([[math.cos(y), -math.sin(y), 0], [math.sin(y), math.cos(y), 0], [0, 0, 1]])
    return Rz @ Ry @ Rx


def quat_matrix(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


@pytest.fixture
def ran():
    return make_ran()


def make_ran():
    cfg = synth.CONFIGS["cfg1"]
    w = synth.PanelWorld(cfg)
    q_r2c = np.array([0.5, -0.5, 0.5, -0.5])                      # optical -> base_link style rotation (x, y, z, w)
    ctx = capi.Context(max_rows=cfg.rows, max_cols=cfg.cols, max_batch=4, persistent_waves=4, max_landmarks=16,
                       r2c_t=(0.18, -0.1, 0.05), r2c_q=tuple(q_r2c))
    ctx.set_camera(w.K, np.zeros(5))
    frs = [w.frame(i) for i in range(4)]
    for i, fr in enumerate(frs):
        ctx.synth_render(i, cfg.rows, cfg.cols, w.K, fr.ids, fr.poses, noise_amp=1, seed=i, download=False)
    ctx.stage_encoders([f.wl for f in frs], [f.wr for f in frs], [f.dt for f in frs])
    ctx.run_staged(0, 4, with_ekf=True)
    ctx.sync()
    return ctx, q_r2c


def test_pose_message(ran):
    ctx, _ = ran
    mu, S = ctx.get_state()
    pos, q, cov = ctx.pose_msg()
    assert np.array_equal(pos, [mu[0], mu[1], 0.1])
    assert np.allclose(quat_matrix(q), rpy_matrix(0, 0, mu[2]), atol=1e-15)
    expect = np.zeros((6, 6))
    for i, a in enumerate((0, 1, 5)):
        for j, b in enumerate((0, 1, 5)):
            expect[a, b] = S[i, j]
    assert np.array_equal(cov, expect)


def test_map_markers(ran):
    ctx, _ = ran
    mu, _ = ctx.get_state()
    mk = ctx.map_markers()
    assert len(mk) == (mu.size - 3) // 3 and len(mk) > 0
    for i, m in enumerate(mk):
        assert m["id"] == i and m["scale"] == (0.27, 0.27, 0.01) and m["color"] == (1.0, 0.5, 1.0, 0.5) and m["lifetime"] == 0.0
        assert np.array_equal(m["position"], [mu[3 + 3 * i], mu[4 + 3 * i], 0.3])
        assert np.allclose(quat_matrix(m["orientation"]), rpy_matrix(0, 1.5708, mu[5 + 3 * i]), atol=1e-14)


def test_detected_markers(ran):
    ctx, q_r2c = ran
    ids, corners, rv, tv = ctx.get_detections()
    mk = ctx.detected_markers()
    assert len(mk) == len(ids) > 0                                 # all inside the 3 m gate in this scene
    Rr = quat_matrix(q_r2c)
    for m, i, r, t in zip(mk, ids, rv, tv):
        assert m["id"] == i and m["color"] == (1.0, 0.0, 0.0, 1.0) and m["lifetime"] == 0.1
        assert np.allclose(m["position"], Rr @ t + np.array([0.18, -0.1, 0.05]), atol=1e-13)
        assert np.allclose(quat_matrix(m["orientation"]), Rr @ orc.rodrigues(r)[0], atol=1e-12)


def test_state_round_trip_through_a_file(ran, tmp_path):
    ctx, _ = ran
    f = tmp_path / "state.bin"
    ctx.save_state(f)
    mu, S = ctx.get_state()
    ids = ctx.get_landmark_ids()
    b = capi.Context(max_rows=64, max_cols=64, max_batch=1, persistent_waves=4, max_landmarks=16)
    b.load_state(f)
    mu2, S2 = b.get_state()
    assert np.array_equal(mu, mu2) and np.array_equal(S, S2) and np.array_equal(ids, b.get_landmark_ids())
    small = capi.Context(max_rows=64, max_cols=64, max_batch=1, persistent_waves=4, max_landmarks=2)
    with pytest.raises(capi.AslamError):
        small.load_state(f)                                        # does not fit
    with pytest.raises(capi.AslamError):
        b.load_state(tmp_path / "missing.bin")


def test_map_file_loader_follows_the_reference_rules(tmp_path):
    """MapLoader::loadMap (map_loader.cpp:7-81): comments, blank lines, short lines, optional fields and their crossed fallbacks"""
    ctx = capi.Context(max_rows=64, max_cols=64, max_batch=1, persistent_waves=4, max_landmarks=4)
    f = tmp_path / "map.txt"
    f.write_text("# id length x y z roll pitch yaw\n"
                 "0 0.27 5.1 0 0.3 0 -1.5708 0\n"
                 "\n"
                 "3\t0.27\t4 0.6 0.3 1.5708 -0 0.25\n"
                 "7 0.2 1 2\n"                      # only the mandatory fields
                 "9 0.2 1\n"                        # too short: skipped
                 "11 0.3 -1 -2 0.5 0.7\n")          # roll given, pitch and yaw missing
    mk = ctx.load_map_txt(f)
    assert [m["id"] for m in mk] == [0, 3, 7, 11]
    assert mk[0]["scale"] == (0.27, 0.27, 0.01) and mk[0]["color"] == (1.0, 1.0, 1.0, 0.5) and mk[0]["lifetime"] == 0.0
    assert np.array_equal(mk[1]["position"], [4, 0.6, 0.3])
    assert np.allclose(quat_matrix(mk[0]["orientation"]), rpy_matrix(0, -1.5708, 0), atol=1e-15)
    assert np.allclose(quat_matrix(mk[1]["orientation"]), rpy_matrix(1.5708, 0, 0.25), atol=1e-15)
    # line "7 ...": z -> 0; no roll -> yaw = 0 (roll stays 1.5708 from the previous line until...) no pitch -> 0; no yaw -> roll = 0
    assert np.array_equal(mk[2]["position"], [1, 2, 0]) and np.allclose(quat_matrix(mk[2]["orientation"]), np.eye(3), atol=1e-15)
    # line "11 ...": roll = 0.7 is read, pitch missing -> 0, yaw missing -> ROLL zeroed (the crossed fallback), yaw keeps 0
    assert np.allclose(quat_matrix(mk[3]["orientation"]), np.eye(3), atol=1e-15)
    g = tmp_path / "bad.txt"
    g.write_text("1 0.27 0 0\nx 1 2 3\n2 0.27 1 1\n")
    assert ctx.load_map_txt(g) == []                                 # malformed line: the loader clears the map and stops
    with pytest.raises(capi.AslamError):
        ctx.load_map_txt(tmp_path / "missing.txt")
