"""Windowed EKF (aruco_slam_amd/csrc/ekf_window.hip): runs of frames whose fused landmarks fit into one set S are processed on the
S x S block (chain kernel: the reference's sequential rank-3 corrections on the matrix cores), the rest of Sigma follows once per
window through Lambda / Psi (scan + flush kernels).  Frames of a window may fuse any subset of S and drop "stationary" observations.
Observation sequences are injected (no detector) and every batch is compared with the numpy LITERAL transcription of the
reference (oracle/ekf_literal.py: dense matrices exactly as aruco_slam.cpp:21-74, 88-263 forms them) and with the per-frame
device chain (ASLAM_NO_WINDOWS).  Runs on the emulation build here and on the real library on the MI355X box."""
import math
import os

import numpy as np
import pytest

from aruco_slam_amd import capi
from oracle.ekf_literal import LiteralSlam, project
from oracle.make_golden import rvec_from_R

K = np.array([[450.0, 0, 319.5], [0, 450.0, 239.5], [0, 0, 1]])
D = np.zeros(5)


def make_case(seed, groups, n_land):
    """groups: list of (n_frames, landmark subset); frames of a group all see the same landmarks (windows), the pose of every
    marker changes from frame to frame (no "stationary" branch) unless a subset entry is negative = 'repeat last pose'."""
    rng = np.random.RandomState(seed)
    ids_all = rng.permutation(np.arange(1, 400))[:n_land]
    s = LiteralSlam(r2c=(0.1, -0.05))
    s.K, s.D = K, D
    hl = 0.135
    obj = [(-hl, hl, 0), (hl, hl, 0), (hl, -hl, 0), (-hl, -hl, 0)]
    frames, exp = [], []
    t = 0.0
    f = 0
    for n_frames, subset, repeat in groups:
        for k in range(n_frames):
            t += 0.05
            wl, wr = rng.uniform(1, 4), rng.uniform(1, 4)
            ids, corners, rvs, tvs = [], [], [], []
            for li in subset:
                rs = np.random.RandomState(1000 * seed + 31 * int(li) + (f - 1 if (repeat and k > 0) else f))
                psi = rs.uniform(-0.5, 0.5)
                c_, s_ = math.cos(psi), math.sin(psi)
                R = np.array([[c_, 0, -s_], [0, -1, 0], [-s_, 0, -c_]])
                tz = rs.uniform(0.8, 2.2)
                tv = np.array([rs.uniform(-0.3, 0.3) * tz, rs.uniform(-0.2, 0.2) * tz, tz])
                rv = rvec_from_R(R)
                pr = project(obj, rv, tv, K, D)
                cr = (pr + rs.uniform(-0.05, 0.05, pr.shape)).astype(np.float32)
                ids.append(int(ids_all[li])); corners.append(cr); rvs.append(rv); tvs.append(tv)
            s.add_encoder(wl, wr, t)
            obs = [s.make_observation(ids[i], corners[i], rvs[i], tvs[i]) for i in range(len(ids))]
            s.add_poses(ids, corners, rvs, tvs)
            frames.append(dict(wl=wl, wr=wr, t=t, ids=np.array(ids, np.int32), obs=obs))
            exp.append(dict(mu=s.mu.copy(), sigma=s.sigma.copy(), log=np.array(s.log, np.int32).reshape(-1, 3)))
            f += 1
    return frames, exp


def run_device(frames, exp, batch, windows=True, check=True, max_landmarks=40, max_updates=24):
    if not windows:
        os.environ["ASLAM_NO_WINDOWS"] = "1"
    try:
        nfr = len(frames)
        ctx = capi.Context(max_rows=64, max_cols=64, max_batch=nfr, persistent_waves=4, max_landmarks=max_landmarks, r2c_t=(0.1, -0.05, 0.0),
                           max_updates_per_frame=max_updates)
    finally:
        os.environ.pop("ASLAM_NO_WINDOWS", None)
    ctx.set_camera(K, D)
    t = [fr["t"] for fr in frames]
    ctx.stage_encoders([fr["wl"] for fr in frames], [fr["wr"] for fr in frames], [0.0] + [t[i] - t[i - 1] for i in range(1, nfr)])
    for f, fr in enumerate(frames):
        obs = fr["obs"]
        ctx.inject_observations(f, fr["ids"], [0 if o is None else 1 for o in obs],
                                np.array([np.zeros(3) if o is None else o["z"] for o in obs]).reshape(-1, 3),
                                np.array([np.ones(3) if o is None else np.diag(o["R"]) for o in obs]).reshape(-1, 3))
    ctx.profile_enable(True)
    ctx.profile_reset()
    worst = 0.0
    for f0 in range(0, nfr, batch):
        nb = min(batch, nfr - f0)
        ctx.run_staged(f0, nb, with_ekf=2)
        ctx.sync()
        f = f0 + nb - 1
        mu, S = ctx.get_state()
        if check:
            ids, idx, act, xyth, R = ctx.get_observations()
            assert np.array_equal(np.stack([ids, idx, act], 1).reshape(-1, 3), exp[f]["log"]), f"frame {f}: pop order / branches differ"
            assert mu.shape == exp[f]["mu"].shape
            assert np.allclose(mu, exp[f]["mu"], rtol=1e-9, atol=1e-11), f"frame {f}: mu differs by {np.abs(mu - exp[f]['mu']).max()}"
            e = np.abs(S - exp[f]["sigma"]).max() / np.abs(S).max()
            worst = max(worst, e)
            assert e <= 1e-9, f"frame {f}: sigma differs by {e} (relative)"
            st = ctx.get_slot_ekf_stats(f0, nb)
            for i in range(nb):
                lg = exp[f0 + i]["log"]
                assert st[i, 1] == int((lg[:, 2] == 0).sum()) and st[i, 2] == int((lg[:, 2] == 1).sum()) and st[i, 3] == int((lg[:, 2] == 2).sum())
    prof = ctx.profile_get()
    return ctx.get_state(), prof, worst


CASES = {
    # one long run on the same 6 landmarks; a second group of other landmarks; back to a mix (pose rows in the middle of the state)
    "two_groups": (1, [(9, [0, 1, 2, 3, 4, 5], False), (8, [6, 7, 8], False), (7, [1, 4, 7, 9, 10], False)], 11),
    # 20 landmarks per frame, two sets whose union (25) needs the 128-wide image
    "full_width": (2, [(6, list(range(20)), False), (5, list(range(5, 25)), False)], 25),
    # a repeated pose inside a run: the "stationary" no-op drops out of its frame, the window goes on
    "stationary_inside": (3, [(4, [0, 1, 2, 3], False), (3, [0, 1, 2, 3], True), (5, [0, 1, 2, 3], False)], 4),
    # a single landmark (s = 6) and a long window (more frames than one window holds: 64)
    "one_landmark_long": (4, [(3, [0, 1, 2], False), (70, [1], False)], 3),
    # frames that see SUBSETS of the window's set (a marker lost to the detector or to a gate), in changing combinations
    "subset_frames": (5, [(3, list(range(8)), False), (3, [1, 3, 5], False), (2, [0, 2, 4, 6, 7], False), (1, [7], False), (3, list(range(8)), False)], 8),
    # a visible set that slides by one landmark per frame: the union of a window grows past 20 landmarks (128-wide image)
    "sliding_set": (6, [(1, list(range(12)), False), (1, list(range(12, 24)), False), (1, list(range(24, 30)), False)]
                    + [(1, list(range(i, i + 12)), False) for i in range(0, 19)], 30),
    # window -> window without anything in between: a long window is closed rather than widened, and the next one starts from P / mu_S
    # derived from the first one's Lambda / Psi / psi while its flush is still running (overlapping sets: entries inside and outside S)
    "window_to_window": (9, [(1, list(range(12)), False), (1, list(range(12, 24)), False), (1, list(range(24, 30)), False),
                             (18, list(range(20)), False), (17, list(range(10, 30)), False), (17, list(range(5, 25)), False)], 30),
    # a frame without any observation inside a run (predict only)
    "empty_frame": (7, [(3, [0, 1, 2], False), (2, [], False), (3, [0, 2], False)], 3),
}
# 50 corrections per frame on 55 landmarks (BASELINE config 3's width): the 192-wide image, 6 worker waves
WIDE = (8, [(3, list(range(50)), False), (3, list(range(5, 55)), False)], 55)


@pytest.mark.parametrize("name", sorted(CASES))
def test_windows_match_the_literal_transcription(name):
    seed, groups, n_land = CASES[name]
    frames, exp = make_case(seed, groups, n_land)
    (mu, S), prof, worst = run_device(frames, exp, batch=len(frames))
    assert prof["k_ekf_win_step"][0] > 0, "no window was formed"
    # the same frames on the per-frame chain
    (mu2, S2), prof2, _ = run_device(frames, exp, batch=len(frames), windows=False)
    assert prof2["k_ekf_win_step"][0] == 0
    assert np.allclose(mu, mu2, rtol=1e-10, atol=1e-12) and np.abs(S - S2).max() <= 1e-10 * np.abs(S).max()


@pytest.mark.parametrize("piece", [2, 3])
def test_scan_continuation_pieces_read_the_previous_piece(piece, monkeypatch):
    """A run is cut into chain pieces; the scan of a continuation piece must read the accumulators of the PREVIOUS piece whatever
    order its 16 workgroups run in.  (They alternate between two sets: with one set a workgroup that finished early overwrote
    what a late one still had to read - a matter of timing on the device, deterministic on the emulation with short pieces.)
    ASLAM_WIN_PIECE shortens the pieces; the second run drops landmarks of the first."""
    monkeypatch.setenv("ASLAM_WIN_PIECE", str(piece))
    for groups in ([(6, list(range(10)), False), (5, list(range(5, 15)), False)],
                   [(6, list(range(20)), False), (5, list(range(5, 20)), False)]):
        frames, exp = make_case(2, groups, 25)
        run_device(frames, exp, batch=len(frames))


def test_wide_window_50_corrections_per_frame():
    seed, groups, n_land = WIDE
    frames, exp = make_case(seed, groups, n_land)
    (mu, S), prof, worst = run_device(frames, exp, batch=len(frames), max_landmarks=60, max_updates=64)
    assert prof["k_ekf_win_step"][0] > 0, "no window was formed"


def test_early_start_equals_waiting_for_the_flush(monkeypatch):
    """ASLAM_WIN_NO_EARLY makes every window wait for its own flush (the classic order): same results to rounding"""
    seed, groups, n_land = CASES["window_to_window"]
    frames, exp = make_case(seed, groups, n_land)
    (mu, S), prof, _ = run_device(frames, exp, batch=len(frames))
    assert prof["k_ekf_win_next"][0] >= 2, "no window started early"
    monkeypatch.setenv("ASLAM_WIN_NO_EARLY", "1")
    (mu2, S2), prof2, _ = run_device(frames, exp, batch=len(frames))
    assert prof2["k_ekf_win_next"][0] == 0
    assert np.allclose(mu, mu2, rtol=1e-10, atol=1e-12) and np.abs(S - S2).max() <= 1e-10 * np.abs(S).max()


def test_windows_really_cover_subsets_and_sliding_sets():
    """the planner must keep such frames INSIDE windows (not fall back to the per-frame chain)"""
    for name in ("subset_frames", "sliding_set", "stationary_inside", "empty_frame"):
        seed, groups, n_land = CASES[name]
        frames, exp = make_case(seed, groups, n_land)
        nfr = len(frames)
        ctx = capi.Context(max_rows=64, max_cols=64, max_batch=nfr, persistent_waves=4, max_landmarks=40, r2c_t=(0.1, -0.05, 0.0))
        ctx.set_camera(K, D)
        t = [fr["t"] for fr in frames]
        ctx.stage_encoders([fr["wl"] for fr in frames], [fr["wr"] for fr in frames], [0.0] + [t[i] - t[i - 1] for i in range(1, nfr)])
        for f, fr in enumerate(frames):
            obs = fr["obs"]
            ctx.inject_observations(f, fr["ids"], [0 if o is None else 1 for o in obs],
                                    np.array([np.zeros(3) if o is None else o["z"] for o in obs]).reshape(-1, 3),
                                    np.array([np.ones(3) if o is None else np.diag(o["R"]) for o in obs]).reshape(-1, 3))
        ctx.profile_reset()
        ctx.run_staged(0, nfr, with_ekf=2)
        ctx.sync()
        ps = ctx.plan_stats()
        n_aug = sum(1 for e in exp if (e["log"][:, 2] == 0).any()) if name != "empty_frame" else 1
        assert ps["frames_in_windows"] >= nfr - n_aug - 2, (name, ps, n_aug)


@pytest.mark.parametrize("batch", [1, 2, 5])
def test_windows_with_small_batches(batch):
    """windows never span calls: any batching gives the reference's result after every call"""
    seed, groups, n_land = CASES["two_groups"]
    frames, exp = make_case(seed, groups, n_land)
    run_device(frames, exp, batch=batch)


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(CASES))
def test_windows_on_gpu(name):
    seed, groups, n_land = CASES[name]
    frames, exp = make_case(seed, groups, n_land)
    (mu, S), prof, worst = run_device(frames, exp, batch=len(frames))
    assert prof["k_ekf_win_step"][0] > 0
    run_device(frames, exp, batch=3)


@pytest.mark.gpu
def test_wide_window_on_gpu():
    seed, groups, n_land = WIDE
    frames, exp = make_case(seed, groups, n_land)
    (mu, S), prof, worst = run_device(frames, exp, batch=len(frames), max_landmarks=60, max_updates=64)
    assert prof["k_ekf_win_step"][0] > 0
    run_device(frames, exp, batch=4, max_landmarks=60, max_updates=64)


@pytest.mark.gpu
@pytest.mark.parametrize("piece", [1, 2, 3])
def test_chain_pieces_on_gpu(piece, monkeypatch):
    """the piece hand-over (P image, Lambda blocks, psi, Psi tiles carried through global memory between launches) on the device,
    with short pieces and a second window that drops landmarks of the first"""
    monkeypatch.setenv("ASLAM_WIN_PIECE", str(piece))
    for groups in ([(6, list(range(10)), False), (5, list(range(5, 15)), False)],
                   [(6, list(range(20)), False), (5, list(range(5, 20)), False)]):
        frames, exp = make_case(2, groups, 25)
        run_device(frames, exp, batch=len(frames))


def test_restaging_slots_of_a_submitted_batch_does_not_reach_it():
    """aslam_run_staged returns before the batch's EKF work is even enqueued (it is deferred by one call); encoder samples staged
    into the same slots afterwards, before any synchronisation, must not be read by the batch that was submitted before them
    (include/aruco_slam_hip.h, re-staging rule)."""
    seed, groups, n_land = CASES["two_groups"]
    frames, exp = make_case(seed, groups, n_land)
    nfr = len(frames)
    ctx = capi.Context(max_rows=64, max_cols=64, max_batch=nfr, persistent_waves=4, max_landmarks=40, r2c_t=(0.1, -0.05, 0.0))
    ctx.set_camera(K, D)
    t = [fr["t"] for fr in frames]
    ctx.stage_encoders([fr["wl"] for fr in frames], [fr["wr"] for fr in frames], [0.0] + [t[i] - t[i - 1] for i in range(1, nfr)])
    for f, fr in enumerate(frames):
        obs = fr["obs"]
        ctx.inject_observations(f, fr["ids"], [0 if o is None else 1 for o in obs],
                                np.array([np.zeros(3) if o is None else o["z"] for o in obs]).reshape(-1, 3),
                                np.array([np.ones(3) if o is None else np.diag(o["R"]) for o in obs]).reshape(-1, 3))
    ctx.run_staged(0, nfr, with_ekf=2)                                  # asynchronous; EKF work still pending
    ctx.stage_encoders([9.0] * nfr, [-9.0] * nfr, [0.5] * nfr)         # the NEXT batch's samples into the same slots
    ctx.sync()
    mu, S = ctx.get_state()
    assert np.allclose(mu, exp[-1]["mu"], rtol=1e-9, atol=1e-11), f"mu differs by {np.abs(mu - exp[-1]['mu']).max()}"
    assert np.abs(S - exp[-1]["sigma"]).max() <= 1e-9 * np.abs(S).max()
