"""GPU parity tests (run on a real MI355X): HIP path through the C-ABI vs the CPU oracle."""
import numpy as np
import pytest

from aruco_slam_amd import capi, synth
from oracle import pyoracle as orc
import parity_common as pc

pytestmark = pytest.mark.gpu


def test_native_library_is_the_one_loaded():
    lib = capi.load()
    assert capi.lib_path().endswith("libaruco_slam_hip.so")
    assert lib is not None


@pytest.mark.parametrize("name", ["cfg1", "cfg2", "cfg3"])
def test_detector_stage_parity_full_size(name):
    """threshold / contours / candidates / ids / corners bit-exact at BASELINE.json's frame sizes."""
    cfg = synth.CONFIGS[name]
    w = synth.PanelWorld(cfg)
    B = 4
    ctx = capi.Context(max_rows=cfg.rows, max_cols=cfg.cols, max_batch=B, max_landmarks=16)
    D = np.zeros(5)
    ctx.set_camera(w.K, D)
    synth.apply_detector(cfg, ctx, orc)              # cfg3 runs its own detector profile (aruco_slam_amd/synth.py)
    try:
        frs = [w.frame(i * (w.frames_per_panel // 2 + 1)) for i in range(B)]
        imgs = [ctx.synth_render(i, cfg.rows, cfg.cols, w.K, fr.ids, fr.poses, noise_amp=2, seed=i * (w.frames_per_panel // 2 + 1))
                for i, fr in enumerate(frs)]
        ctx.run_staged(0, B, with_ekf=False)
        ctx.sync()
        for i in range(B):
            ids, corners, rv, tv = pc.check_stages(ctx, i, imgs[i], expect_ids=frs[i].ids)
            pc.check_poses(ids, corners, rv, tv, w.K, D)
    finally:
        orc.set_detector_params()


def test_detector_bgr_input_and_distortion():
    rows, cols, f = 480, 640, 450.0
    ids, poses, K = synth.simple_scene(rows, cols, f, 4, seed=3)
    D = np.array([0.0416, -0.0477, -0.00326, -0.00399, 0.0111])      # plumb_bob of the reference's default.yaml:16-20
    ctx = capi.Context(max_rows=rows, max_cols=cols, max_batch=1, max_landmarks=8)
    ctx.set_camera(K, D)
    gray = ctx.synth_render(0, rows, cols, K, ids, poses, noise_amp=2, seed=1)
    rng = np.random.RandomState(0)
    bgr = np.stack([np.clip(gray.astype(int) + rng.randint(-6, 7, gray.shape), 0, 255)] * 3, axis=-1).astype(np.uint8)
    bgr[..., 1] = gray
    ctx.stage_frames(bgr)
    ctx.run_staged(0, 1, with_ekf=False)
    ctx.sync()
    g = orc.bgr2gray(bgr)
    ids_g, c_g, rv, tv = pc.check_stages(ctx, 0, g)
    ids_o, c_o = orc.detect(bgr)
    assert np.array_equal(ids_o, ids_g) and np.array_equal(c_o, c_g)
    assert len(ids_g) == 4
    pc.check_poses(ids_g, c_g, rv, tv, K, D)


@pytest.mark.parametrize("link", ["lds", "serial"])
def test_contours_on_degenerate_images(link, monkeypatch):
    """random / blocky / thin-line / smooth-blob images, a spiral and a comb (one border of thousands of points through hundreds of
    nodes): every border the sequential Suzuki scan finds, in its order, point for point - through both forms of the cycle
    resolution (k_link in LDS; k_link_serial, which pure noise needs anyway: more nodes than the LDS image holds)"""
    from scipy.ndimage import gaussian_filter
    from test_pipeline_emu import _spiral
    if link == "serial":
        monkeypatch.setenv("ASLAM_LINK_LDS_NODES", "0")
    rows, cols = 240, 320
    ctx = capi.Context(max_rows=rows, max_cols=cols, max_batch=1, max_landmarks=4, cap_contours_per_frame=1 << 15,
                       cap_points_per_frame=1 << 21, cap_starts_per_frame=1 << 17)
    ctx.set_camera(synth.camera_matrix(rows, cols, 200.0), np.zeros(5))
    rng = np.random.RandomState(7)
    for trial in range(21):
        kind = trial % 7
        if kind == 0:
            img = rng.randint(0, 256, (rows, cols)).astype(np.uint8)
        elif kind == 1:
            img = np.kron(rng.randint(0, 256, (rows // 8, cols // 8)), np.ones((8, 8))).astype(np.uint8)
        elif kind == 2:
            img = np.kron(rng.randint(0, 256, (rows // 3, cols // 4)), np.ones((3, 4))).astype(np.uint8)[:rows, :cols]
            img = np.ascontiguousarray(np.pad(img, ((0, rows - img.shape[0]), (0, cols - img.shape[1])), mode="edge"))
        elif kind == 3:
            img = np.full((rows, cols), 210, np.uint8)
            for _ in range(60):
                x0, y0 = rng.randint(0, cols), rng.randint(0, rows)
                img[y0:y0 + rng.randint(1, 40), x0:x0 + rng.randint(1, 40)] = rng.randint(0, 256)
        elif kind == 4:
            img = _spiral(rows, cols, 3 + trial // 7)
        elif kind == 5:
            img = np.full((rows, cols), 210, np.uint8)
            img[10:rows - 10, 8:cols - 8:5 + trial // 7] = 30
            img[10:14, 8:cols - 8] = 30
        else:
            g = gaussian_filter(rng.standard_normal((rows, cols)), 2.0 + trial // 7)
            img = np.clip(128 + 900 * g, 0, 255).astype(np.uint8)           # smooth blobs: curved borders, touching the frame
        long_borders = kind >= 4
        ctx.set_detector_params(maxMarkerPerimeterRate=40.0 if long_borders else 4.0)
        ctx.stage_frames(img)
        try:
            ctx.run_staged(0, 1, with_ekf=False)
            ctx.sync()
        except capi.AslamError as e:
            assert e.code == -4            # candidate-list capacity on pure-noise input is reported, never silent
            continue
        if long_borders:
            pc.check_contours(ctx, 0, img, perim_rates=(0.03, 40.0))
        else:
            pc.check_stages(ctx, 0, img)


def test_slam_sequence_cfg1_literal_oracle():
    """cfg1 (640x480, 4 markers/frame, 12 landmarks): per-frame comparison against the LITERAL dense-O(N^3) oracle"""
    cfg = synth.CONFIGS["cfg1"]
    w = synth.PanelWorld(cfg)
    stats, ctx, o = pc.run_slam_sequence(cfg, w.lap_length() + 9, batch=8, literal=True, per_frame_check=True)
    assert stats["augments"] == w.L
    assert stats["updates"] > 4 * w.frames_per_panel
    assert stats["max_sigma"] < pc.TIGHT and stats["max_mu"] < 1e-9


def test_slam_sequence_cfg2_headline():
    """cfg2 (1280x720, 20 markers/frame, 200 landmarks): whole lap, batched, against the rank-3 sequential oracle"""
    cfg = synth.CONFIGS["cfg2"]
    w = synth.PanelWorld(cfg)
    stats, ctx, o = pc.run_slam_sequence(cfg, w.lap_length() + 24, batch=24, literal=False)
    assert stats["landmarks"] == w.L                                 # all 200 landmarks entered the map via the augment path
    mu, S = ctx.get_state()
    assert mu.size == 3 + 3 * w.L
    assert stats["max_sigma"] < pc.TIGHT
    # size-independent properties: symmetric to rounding, positive diagonal
    assert np.abs(S - S.T).max() <= 1e-9 * np.abs(S).max()
    assert (np.diag(S) > 0).all()


@pytest.mark.parametrize("name", ["cfg2", "cfg3"])
def test_reference_default_detector_whole_lap(name):
    """THE configuration the reference runs: cv::aruco::detectMarkers with default DetectorParameters (aruco_slam.cpp:313:
    polygonalApproxAccuracyRate 0.05, integer corners), full size, one whole lap through the augment path plus frames of the second
    lap.  Nothing is forced: with these defaults some markers lose to their inner border contour and some observations to the
    covariance gate (aruco_slam.cpp:367-368), so frames fuse FEWER than M corrections; per frame the ids / corners, the gate
    decisions and the augment / update / stationary counts, and per batch the pop order, R, mu and Sigma must equal the oracle's."""
    cfg = synth.CONFIGS[name]
    w = synth.make_world(cfg)
    n = w.lap_length() + 2 * w.frames_per_panel
    stats, ctx, o = pc.run_slam_sequence(cfg, n, batch=16, literal=False, detector="reference",
                                         ctx_kwargs=dict(max_updates_per_frame=24 if w.M <= 24 else 64))
    assert stats["max_sigma"] < pc.TIGHT
    assert stats["frames_short_of_M"] > 0, "the defaults were expected to cost observations (DESIGN.md §5)"
    assert stats["fused_total"] > 0.8 * w.M * (n - w.cfg.n_panels)
    mu, S = ctx.get_state()
    assert np.abs(S - S.T).max() <= 1e-9 * np.abs(S).max() and (np.diag(S) > 0).all()


@pytest.mark.parametrize("detector", ["scene", "reference"])
def test_sliding_visibility_world(detector):
    """the headline sizes on a world whose visible set changes every 2.5 frames (synth.RingWorld): a lap that builds the 200-landmark
    map one marker at a time, then frames of the second lap (every marker known, the landmark set sliding under the windows)"""
    cfg = synth.CONFIGS["cfg2_sliding"]
    w = synth.make_world(cfg)
    n = w.lap_length() + 60
    stats, ctx, o = pc.run_slam_sequence(cfg, n, batch=20, literal=False, detector=detector)
    assert stats["landmarks"] == w.L if detector == "scene" else stats["landmarks"] > 0.8 * w.L
    assert stats["max_sigma"] < pc.TIGHT
    assert stats["fused_total"] > 12 * n


@pytest.mark.gpu
@pytest.mark.parametrize("waves", [256, 4096, 8192])
def test_results_do_not_depend_on_the_number_of_work_queue_waves(waves):
    """the work-queue kernels (k_seg, k_trace_write, k_quads, k_identify) hand out work dynamically: any wave count gives the same frames"""
    cfg = synth.CONFIGS["cfg2"]
    stats, ctx, o = pc.run_slam_sequence(cfg, 48, batch=24, literal=False, ctx_kwargs=dict(persistent_waves=waves))
    assert stats["max_sigma"] < pc.TIGHT


def test_single_frame_api_matches_staged_api():
    """aslam_add_encoder / aslam_add_image (the ArucoSlam::addEncoder / addImage surface) == staged stream API"""
    cfg = synth.CONFIGS["cfg1"]
    w = synth.PanelWorld(cfg)
    D = np.zeros(5)
    a = capi.Context(max_rows=cfg.rows, max_cols=cfg.cols, max_batch=4, max_landmarks=16)
    b = capi.Context(max_rows=cfg.rows, max_cols=cfg.cols, max_batch=4, max_landmarks=16)
    a.set_camera(w.K, D); b.set_camera(w.K, D)
    frs = [w.frame(i) for i in range(4)]
    imgs = [a.synth_render(i, cfg.rows, cfg.cols, w.K, fr.ids, fr.poses, noise_amp=1, seed=i) for i, fr in enumerate(frs)]
    a.stage_encoders([f.wl for f in frs], [f.wr for f in frs], [f.dt for f in frs])
    a.run_staged(0, 4, with_ekf=True); a.sync()
    t = 0.0
    for fr, img in zip(frs, imgs):
        t += fr.dt
        b.add_encoder(fr.wl, fr.wr, t)
        b.add_image(img)
    mu_a, S_a = a.get_state(); mu_b, S_b = b.get_state()
    # identical arithmetic; only dt differs in its last bits (t_k - t_{k-1} of accumulated times vs the staged dt)
    assert np.allclose(mu_a, mu_b, rtol=1e-9, atol=1e-12) and np.allclose(S_a, S_b, rtol=1e-9, atol=1e-15)


def test_no_image_work_before_first_encoder_message():
    cfg = synth.CONFIGS["cfg1"]
    w = synth.PanelWorld(cfg)
    c = capi.Context(max_rows=cfg.rows, max_cols=cfg.cols, max_batch=1, max_landmarks=16)
    c.set_camera(w.K, np.zeros(5))
    fr = w.frame(0)
    img = c.synth_render(0, cfg.rows, cfg.cols, w.K, fr.ids, fr.poses)
    c.add_image(img)                                # aruco_slam.cpp:84-85: ignored until is_init_
    mu, S = c.get_state()
    assert mu.size == 3 and not mu.any() and not S.any()


def test_export_map_records():
    cfg = synth.CONFIGS["cfg1"]
    stats, ctx, o = pc.run_slam_sequence(cfg, 6, batch=6, literal=False)
    rec = np.frombuffer(ctx.export_map().tobytes(), dtype=np.dtype([("id", "<i4"), ("index", "<i4"), ("x", "<f8"), ("y", "<f8"),
                                                                    ("theta", "<f8"), ("S", "<f8", (9,))]))
    mu, S = ctx.get_state()
    L = (mu.size - 3) // 3
    assert (rec["id"][:L] == ctx.get_landmark_ids()).all() and (rec["id"][L:] == -1).all()
    assert np.allclose(rec["x"][:L], mu[3::3]) and np.allclose(rec["theta"][:L], mu[5::3])
    assert np.allclose(rec["S"][0].reshape(3, 3), S[3:6, 3:6])


def test_cfg3_medium_chain_50_markers():
    """BASELINE config 3 (1920x1080, 50 markers/frame): 25..64 fused updates per frame -> the medium EKF chain"""
    cfg = synth.CONFIGS["cfg3"]
    stats, ctx, o = pc.run_slam_sequence(cfg, 20, batch=10, literal=False, ctx_kwargs=dict(max_updates_per_frame=64))
    assert stats["landmarks"] == 100                      # two panels of 50 entered the map
    assert stats["max_sigma"] < pc.TIGHT


def test_cfg3_full_workload_1000_landmarks_50_updates():
    """BASELINE config 3 AS STATED, end to end: the 1920x1080 stream until the map holds 1000 landmarks (one lap through the
    augment path), then frames that fuse 50 corrections each at N = 3003; ids / corners per frame, pop order, observation
    covariances and mu / Sigma against the oracle (rank-3 sequential form) after every batch"""
    cfg = synth.CONFIGS["cfg3"]
    w = synth.PanelWorld(cfg)
    n = w.lap_length() + 12
    stats, ctx, o = pc.run_slam_sequence(cfg, n, batch=16, literal=False, ctx_kwargs=dict(max_updates_per_frame=64))
    assert stats["landmarks"] == 1000
    mu, S = ctx.get_state()
    assert mu.size == 3003
    st = ctx.get_slot_ekf_stats(0, 12)                    # the last batch: second lap, every marker already mapped
    assert (st[:, 0] == 50).all() and (st[:, 2] == 50).all() and (st[:, 1] == 0).all(), st.tolist()
    assert stats["max_sigma"] < pc.TIGHT
    assert np.abs(S - S.T).max() <= 1e-9 * np.abs(S).max() and (np.diag(S) > 0).all()


def test_result_surface_and_persistence_on_the_device(tmp_path):
    """SURVEY §8 f1 / f4 on the real library: toRosPose scatter (aruco_slam.cpp:399-407), the marker arrays (:265-281,
    :336-347), state save / load and the map.txt loader - the bodies of tests/test_host_surface.py"""
    import test_host_surface as ths
    ran = ths.make_ran()
    ths.test_pose_message(ran)
    ths.test_map_markers(ran)
    ths.test_detected_markers(ran)
    ths.test_state_round_trip_through_a_file(ran, tmp_path)
    ths.test_map_file_loader_follows_the_reference_rules(tmp_path)


def test_cfg5_detect_batch_64_frames():
    """BASELINE config 5: 64 x 640x480 frames, 4 markers each, detect + PnP only, through aslam_detect_batch (host frames in)"""
    rows, cols, f = 480, 640, 450.0
    D = np.zeros(5)
    ctx = capi.Context(max_rows=rows, max_cols=cols, max_batch=64, max_landmarks=8)
    frames, expect = [], []
    K = None
    for i in range(64):
        ids, poses, K = synth.simple_scene(rows, cols, f, 4, seed=i)
        frames.append(ctx.synth_render(i, rows, cols, K, ids, poses, noise_amp=2, seed=100 + i))
        expect.append(sorted(ids.tolist()))
    ctx.set_camera(K, D)
    counts, ids_g, corners_g, rv_g, tv_g = ctx.detect_batch(np.stack(frames), max_per_frame=16)
    for i in range(64):
        ids_o, c_o = orc.detect(frames[i])
        n = counts[i]
        assert np.array_equal(ids_o, ids_g[i, :n]) and np.array_equal(c_o, corners_g[i, :n])
        assert sorted(ids_o.tolist()) == expect[i]
        if i % 8 == 0:
            pc.check_poses(ids_o, c_o, rv_g[i, :n], tv_g[i, :n], K, D)


@pytest.mark.parametrize("seed", range(6))
def test_randomised_scenes_against_oracle(seed):
    """frame sizes off the tile grid, bgr / gray input, lens distortion, noise levels, marker counts: ids, corners, poses"""
    rng = np.random.RandomState(100 + seed)
    rows = int(rng.choice([360, 480, 601, 720]))
    cols = int(rng.choice([487, 640, 853, 1280]))
    f = 0.7 * cols
    n = int(rng.randint(1, 9))
    ids, poses, K = synth.simple_scene(rows, cols, f, n, seed=seed, tz=(1.0, 2.4), max_yaw_deg=40.0)
    D = np.zeros(5) if seed % 2 == 0 else np.array([-0.12, 0.06, 0.001, -0.0015, 0.0])
    ctx = capi.Context(max_rows=rows, max_cols=cols, max_batch=1, max_landmarks=16)
    ctx.set_camera(K, D)
    gray = ctx.synth_render(0, rows, cols, K, ids, poses, noise_amp=int(rng.randint(0, 7)), seed=seed,
                            background=int(rng.choice([90, 128, 200])))
    if seed % 3 == 0:
        img = np.stack([gray, gray, gray], -1)
        img[..., 0] = np.roll(gray, 2, 0)                      # channels differ so that the grey conversion matters
        ctx.stage_frames(img)
        gray = orc.bgr2gray(img)
    ctx.run_staged(0, 1, with_ekf=False)
    ctx.sync()
    got, corners, rv, tv = pc.check_stages(ctx, 0, gray)
    if len(got):
        pc.check_poses(got, corners, rv, tv, K, D)


def test_alternating_detection_only_and_full_calls_share_slots_safely():
    """detection-only calls run on the whole GPU, calls with an EKF chain on the CU-masked stream: switching between them on
    the same slots must serialise correctly (events), whatever is still in flight"""
    cfg = synth.CONFIGS["cfg2"]
    w = synth.PanelWorld(cfg)
    n = 24
    frs = [w.frame(i) for i in range(n)]
    a = capi.Context(max_rows=cfg.rows, max_cols=cfg.cols, max_batch=n, max_landmarks=w.L + 8)
    b = capi.Context(max_rows=cfg.rows, max_cols=cfg.cols, max_batch=n, max_landmarks=w.L + 8)
    for c in (a, b):
        c.set_camera(w.K, np.zeros(5))
        for i, fr in enumerate(frs):
            c.synth_render(i, cfg.rows, cfg.cols, w.K, fr.ids, fr.poses, noise_amp=2, seed=i, download=False)
        c.stage_encoders([f.wl for f in frs], [f.wr for f in frs], [f.dt for f in frs])
    a.run_staged(0, n, with_ekf=True); a.sync()                 # reference: one full call
    # b: the same work cut into pieces that hop between the two detection streams without any host synchronisation
    b.run_staged(0, n, with_ekf=False)
    b.run_staged(0, 8, with_ekf=True)
    b.run_staged(8, 16, with_ekf=False)
    b.run_staged(8, 8, with_ekf=True)
    b.run_staged(0, 4, with_ekf=False)                          # overlaps slots whose EKF steps may still be running
    b.run_staged(16, 8, with_ekf=True)
    b.sync()
    mu_a, S_a = a.get_state(); mu_b, S_b = b.get_state()
    # (the calls cut the stream into different EKF windows: the same arithmetic regrouped, equal to rounding)
    assert np.allclose(mu_a, mu_b, rtol=1e-10, atol=1e-13) and np.abs(S_a - S_b).max() <= 1e-10 * np.abs(S_a).max()
    for i in (0, 3, 9, 23):
        da, db = a.get_slot_detections(i), b.get_slot_detections(i)
        assert all(np.array_equal(x, y) for x, y in zip(da, db))


def test_windowed_and_per_frame_ekf_agree_on_the_full_pipeline():
    """cfg2 frames through detection + pose + EKF twice: with the windowed EKF (ekf_window.hip: chain / scan / flush, host planner)
    and with every frame on the per-frame chain (ASLAM_NO_WINDOWS) - the same arithmetic regrouped, equal to rounding; the windowed
    run must really have formed windows"""
    import os
    cfg = synth.CONFIGS["cfg2"]
    w = synth.PanelWorld(cfg)
    n = 96
    frs = [w.frame(i) for i in range(n)]
    ctxs = []
    for windows in (True, False):
        if not windows:
            os.environ["ASLAM_NO_WINDOWS"] = "1"
        try:
            c = capi.Context(max_rows=cfg.rows, max_cols=cfg.cols, max_batch=n, max_landmarks=w.L + 8)
        finally:
            os.environ.pop("ASLAM_NO_WINDOWS", None)
        c.set_camera(w.K, np.zeros(5))
        synth.apply_detector(cfg, c)
        for i, fr in enumerate(frs):
            c.synth_render(i, cfg.rows, cfg.cols, w.K, fr.ids, fr.poses, noise_amp=2, seed=i, download=False)
        c.stage_encoders([f.wl for f in frs], [f.wr for f in frs], [f.dt for f in frs])
        c.profile_enable(True); c.profile_reset()
        c.run_staged(0, n // 2, with_ekf=True)
        c.run_staged(n // 2, n - n // 2, with_ekf=True)
        c.sync()
        ctxs.append(c)
    a, b = ctxs
    assert a.profile_get()["k_ekf_win_step"][0] > 0 and b.profile_get()["k_ekf_win_step"][0] == 0
    mu_a, S_a = a.get_state(); mu_b, S_b = b.get_state()
    assert mu_a.shape == mu_b.shape and mu_a.size > 3 + 3 * 20
    assert np.allclose(mu_a, mu_b, rtol=1e-10, atol=1e-12) and np.abs(S_a - S_b).max() <= 1e-10 * np.abs(S_a).max()
    sa, sb = a.get_slot_ekf_stats(0, n), b.get_slot_ekf_stats(0, n)
    assert np.array_equal(sa, sb)                                # markers, augments, updates, stationary per frame


def test_pipelined_map_gather_delivers_the_same_records():
    """MapGather.gather_pipelined (export enqueued behind the EKF chain, collective one call behind) == the blocking export"""
    import torch
    from aruco_slam_amd.dist import MapGather, MAP_DTYPE
    cfg = synth.CONFIGS["cfg1"]
    w = synth.PanelWorld(cfg)
    n = 8
    ctx = capi.Context(max_rows=cfg.rows, max_cols=cfg.cols, max_batch=n, max_landmarks=16)
    ctx.set_camera(w.K, np.zeros(5))
    frs = [w.frame(i) for i in range(n)]
    for i, fr in enumerate(frs):
        ctx.synth_render(i, cfg.rows, cfg.cols, w.K, fr.ids, fr.poses, noise_amp=1, seed=i, download=False)
    ctx.stage_encoders([f.wl for f in frs], [f.wr for f in frs], [f.dt for f in frs])
    g = MapGather(ctx, device="cuda:0")
    ctx.run_staged(0, 4, with_ekf=True)
    g.gather_pipelined()                                       # nothing delivered yet
    ctx.sync()
    after_first = np.frombuffer(ctx.export_map().tobytes(), dtype=MAP_DTYPE).copy()
    ctx.run_staged(4, 4, with_ekf=True)
    g.gather_pipelined()                                       # delivers the map as of the first call
    torch.cuda.synchronize()
    assert np.array_equal(g.records()[0], after_first)
    g.flush()                                                  # delivers the map as of the second call
    ctx.sync()
    final = np.frombuffer(ctx.export_map().tobytes(), dtype=MAP_DTYPE)
    assert np.array_equal(g.records()[0], final)
    assert (final["id"] >= 0).sum() == len(ctx.get_landmark_ids())


def test_pipelined_map_gather_over_rccl():
    """the same through a real process group (one rank, backend nccl = RCCL), in a child process"""
    import os
    import subprocess
    import sys
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
    out = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "_nccl_gather_worker.py")],
                         env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "nccl gather ok" in out.stdout, out.stdout + out.stderr


def test_library_first_then_torch_share_one_hip_runtime():
    """importing / using the library BEFORE torch must not leave torch without devices (one libamdhip64 in the process)"""
    import os
    import subprocess
    import sys
    out = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "_lib_first_worker.py")],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "one hip runtime ok" in out.stdout, out.stdout + out.stderr


def test_map_gather_through_the_c_abi_rccl():
    """aslam_comm_*: the all-gather a C / C++ node uses (RCCL dlopen'ed by the library, no torch), one rank on this box"""
    from aruco_slam_amd.dist import MAP_DTYPE
    cfg = synth.CONFIGS["cfg1"]
    stats, ctx, o = pc.run_slam_sequence(cfg, 6, batch=6, literal=False)
    uid = capi.Context.comm_unique_id()
    assert len(uid) == 128
    ctx.comm_create(uid, 1, 0)
    rec = np.frombuffer(ctx.comm_gather_maps().tobytes(), dtype=MAP_DTYPE)
    ref = np.frombuffer(ctx.export_map().tobytes(), dtype=MAP_DTYPE)
    assert np.array_equal(rec, ref) and (ref["id"] >= 0).sum() == len(ctx.get_landmark_ids()) > 0
    with pytest.raises(capi.AslamError):
        ctx.comm_create(uid, 1, 0)                 # already created
    ctx.comm_destroy()
