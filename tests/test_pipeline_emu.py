"""Small-size parity of the kernel LOGIC without a GPU: the same kernel sources compiled against the CPU emulation
of the HIP subset (tests/hipemu) are driven through the same C-ABI and compared with the oracle.  On a box with a GPU
these run against the real library instead."""
import numpy as np
import pytest

from aruco_slam_amd import capi, synth
from oracle import pyoracle as orc
import parity_common as pc


def small_ctx(rows, cols, batch=1, **kw):
    args = dict(max_rows=rows, max_cols=cols, max_batch=batch, persistent_waves=4, max_landmarks=16)
    args.update(kw)
    return capi.Context(**args)


def test_detector_stages_small_scene():
    rows, cols, f = 240, 320, 300.0
    ids, poses, K = synth.simple_scene(rows, cols, f, 3, seed=1, tz=(0.9, 1.4))
    ctx = small_ctx(rows, cols)
    D = np.zeros(5)
    ctx.set_camera(K, D)
    img = ctx.synth_render(0, rows, cols, K, ids, poses, noise_amp=3, seed=5)
    ctx.run_staged(0, 1, with_ekf=False)
    ctx.sync()
    got_ids, corners, rv, tv = pc.check_stages(ctx, 0, img, expect_ids=ids)
    pc.check_poses(got_ids, corners, rv, tv, K, D)
    # list sizes of the pass (aslam_debug_get_frame_counts): every kept contour has its write tickets, 64 points each
    fc = ctx.debug_frame_counts(0)
    n_contours = sum(len(ctx.debug_contours(0, s)[0]) for s in range(3))
    assert fc["contours"] == n_contours and fc["nodes"] >= fc["contours"] and fc["serial_link"] == 0
    assert fc["points"] <= 64 * fc["write_tickets"] < fc["points"] + 64 * fc["contours"]


def test_detector_bgr_and_odd_size():
    rows, cols, f = 150, 205, 200.0            # not a multiple of the 64x32 tile
    ids, poses, K = synth.simple_scene(rows, cols, f, 1, seed=2, tz=(0.7, 0.9))
    ctx = small_ctx(rows, cols)
    ctx.set_camera(K, np.zeros(5))
    gray = ctx.synth_render(0, rows, cols, K, ids, poses, noise_amp=2, seed=1)
    bgr = np.stack([gray, np.roll(gray, 1, 1), gray], -1)
    ctx.stage_frames(bgr)
    ctx.run_staged(0, 1, with_ekf=False)
    ctx.sync()
    pc.check_stages(ctx, 0, orc.bgr2gray(bgr))
    ids_o, c_o = orc.detect(bgr)
    ids_g, c_g, _, _ = ctx.get_slot_detections(0)
    assert np.array_equal(ids_o, ids_g) and np.array_equal(c_o, c_g)


def _spiral(rows, cols, step):
    """a one-pixel-wide dark spiral on a bright ground: one very long border in a small area (many nodes on one cycle)"""
    img = np.full((rows, cols), 220, np.uint8)
    x0, y0, x1, y1 = 4, 4, cols - 5, rows - 5
    while x1 - x0 > 2 * step and y1 - y0 > 2 * step:
        img[y0, x0:x1 + 1] = 20
        img[y0:y1 + 1, x1] = 20
        img[y1, x0 + step:x1 + 1] = 20
        img[y0 + step:y1 + 1, x0 + step] = 20
        img[y0 + step, x0 + step:x1 - step + 1] = 20
        x0 += step; y0 += step; x1 -= step; y1 -= step
    return img


@pytest.mark.parametrize("link", ["lds", "serial"])
def test_contours_degenerate_images(link, monkeypatch):
    """blocky noise, overlapping rectangles (axis-aligned borders: segments as long as the cut lattice allows), a spiral and a comb
    (one border of thousands of points through hundreds of nodes): contours bit-identical to the sequential scan, through both
    forms of the cycle resolution (k_link in LDS, k_link_serial through global memory)"""
    if link == "serial":
        monkeypatch.setenv("ASLAM_LINK_LDS_NODES", "0")
    rng = np.random.RandomState(0)
    checked = 0
    for rows, cols, trials in ((64, 96, 6), (150, 200, 5)):
        ctx = small_ctx(rows, cols, cap_starts_per_frame=1 << 16, cap_contours_per_frame=1 << 13, cap_points_per_frame=1 << 19)
        ctx.set_camera(synth.camera_matrix(rows, cols, 100.0), np.zeros(5))
        ctx.set_detector_params(maxMarkerPerimeterRate=40.0)       # keep the long borders: they are the point
        for trial in range(trials):
            if trial % 5 == 2 and rows > 100:
                img = rng.randint(0, 256, (rows, cols)).astype(np.uint8)   # pure noise: more kept borders than k_link has slots -> it hands the frame over
            elif trial % 5 == 3:
                img = _spiral(rows, cols, 4 + trial)
            elif trial % 5 == 4:
                img = np.full((rows, cols), 210, np.uint8)
                img[10:rows - 10, 8:cols - 8:6] = 30                # a comb: teeth 1 px wide ...
                img[10:14, 8:cols - 8] = 30                         # ... on one spine
            elif trial % 2 == 0:
                img = np.kron(rng.randint(0, 256, (rows // 4 + 1, cols // 4 + 1)), np.ones((4, 4))).astype(np.uint8)[:rows, :cols].copy()
            else:
                img = np.full((rows, cols), 200, np.uint8)
                for _ in range(25):
                    x0, y0 = rng.randint(0, cols), rng.randint(0, rows)
                    img[y0:y0 + rng.randint(1, 30), x0:x0 + rng.randint(1, 30)] = rng.randint(0, 256)
            ctx.stage_frames(img)
            try:
                ctx.run_staged(0, 1, with_ekf=False)
                ctx.sync()
            except capi.AslamError as e:
                assert e.code == -4
                continue
            pc.check_contours(ctx, 0, img, perim_rates=(0.03, 40.0))
            checked += 1
    assert checked >= 8


def test_empty_and_blank_frames():
    rows, cols = 64, 96
    ctx = small_ctx(rows, cols)
    ctx.set_camera(synth.camera_matrix(rows, cols, 100.0), np.zeros(5))
    for value in (0, 128, 255):
        ctx.stage_frames(np.full((rows, cols), value, np.uint8))
        ctx.run_staged(0, 1, with_ekf=False)
        ctx.sync()
        assert len(ctx.get_slot_detections(0)[0]) == 0


def test_slam_sequence_small_literal():
    cfg = synth.SceneConfig(rows=240, cols=320, f=225.0, grid=(2, 2), n_panels=3, col_spacing=0.9, row_spacing=0.7, step=0.05,
                            tz_far=2.4, tz_near=1.9, r2c=(0.1, -0.05))
    w = synth.PanelWorld(cfg)
    stats, ctx, o = pc.run_slam_sequence(cfg, w.lap_length() + 5, batch=4, literal=True, per_frame_check=True,
                                         ctx_kwargs=dict(persistent_waves=4))
    assert stats["augments"] == w.L and stats["updates"] > 20
    assert stats["max_sigma"] < pc.TIGHT


def test_argument_validation():
    ctx = small_ctx(64, 96)
    with pytest.raises(capi.AslamError):
        ctx.stage_frames(np.zeros((65, 96), np.uint8))        # larger than max_rows
    with pytest.raises(capi.AslamError):
        ctx.run_staged(0, 2, with_ekf=False)                   # outside max_batch
    ctx.stage_frames(np.zeros((64, 96), np.uint8))
    with pytest.raises(capi.AslamError) as e:
        ctx.run_staged(0, 1, with_ekf=False)                   # camera parameters not set
    assert e.value.code == -5
    with pytest.raises(capi.AslamError):
        capi.Context(markers_dictionary=10)                    # only DICT_ARUCO_ORIGINAL can be generated offline
