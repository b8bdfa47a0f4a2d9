"""cv::aruco::DetectorParameters as data (aslam_set_detector_params): non-default values give the same result as the oracle
run with the same values; what is compiled in is refused, not ignored."""
import numpy as np
import pytest

from aruco_slam_amd import capi, synth
from oracle import pyoracle as orc
import parity_common as pc

ALT = dict(perspectiveRemovePixelPerCell=4, adaptiveThreshConstant=9.5, minMarkerPerimeterRate=0.08, maxMarkerPerimeterRate=3.0, polygonalApproxAccuracyRate=0.03,
           minCornerDistanceRate=0.08, minDistanceToBorder=5, minMarkerDistanceRate=0.02, perspectiveRemoveIgnoredMarginPerCell=0.26,
           maxErroneousBitsInBorderRate=0.2, minOtsuStdDev=8.0, errorCorrectionRate=1.0)


@pytest.fixture
def alt_params():
    orc.set_detector_params(**ALT)
    yield ALT
    orc.set_detector_params()


def _scene(rows, cols, f, n, seed, tz):
    ids, poses, K = synth.simple_scene(rows, cols, f, n, seed=seed, tz=tz)
    ctx = capi.Context(max_rows=rows, max_cols=cols, max_batch=1, persistent_waves=4 if rows < 400 else 0, max_landmarks=16)
    ctx.set_camera(K, np.zeros(5))
    return ctx, ids, poses, K


def test_non_default_parameters_match_oracle(alt_params):
    rows, cols = 240, 320
    ctx, ids, poses, K = _scene(rows, cols, 300.0, 3, 1, (0.9, 1.4))
    ctx.set_detector_params(**alt_params)
    img = ctx.synth_render(0, rows, cols, K, ids, poses, noise_amp=3, seed=5)
    ctx.run_staged(0, 1, with_ekf=False)
    ctx.sync()
    pc.check_stages(ctx, 0, img, expect_ids=ids, perim_rates=(0.08, 3.0), thresh_c=9.5)


def test_parameters_change_the_result():
    """a perimeter floor above every marker's contour removes all detections (the knob is really wired through)"""
    rows, cols = 240, 320
    ctx, ids, poses, K = _scene(rows, cols, 300.0, 3, 1, (0.9, 1.4))
    ctx.synth_render(0, rows, cols, K, ids, poses, noise_amp=3, seed=5, download=False)
    ctx.run_staged(0, 1, with_ekf=False); ctx.sync()
    assert len(ctx.get_slot_detections(0)[0]) == 3
    ctx.set_detector_params(minMarkerPerimeterRate=2.0)
    ctx.run_staged(0, 1, with_ekf=False); ctx.sync()
    assert len(ctx.get_slot_detections(0)[0]) == 0
    ctx.set_detector_params()
    ctx.run_staged(0, 1, with_ekf=False); ctx.sync()
    assert len(ctx.get_slot_detections(0)[0]) == 3


@pytest.mark.parametrize("bad", [dict(adaptiveThreshWinSizeMax=33), dict(perspectiveRemovePixelPerCell=12), dict(markerBorderBits=2),
                                 dict(doCornerRefinement=1, cornerRefinementWinSize=9), dict(maxMarkerPerimeterRate=2000.0), dict(polygonalApproxAccuracyRate=0.0)])
def test_compiled_in_or_invalid_values_are_refused(bad):
    ctx = capi.Context(max_rows=64, max_cols=64, max_batch=1, persistent_waves=4, max_landmarks=16)
    with pytest.raises(capi.AslamError):
        ctx.set_detector_params(**bad)


@pytest.mark.gpu
def test_non_default_parameters_full_frame(alt_params):
    rows, cols = 720, 1280
    ctx, ids, poses, K = _scene(rows, cols, 900.0, 20, 3, (1.9, 2.6))
    ctx.set_detector_params(**alt_params)
    img = ctx.synth_render(0, rows, cols, K, ids, poses, noise_amp=2, seed=7)
    ctx.run_staged(0, 1, with_ekf=False)
    ctx.sync()
    pc.check_stages(ctx, 0, img, expect_ids=ids, perim_rates=(0.08, 3.0), thresh_c=9.5)


@pytest.fixture
def refine_params():
    kw = dict(doCornerRefinement=1, cornerRefinementWinSize=5, cornerRefinementMaxIterations=30, cornerRefinementMinAccuracy=0.1)
    orc.set_detector_params(**kw)
    yield kw
    orc.set_detector_params()


def _refined(ctx, rows, cols, K, ids, poses, kw, seed, bgr=False):
    gray = ctx.synth_render(0, rows, cols, K, ids, poses, noise_amp=2, seed=seed)
    ctx.run_staged(0, 1, with_ekf=False); ctx.sync()
    plain = ctx.get_slot_detections(0)
    ctx.set_detector_params(**kw)
    if bgr:
        img = np.stack([gray, gray, gray], -1)
        ctx.stage_frames(img)
        gray = orc.bgr2gray(img)
    ctx.run_staged(0, 1, with_ekf=False); ctx.sync()
    got, corners, rv, tv = pc.check_stages(ctx, 0, gray, expect_ids=ids)       # final corners compared bit for bit with the oracle's
    assert np.array_equal(np.sort(plain[0]), np.sort(got))
    assert np.abs(corners - np.round(corners)).max() > 1e-3                   # no longer on the pixel grid
    order = [list(plain[0]).index(i) for i in got]
    assert np.abs(corners - plain[1][order]).max() < 5.0                       # within the search window
    return got, corners, rv, tv


def test_corner_refinement_matches_oracle(refine_params):
    rows, cols = 240, 320
    ctx, ids, poses, K = _scene(rows, cols, 300.0, 3, 1, (0.9, 1.4))
    got, corners, rv, tv = _refined(ctx, rows, cols, K, ids, poses, refine_params, 5)
    pc.check_poses(got, corners, rv, tv, K, np.zeros(5))


def test_corner_refinement_near_the_image_border(refine_params):
    """a marker a few pixels from the frame edge: the sampling window leaves the image (clipped / replicated path)"""
    rows, cols = 200, 260
    K = synth.camera_matrix(rows, cols, 260.0)
    R, t = synth.marker_pose((-0.296, -0.192, 0.9), 0.1)
    poses = np.concatenate([R.reshape(-1), t])[None, :]
    ids = np.array([77], np.int32)
    ctx = capi.Context(max_rows=rows, max_cols=cols, max_batch=1, persistent_waves=4, max_landmarks=16)
    ctx.set_camera(K, np.zeros(5))
    got, corners, rv, tv = _refined(ctx, rows, cols, K, ids, poses, refine_params, 2, bgr=True)
    assert corners.min() < 9.0                                                  # really next to the border


@pytest.mark.gpu
def test_corner_refinement_full_frame(refine_params):
    rows, cols = 720, 1280
    ctx, ids, poses, K = _scene(rows, cols, 900.0, 20, 3, (1.9, 2.6))
    got, corners, rv, tv = _refined(ctx, rows, cols, K, ids, poses, refine_params, 7)
    pc.check_poses(got, corners, rv, tv, K, np.zeros(5))


@pytest.mark.parametrize("wins,kw", [((5, 13, 21), dict(adaptiveThreshWinSizeMin=5, adaptiveThreshWinSizeMax=21, adaptiveThreshWinSizeStep=8)),
                                     ((7,), dict(adaptiveThreshWinSizeMin=6, adaptiveThreshWinSizeMax=6, adaptiveThreshWinSizeStep=4)),
                                     ((9, 19), dict(adaptiveThreshWinSizeMin=9, adaptiveThreshWinSizeMax=22, adaptiveThreshWinSizeStep=10))])
def test_other_threshold_windows_match_oracle(wins, kw):
    """window sizes min + i step (even sizes bumped to odd, aruco.cpp::_detectInitialCandidates): 1..3 of them, 3..23 pixels"""
    rows, cols = 240, 320
    ctx, ids, poses, K = _scene(rows, cols, 300.0, 3, 1, (0.9, 1.4))
    orc.set_detector_params(**kw)
    try:
        ctx.set_detector_params(**kw)
        img = ctx.synth_render(0, rows, cols, K, ids, poses, noise_amp=3, seed=5)
        ctx.run_staged(0, 1, with_ekf=False)
        ctx.sync()
        pc.check_stages(ctx, 0, img, expect_ids=ids, windows=wins)
    finally:
        orc.set_detector_params()
