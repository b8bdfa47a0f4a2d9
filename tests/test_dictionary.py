"""Dictionaries handed over as data (aslam_set_dictionary / aslam_set_dictionary_bytes): the stand-in for
cv::aruco::getPredefinedDictionary (aruco_slam.cpp:11-12) for every table that only lives inside OpenCV, e.g. the
DICT_6X6_250 of BASELINE config 0.  Parity against the oracle with the same dictionary, incl. error correction."""
import numpy as np
import pytest

from aruco_slam_amd import capi, synth
from oracle import pyoracle as orc
import parity_common as pc


@pytest.fixture
def dict6():
    bits, maxcorr = synth.random_dictionary(6, 40, 11, seed=3)       # maxCorrectionBits 5 -> int(5 * 0.6) = 3 bits corrected
    orc.set_dictionary(bits, maxcorr)
    yield bits, maxcorr
    orc.set_dictionary(None)


def scene(rows, cols, f, n, seed):
    ids, poses, K = synth.simple_scene(rows, cols, f, n, seed=seed, tz=(0.9, 1.3))
    return np.arange(n, dtype=np.int32) + 5, poses, K


# 6x6 markers are rendered on a white sheet: with the renderer's one-cell quiet zone on a grey background the outline of the
# quiet zone is itself a quad whose corners lie sqrt(2)/8 of the side from the marker's, inside minMarkerDistanceRate * 4
# sides = 0.2 sides, so _filterTooCloseCandidates would keep the (larger) outline and drop the marker.
BG = 255


def run(ctx, rows, cols, K, ids, poses, seed=2):
    img = ctx.synth_render(0, rows, cols, K, ids, poses, noise_amp=2, seed=seed, background=BG)
    ctx.run_staged(0, 1, with_ekf=False)
    ctx.sync()
    return img


def test_custom_6x6_dictionary_matches_oracle(dict6):
    bits, maxcorr = dict6
    rows, cols, f = 240, 320, 300.0
    ids, poses, K = scene(rows, cols, f, 3, 4)
    ctx = capi.Context(max_rows=rows, max_cols=cols, max_batch=1, persistent_waves=4, max_landmarks=16)
    ctx.set_camera(K, np.zeros(5))
    ctx.set_dictionary(bits, maxcorr)
    img = run(ctx, rows, cols, K, ids, poses)
    got, corners, rv, tv = pc.check_stages(ctx, 0, img, expect_ids=ids)
    pc.check_poses(got, corners, rv, tv, K, np.zeros(5))


def test_error_correction_recovers_flipped_cells(dict6):
    bits, maxcorr = dict6
    rows, cols, f = 240, 320, 300.0
    ids, poses, K = scene(rows, cols, f, 3, 6)
    ctx = capi.Context(max_rows=rows, max_cols=cols, max_batch=1, persistent_waves=4, max_landmarks=16)
    ctx.set_camera(K, np.zeros(5))
    damaged = bits.copy()
    damaged[ids[0], 1, 2] ^= 1; damaged[ids[0], 4, 4] ^= 1            # two wrong cells: still within int(5 * 0.6) = 3
    damaged[ids[1], 0, 0] ^= 1; damaged[ids[1], 2, 3] ^= 1; damaged[ids[1], 3, 1] ^= 1; damaged[ids[1], 5, 5] ^= 1   # four: rejected
    ctx.set_dictionary(damaged, maxcorr)                               # the renderer draws from the installed dictionary
    img = ctx.synth_render(0, rows, cols, K, ids, poses, noise_amp=2, seed=1, background=BG)
    ctx.set_dictionary(bits, maxcorr)
    ctx.run_staged(0, 1, with_ekf=False)
    ctx.sync()
    got, _, _, _ = pc.check_stages(ctx, 0, img)
    assert sorted(got.tolist()) == sorted([int(ids[0]), int(ids[2])])


def test_opencv_bytes_list_layout_is_equivalent(dict6):
    bits, maxcorr = dict6
    rows, cols, f = 240, 320, 300.0
    ids, poses, K = scene(rows, cols, f, 2, 8)
    a = capi.Context(max_rows=rows, max_cols=cols, max_batch=1, persistent_waves=4, max_landmarks=16)
    b = capi.Context(max_rows=rows, max_cols=cols, max_batch=1, persistent_waves=4, max_landmarks=16)
    for c in (a, b):
        c.set_camera(K, np.zeros(5))
    a.set_dictionary(bits, maxcorr)
    b.set_dictionary_bytes(synth.opencv_bytes_list(bits), 6, maxcorr)
    ia = run(a, rows, cols, K, ids, poses)
    ib = run(b, rows, cols, K, ids, poses)
    assert np.array_equal(ia, ib)
    ra, rb = a.get_slot_detections(0), b.get_slot_detections(0)
    assert len(ra[0]) == 2 and np.array_equal(ra[0], rb[0]) and np.array_equal(ra[1], rb[1])


def test_bytes_list_of_the_builtin_dictionary_round_trips():
    """DICT_ARUCO_ORIGINAL through the bytesList door == the built-in one (25 bits: the last byte holds a single bit)"""
    rows, cols, f = 240, 320, 300.0
    ids, poses, K = synth.simple_scene(rows, cols, f, 3, seed=1, tz=(0.9, 1.4))
    bits = np.stack([synth.aruco_original_bits(i) for i in range(1024)])
    ob = orc.dict_bytes()                                            # oracle layout: id x rotation x byte
    assert np.array_equal(synth.opencv_bytes_list(bits), ob.transpose(0, 2, 1))
    a = capi.Context(max_rows=rows, max_cols=cols, max_batch=1, persistent_waves=4, max_landmarks=16)
    b = capi.Context(max_rows=rows, max_cols=cols, max_batch=1, persistent_waves=4, max_landmarks=16)
    for c in (a, b):
        c.set_camera(K, np.zeros(5))
    b.set_dictionary_bytes(synth.opencv_bytes_list(bits), 5, 0)
    run(a, rows, cols, K, ids, poses)
    run(b, rows, cols, K, ids, poses)
    ra, rb = a.get_slot_detections(0), b.get_slot_detections(0)
    assert sorted(ra[0].tolist()) == sorted(ids.tolist())
    assert np.array_equal(ra[0], rb[0]) and np.array_equal(ra[1], rb[1])


def test_dictionary_arguments_are_checked():
    ctx = capi.Context(max_rows=64, max_cols=64, max_batch=1, persistent_waves=4, max_landmarks=16)
    with pytest.raises(capi.AslamError):
        ctx.set_dictionary(np.zeros((4, 8, 8), np.uint8))            # marker size > 7


@pytest.mark.gpu
def test_6x6_dictionary_full_frame(dict6):
    bits, maxcorr = dict6
    rows, cols, f = 720, 1280, 900.0
    ids, poses, K = synth.simple_scene(rows, cols, f, 20, seed=3, tz=(1.9, 2.6))
    ids = np.arange(20, dtype=np.int32)
    ctx = capi.Context(max_rows=rows, max_cols=cols, max_batch=1, max_landmarks=16)
    ctx.set_camera(K, np.zeros(5))
    ctx.set_dictionary(bits, maxcorr)
    img = run(ctx, rows, cols, K, ids, poses)
    pc.check_stages(ctx, 0, img, expect_ids=ids)
