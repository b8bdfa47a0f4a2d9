"""Known-answer tests that pin the CPU oracle (oracle/).  The reference ships no tests or fixtures
(SURVEY.md §4), so these KATs are hand-derived from reading the reference source (file:line cited per test)."""
import math

import numpy as np

from oracle import pyoracle as orc
from aruco_slam_amd import synth


def test_norm_angle_wraps_once():                       # aruco_slam.cpp:412-421
    assert orc.norm_angle(math.pi) == math.pi - 2 * math.pi
    assert orc.norm_angle(-math.pi) == -math.pi          # strict '<'
    assert orc.norm_angle(3 * math.pi) == math.pi        # wrapped once only: still >= pi
    assert abs(orc.norm_angle(-3.2) - (-3.2 + 2 * math.pi)) < 1e-15
    assert orc.norm_angle(0.3) == 0.3


def test_predict_first_call_only_arms():                 # aruco_slam.cpp:24-29
    s = orc.Slam()
    s.add_encoder(5.0, 7.0, 10.0)
    mu, S = s.get_state()
    assert not mu.any() and not S.any()


def test_predict_kat():                                  # aruco_slam.cpp:35-73, SURVEY §4
    for literal in (True, False):
        s = orc.Slam(kl=0.05, kr=0.05, b=0.09, Q_k=0.01, literal=literal)
        s.add_encoder(1.0, 1.0, 0.0)
        s.add_encoder(1.0, 1.0, 1.0)
        mu, S = s.get_state()
        assert np.allclose(mu, [0.05, 0.0, 0.0], atol=1e-17)
        exp = np.diag([1.25e-5, 0.0, 2 * (0.5 * 0.05 / 0.09) ** 2 * 0.01])
        assert np.allclose(S, exp, rtol=1e-13, atol=1e-20)
        assert abs(S[2, 2] - 1.5432098765e-3) < 1e-12


def test_predict_uses_kl_for_both_wheels_in_noise():     # quirk Q7, aruco_slam.cpp:62
    a = orc.Slam(kl=0.05, kr=0.08); b = orc.Slam(kl=0.05, kr=0.05)
    for s in (a, b):
        s.add_encoder(0, 0, 0.0); s.add_encoder(0.0, 2.0, 1.0)
    Sa = a.get_state()[1]; Sb = b.get_state()[1]
    # the heading-noise term does not depend on kr: (0.5*kl*dt/b)^2 * Q_k*|wr|
    assert abs(Sa[2, 2] - Sb[2, 2]) < 1e-18


def test_predict_leaves_landmark_block_untouched():      # Hx is identity outside the 3x3 corner, :64-73
    rng = np.random.RandomState(0)
    N = 12
    A = rng.randn(N, N); S0 = A @ A.T * 1e-3
    mu0 = rng.randn(N)
    for literal in (True, False):
        s = orc.Slam(literal=literal)
        s.add_encoder(0, 0, 0.0)
        s.set_state(mu0, S0, [5, 6, 7])
        s.add_encoder(2.0, 3.0, 0.5)
        mu, S = s.get_state()
        assert np.array_equal(mu[3:], mu0[3:])
        assert np.array_equal(S[3:, 3:], S0[3:, 3:])
        assert not np.array_equal(S[:3, :], S0[:3, :])


def test_priority_queue_pop_order_probe():               # aruco_slam.h:85-88; SURVEY §4 probe (g++ 11.4 libstdc++)
    assert orc.heap_order([-1] * 20).tolist() == [0, 2, 6, 14, 19, 18, 17, 16, 13, 15, 12, 11, 10, 9, 8, 5, 7, 4, 1, 3]
    # new (-1) first, then ascending landmark index
    order = orc.heap_order([4, -1, 2, 7, -1, 0])
    idx = np.array([4, -1, 2, 7, -1, 0])[order]
    assert idx.tolist() == sorted(idx.tolist())


def test_dictionary_aruco_original():                    # parameters.yaml:16, aruco_slam.cpp:11-12
    b = orc.dict_bits(0)
    assert b.tolist() == [[1, 0, 0, 0, 0]] * 5            # id 0: every row is the word 10000
    assert orc.dict_bits(1023).tolist() == [[0, 1, 1, 1, 0]] * 5
    assert orc.dict_bits(1)[4].tolist() == [1, 0, 1, 1, 1] and orc.dict_bits(1)[0].tolist() == [1, 0, 0, 0, 0]
    assert np.array_equal(orc.dict_bits(77), synth.aruco_original_bits(77))
    by = orc.dict_bytes()
    assert by.shape == (1024, 4, 4)
    # rotation 0 bytes are the row-major bits, MSB first, 25 bits in 4 bytes (last byte holds one bit)
    bits = orc.dict_bits(77).reshape(-1)
    val = [int("".join(map(str, bits[i:i + 8])), 2) for i in (0, 8, 16)] + [int(bits[24])]
    assert by[77, 0].tolist() == val


def test_bgr2gray_fixed_point():                         # cvtColor BGR2GRAY 8u: (B*1868 + G*9617 + R*4899 + 8192) >> 14
    px = np.array([[[255, 0, 0], [0, 255, 0], [0, 0, 255], [255, 255, 255], [10, 20, 30]]], np.uint8)
    g = orc.bgr2gray(px)
    assert g.tolist() == [[29, 150, 76, 255, (10 * 1868 + 20 * 9617 + 30 * 4899 + 8192) >> 14]]


def test_box_mean_and_threshold():                       # adaptiveThreshold(MEAN_C, BINARY_INV, k, 7), BORDER_REPLICATE
    rng = np.random.RandomState(3)
    img = rng.randint(0, 256, (37, 53)).astype(np.uint8)
    for k in (3, 13, 23):
        r = k // 2
        pad = np.pad(img.astype(np.int64), r, mode="edge")
        ref = np.zeros(img.shape, np.int64)
        for dy in range(k):
            for dx in range(k):
                ref += pad[dy:dy + img.shape[0], dx:dx + img.shape[1]]
        mean = np.floor(ref / (k * k) + 0.5).astype(np.int64)      # k*k odd: no ties
        assert np.array_equal(orc.box_mean(img, k), mean.astype(np.uint8))
        th = orc.threshold(img, k)
        assert np.array_equal(th > 0, img.astype(np.int64) - mean <= -7)


def test_find_contours_square_with_hole():               # findContours RETR_LIST / CHAIN_APPROX_NONE
    b = np.zeros((12, 12), np.uint8)
    b[2:9, 3:10] = 255
    b[4:7, 5:8] = 0
    sizes, keys, hole, pts = orc.find_contours(b)
    # returned in reverse discovery order: hole border first, outer border last
    assert hole.tolist() == [1, 0]
    assert sizes.tolist() == [12, 24]                     # hole border skips the 4 diagonal corner pixels (8-connectivity)
    outer = pts[12:]
    assert outer[0].tolist() == [3, 2]                    # starts at the top-left pixel
    assert outer[1].tolist() == [3, 3]                    # and runs down first (counter-clockwise on screen)
    assert keys.tolist() == [4 * 12 + 5, 2 * 12 + 3]      # discovery positions: hole at its first 0-pixel, outer at its first pixel
    single = np.zeros((5, 5), np.uint8); single[2, 2] = 1
    s2, _, _, p2 = orc.find_contours(single)
    assert s2.tolist() == [1] and p2.tolist() == [[2, 2]]


def test_approx_poly_square():                           # approxPolyDP closed
    b = np.zeros((60, 60), np.uint8)
    b[10:51, 15:46] = 255
    sizes, _, _, pts = orc.find_contours(b)
    a = orc.approx_poly(pts, sizes[0] * 0.05)
    assert sorted(map(tuple, a.tolist())) == [(15, 10), (15, 50), (45, 10), (45, 50)]


def test_rodrigues_roundtrip_and_jacobian():             # cv::Rodrigues (aruco_slam.cpp:354)
    rng = np.random.RandomState(5)
    for _ in range(20):
        r = rng.randn(3)
        r *= rng.uniform(0.01, 3.1) / np.linalg.norm(r)        # rotation angle < pi so that the inverse is unique
        R, J = orc.rodrigues(r)
        assert np.allclose(R @ R.T, np.eye(3), atol=1e-14) and abs(np.linalg.det(R) - 1) < 1e-14
        assert np.allclose(orc.rodrigues_inv(R), r, atol=1e-12)
        eps = 1e-7
        for i in range(3):
            d = np.zeros(3); d[i] = eps
            num = (orc.rodrigues(r + d)[0] - orc.rodrigues(r - d)[0]) / (2 * eps)
            assert np.allclose(J[i].reshape(3, 3), num, atol=1e-7)
    R0, _ = orc.rodrigues(np.zeros(3))
    assert np.array_equal(R0, np.eye(3))


def test_project_points_jacobians():                     # cv::projectPoints (aruco_slam.cpp:441)
    rng = np.random.RandomState(6)
    K = np.array([[525.0, 0, 472.8], [0, 525.2, 264.7], [0, 0, 1]])
    D = np.array([0.0416, -0.0477, -0.00326, -0.00399, 0.0111])
    obj = np.array([[-0.135, 0.135, 0], [0.135, 0.135, 0], [0.135, -0.135, 0], [-0.135, -0.135, 0]])
    r = np.array([2.9, 0.2, -0.3]); t = np.array([0.2, -0.1, 1.7])
    p, dr, dt = orc.project_points(obj, r, t, K, D)
    eps = 1e-7
    for i in range(3):
        d = np.zeros(3); d[i] = eps
        num_r = (orc.project_points(obj, r + d, t, K, D)[0] - orc.project_points(obj, r - d, t, K, D)[0]) / (2 * eps)
        num_t = (orc.project_points(obj, r, t + d, K, D)[0] - orc.project_points(obj, r, t - d, K, D)[0]) / (2 * eps)
        assert np.allclose(dr[:, i], num_r.reshape(-1), rtol=1e-5, atol=1e-5)
        assert np.allclose(dt[:, i], num_t.reshape(-1), rtol=1e-5, atol=1e-5)


def test_solve_pnp_recovers_exact_pose():                # estimatePoseSingleMarkers (aruco_slam.cpp:314)
    rng = np.random.RandomState(8)
    K = synth.camera_matrix(720, 1280, 900.0)
    obj = np.array([[-0.135, 0.135, 0], [0.135, 0.135, 0], [0.135, -0.135, 0], [-0.135, -0.135, 0]], np.float32).astype(float)
    for D in (np.zeros(5), np.array([0.0416, -0.0477, -0.00326, -0.00399, 0.0111])):
        for _ in range(10):
            R, t = synth.marker_pose((rng.uniform(-0.8, 0.8), rng.uniform(-0.4, 0.4), rng.uniform(1.2, 2.9)), rng.uniform(-0.6, 0.6))
            r = orc.rodrigues_inv(R)
            px, _, _ = orc.project_points(obj, r, t, K, D)
            rv, tv, it = orc.solve_pnp(px.astype(np.float32), 0.27, K, D)
            p2, _, _ = orc.project_points(obj, rv, tv, K, D)
            assert np.abs(p2 - px.astype(np.float32)).max() < 1e-3            # float32 corners limit the fit
            assert np.allclose(tv, t, rtol=2e-4, atol=2e-4) and it <= 20


def test_observation_covariance_formula():               # aruco_slam.cpp:465-470 via the full add_poses path
    K = synth.camera_matrix(480, 640, 450.0)
    s = orc.Slam(r2c_tx=0.18, r2c_ty=-0.1)
    s.set_camera(K, np.zeros(5))
    s.add_encoder(0, 0, 0.0)
    R, t = synth.marker_pose((0.1, 0.05, 1.5), 0.2)
    r = orc.rodrigues_inv(R)
    obj = np.array([[-0.135, 0.135, 0], [0.135, 0.135, 0], [0.135, -0.135, 0], [-0.135, -0.135, 0]], np.float32).astype(float)
    px, _, _ = orc.project_points(obj, r, t, K, np.zeros(5))
    corners = (np.round(px * 4) / 4).astype(np.float32)       # small residual so the covariance gate (:367) passes
    s.add_poses([7], corners[None], r[None], t[None])
    ids, idx, act, xyth, Rm = s.log_observations()
    assert ids.tolist() == [7] and idx.tolist() == [-1] and act.tolist() == [0]
    pf = px.astype(np.float32).astype(float)
    e = (np.sum((corners - pf) ** 2) / 4.0 / np.linalg.norm(corners[0] - corners[2])) * (np.linalg.norm(t) / 0.27)
    assert np.allclose(np.diag(Rm[0]), [e * 100 + 1e-2, e * 100 + 1e-2, e * 10 + 1e-3], rtol=1e-12)
    assert np.allclose(xyth[0], [t[2] + 0.18, -t[0] - 0.1, math.pi - 0.2], atol=1e-12)
    mu, S = s.get_state()
    assert mu.size == 6 and np.allclose(mu[3:], xyth[0], atol=1e-12)          # robot at the origin: landmark = observation


def test_range_gate_is_3m_float():                       # quirk Q11: key typo leaves the default 3 (aruco_slam.h:58)
    K = synth.camera_matrix(480, 640, 450.0)
    s = orc.Slam(); s.set_camera(K, np.zeros(5)); s.add_encoder(0, 0, 0.0)
    obj = np.array([[-0.135, 0.135, 0], [0.135, 0.135, 0], [0.135, -0.135, 0], [-0.135, -0.135, 0]])
    for z, kept in ((2.99, 1), (3.01, 0)):
        R, t = synth.marker_pose((0.0, 0.0, z), 0.0)
        r = orc.rodrigues_inv(R)
        px, _, _ = orc.project_points(obj, r, t, K, np.zeros(5))
        s.add_poses([3], px.astype(np.float32)[None], r[None], t[None])
        assert len(s.log_observations()[0]) == kept


def test_oracle_detects_rendered_markers_python_renderer():
    """independent of the product renderer: a numpy-rendered fronto-parallel marker is found with the right id/corners"""
    img = np.full((200, 240), 128, np.uint8)
    cell = 12
    bits = synth.aruco_original_bits(123)
    x0, y0 = 70, 50
    img[y0 - cell:y0 + 8 * cell, x0 - cell:x0 + 8 * cell] = 255
    img[y0:y0 + 7 * cell, x0:x0 + 7 * cell] = 0
    for r in range(5):
        for c in range(5):
            if bits[r, c]:
                img[y0 + (r + 1) * cell:y0 + (r + 2) * cell, x0 + (c + 1) * cell:x0 + (c + 2) * cell] = 255
    ids, corners = orc.detect(img)
    assert ids.tolist() == [123]
    assert corners[0].tolist() == [[x0, y0], [x0 + 7 * cell - 1, y0], [x0 + 7 * cell - 1, y0 + 7 * cell - 1], [x0, y0 + 7 * cell - 1]]
    # rotate the image by 90 degrees: same id, corner list still starts at the marker's own top-left corner
    ids2, corners2 = orc.detect(np.ascontiguousarray(np.rot90(img)))
    assert ids2.tolist() == [123]
    assert corners2[0][0].tolist() == [y0, 240 - 1 - x0]
