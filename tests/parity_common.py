"""Shared bodies of the parity tests: the HIP path (through the C-ABI) against the CPU oracle on the same inputs.
Bars (BASELINE.json north_star): marker ids, corners, contours, candidates: bit-exact; poses / EKF state: 1e-4
relative (the tests use much tighter bounds where the arithmetic allows)."""
import numpy as np

from aruco_slam_amd import capi, synth
from oracle import pyoracle as orc

POSE_RTOL = 1e-4      # north_star tolerance for poses and state
TIGHT = 1e-9          # what the implementation actually achieves (fused rank-3M form vs sequential oracle)


def nbr_from_binary(fg):
    rows, cols = fg.shape
    pad = np.pad(fg.astype(bool), 1)
    dxs = [1, 1, 0, -1, -1, -1, 0, 1]
    dys = [0, -1, -1, -1, 0, 1, 1, 1]
    m = np.zeros((rows, cols), np.uint8)
    for d, (dx, dy) in enumerate(zip(dxs, dys)):
        m |= pad[1 + dy:1 + dy + rows, 1 + dx:1 + dx + cols].astype(np.uint8) << d
    return m


def check_contours(ctx, slot, img, perim_rates=(0.03, 4.0), thresh_c=7.0, windows=(3, 13, 23)):
    """threshold -> contours of one staged frame against the oracle's sequential Suzuki-Abe scan: bit-identical"""
    rows, cols = img.shape
    lo, hi = int(perim_rates[0] * max(rows, cols)), int(perim_rates[1] * max(rows, cols))
    for s, k in enumerate(windows):
        th = orc.threshold(img, k, thresh_c)
        assert np.array_equal(ctx.debug_nbr(slot, s, rows, cols), nbr_from_binary(th > 0)), f"neighbour masks differ at scale {s}"
        sizes, keys, hole, pts = orc.find_contours(th)
        sel = (sizes >= lo) & (sizes <= hi)
        gs, gk, gp = ctx.debug_contours(slot, s)
        offs = np.concatenate([[0], np.cumsum(sizes)])
        opts = (np.concatenate([pts[offs[i]:offs[i + 1]] for i in np.nonzero(sel)[0]]) if sel.any() else np.zeros((0, 2), np.int32))
        assert np.array_equal(sizes[sel], gs), f"contour sizes differ at scale {s}"
        assert np.array_equal(keys[sel], gk), f"contour order differs at scale {s}"
        assert np.array_equal(opts, gp), f"contour points differ at scale {s}"


def check_stages(ctx, slot, img, expect_ids=None, perim_rates=(0.03, 4.0), thresh_c=7.0, windows=(3, 13, 23)):
    """threshold -> contours -> candidates -> detections of one staged frame against the oracle."""
    check_contours(ctx, slot, img, perim_rates, thresh_c, windows)
    co, so, _, _ = orc.candidates(img, 0)
    cg, sg, _ = ctx.debug_candidates(slot, 0)
    assert np.array_equal(co, cg) and np.array_equal(so, sg), "quad candidates differ"
    co, so, _, _ = orc.candidates(img, 2)
    cg, sg, _ = ctx.debug_candidates(slot, 2)
    assert np.array_equal(co, cg) and np.array_equal(so, sg), "filtered candidates differ"
    ids_o, c_o = orc.detect(img)
    ids_g, c_g, rv_g, tv_g = ctx.get_slot_detections(slot)
    assert np.array_equal(ids_o, ids_g), f"marker ids differ: {ids_o} vs {ids_g}"
    assert np.array_equal(c_o, c_g), "marker corners differ"
    if expect_ids is not None:
        assert sorted(ids_g.tolist()) == sorted(np.asarray(expect_ids).tolist()), "rendered markers not all detected"
    return ids_g, c_g, rv_g, tv_g


def check_poses(ids, corners, rv_g, tv_g, K, D, marker_length=0.27, rtol=POSE_RTOL):
    for j in range(len(ids)):
        rv, tv, _ = orc.solve_pnp(corners[j], marker_length, K, D)
        assert np.allclose(rv, rv_g[j], rtol=rtol, atol=rtol * 1e-2), f"rvec differs for marker {ids[j]}"
        assert np.allclose(tv, tv_g[j], rtol=rtol, atol=rtol * 1e-2), f"tvec differs for marker {ids[j]}"


def run_slam_sequence(cfg, n_frames, batch, literal, noise_amp=2, per_frame_check=False, ctx_kwargs=None, D=None, detector="scene"):
    """Drive the HIP path (staged stream API) and the oracle over the same synthetic tour; compare after each batch.
    detector: "scene" = the scene's own DetectorParameters profile (synth.CONFIGS), "reference" = the defaults, which is what
    the reference runs (aruco_slam.cpp:313).  Nothing is forced either way: per frame, the detections, the observations that pass
    the gates (aruco_slam.cpp:327-333, 367-368) and the augment / update / stationary counts must EQUAL the oracle's."""
    w = synth.make_world(cfg)
    D = np.zeros(5) if D is None else D
    kw = dict(max_rows=cfg.rows, max_cols=cfg.cols, max_batch=batch, max_landmarks=max(w.L + 8, 16),
              r2c_t=(cfg.r2c[0], cfg.r2c[1], 0.0))
    kw.update(ctx_kwargs or {})
    ctx = capi.Context(**kw)
    ctx.set_camera(w.K, D)
    o = orc.Slam(r2c_tx=cfg.r2c[0], r2c_ty=cfg.r2c[1], literal=literal)
    o.set_camera(w.K, D)
    if detector == "scene":
        synth.apply_detector(cfg, ctx, o)
    t_now = 0.0
    stats = dict(frames=0, updates=0, augments=0, stationary=0, max_mu=0.0, max_sigma=0.0, frames_short_of_M=0, markers=0, fused_total=0)
    for f0 in range(0, n_frames, batch):
        nb = min(batch, n_frames - f0)
        frs = [w.frame(f0 + i) for i in range(nb)]
        imgs = [ctx.synth_render(i, cfg.rows, cfg.cols, w.K, fr.ids, fr.poses, noise_amp=noise_amp, seed=f0 + i)
                for i, fr in enumerate(frs)]
        ctx.stage_encoders([fr.wl for fr in frs], [fr.wr for fr in frs], [fr.dt for fr in frs])
        groups = [(i, 1) for i in range(nb)] if per_frame_check else [(0, nb)]
        for first, cnt in groups:
            ctx.run_staged(first, cnt, with_ekf=True)
            ctx.sync()
            expect_stats = []
            for i in range(first, first + cnt):
                t_now += frs[i].dt
                o.add_encoder(frs[i].wl, frs[i].wr, t_now)
                o.add_image(imgs[i])
                d_ids, d_c = o.log_detections()[:2]
                g_ids, g_c = ctx.get_slot_detections(i)[:2]
                assert np.array_equal(d_ids, g_ids), f"frame {f0 + i}: marker ids differ"
                assert np.array_equal(d_c, g_c), f"frame {f0 + i}: marker corners differ"
                # gate decisions (range gate aruco_slam.cpp:327-333, covariance gate :367-368): the observations that reach the queue
                fi, _, fa, _, _ = o.log_observations()
                r_ids, r_valid, _, _ = ctx.get_slot_raw_observations(i)
                assert sorted(r_ids[r_valid == 1].tolist()) == sorted(fi.tolist()), f"frame {f0 + i}: gate decisions differ"
                expect_stats.append([len(d_ids), int((fa == 0).sum()), int((fa == 1).sum()), int((fa == 2).sum())])
                stats["frames_short_of_M"] += int(len(fi) < w.M)
                stats["markers"] += len(d_ids)
                stats["fused_total"] += int((fa == 1).sum())
            got_stats = ctx.get_slot_ekf_stats(first, cnt)
            assert np.array_equal(got_stats, np.array(expect_stats)), "per-frame detections / augments / updates / stationary counts differ"
            oi, ox, oa, oz, oR = o.log_observations()
            gi, gx, ga, gz, gR = ctx.get_observations()
            assert np.array_equal(oi, gi), "pop order (ids) differs"
            assert np.array_equal(ox, gx), "landmark indices differ"
            assert np.array_equal(oa, ga), "update/augment/stationary decisions differ"
            assert np.allclose(oz, gz, rtol=POSE_RTOL, atol=1e-9)
            # observe_covariance_ of ArucoSlam::CalculateCovariance (aruco_slam.cpp:437-471), directly: diag(e R_x + 1e-2, ...)
            oRd = np.stack([oR[:, k, k] for k in range(3)], -1) if len(oR) else np.zeros((0, 3))
            assert len(oR) == 0 or np.all(oR == oRd[:, :, None] * np.eye(3)), "the reference's observation covariance is diagonal"
            assert np.allclose(oRd, gR, rtol=POSE_RTOL, atol=1e-12), f"observation covariance differs by {np.abs(oRd - gR).max()}"
            st = ctx.get_slot_ekf_stats(first + cnt - 1, 1)[0]
            assert st[1] == int((ga == 0).sum()) and st[2] == int((ga == 1).sum()) and st[3] == int((ga == 2).sum()), "per-slot EKF stats differ"
            mu_o, S_o = o.get_state()
            mu_g, S_g = ctx.get_state()
            assert mu_o.shape == mu_g.shape, "state size differs"
            e_mu = np.abs(mu_o - mu_g).max()
            e_S = np.abs(S_o - S_g).max() / max(np.abs(S_o).max(), 1e-300)
            stats["max_mu"] = max(stats["max_mu"], e_mu)
            stats["max_sigma"] = max(stats["max_sigma"], e_S)
            assert np.allclose(mu_o, mu_g, rtol=POSE_RTOL, atol=1e-8), f"mu differs by {e_mu}"
            assert e_S < POSE_RTOL, f"sigma differs by {e_S} (relative to max |sigma|)"
            stats["updates"] += int((ga == 1).sum()); stats["augments"] += int((ga == 0).sum()); stats["stationary"] += int((ga == 2).sum())
        stats["frames"] += nb
    assert np.array_equal(o.landmark_ids(), ctx.get_landmark_ids()), "landmark id table differs"
    stats["landmarks"] = len(ctx.get_landmark_ids())
    return stats, ctx, o
