"""development aid: the EKF alone (observations injected, no detector) on three scenes - cfg2 panels, cfg2 sliding ring, cfg3 panels -
per-kernel HIP-event times per frame and the planner's window statistics.  usage: python scripts/ekf_window_timing.py [cfg2 cfg2_sliding cfg3]"""
import math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from aruco_slam_amd import capi, synth


def observations(world, fr, rng):
    px, py, phi = fr.true_pose
    c, s = math.cos(phi), math.sin(phi)
    out = []
    for li in fr.landmark_index:
        wx, wy, wth = world.world[li]
        dx, dy = wx - px, wy - py
        th = (wth - phi + math.pi) % (2 * math.pi) - math.pi
        out.append((dx * c + dy * s + rng.normal(0, 2e-3), -dx * s + dy * c + rng.normal(0, 2e-3), th + rng.normal(0, 1e-3)))
    return np.array(out).reshape(-1, 3)


def run(name, laps=2):
    cfg = synth.CONFIGS[name]
    w = synth.make_world(cfg)
    lap = w.lap_length()
    ctx = capi.Context(max_rows=64, max_cols=64, max_batch=lap, max_landmarks=w.L + 8, max_updates_per_frame=24 if w.M <= 24 else 64)
    ctx.set_camera(synth.camera_matrix(64, 64, 50.0), np.zeros(5))
    rng = np.random.RandomState(3)
    frames = [w.frame(i) for i in range(lap)]
    turn = w.frame(lap)
    def stage(first_lap):
        enc = [(f.wl, f.wr, f.dt) for f in frames]
        if not first_lap:
            enc[0] = (turn.wl, turn.wr, turn.dt)
        ctx.stage_encoders([e[0] for e in enc], [e[1] for e in enc], [e[2] for e in enc])
        for i, f in enumerate(frames):
            z = observations(w, f, rng)
            ctx.inject_observations(i, f.ids, np.ones(len(f.ids), np.int32), z, np.tile([0.05, 0.05, 0.01], (len(f.ids), 1)))
    stage(True)
    ctx.run_staged(0, lap, with_ekf=2); ctx.sync()          # builds the map
    N = ctx.get_state()[0].size
    stage(False)
    ctx.run_staged(0, lap, with_ekf=2); ctx.sync()          # warm
    stage(False)
    t0 = time.perf_counter()
    ctx.run_staged(0, lap, with_ekf=2); ctx.sync()
    dt = time.perf_counter() - t0
    stage(False)
    ctx.profile_enable(True); ctx.profile_reset()
    ctx.run_staged(0, lap, with_ekf=2); ctx.sync()
    prof = ctx.profile_get(); ps = ctx.plan_stats()
    mu, S = ctx.get_state()
    err = np.abs(mu[3:].reshape(-1, 3)[:, :2] - (w.world[:, :2] - np.array(frames[0].true_pose[:2]))).max()
    print(f"{name}: N={N} lap {lap} frames: {dt / lap * 1e6:.1f} us/frame EKF-only ({lap / dt:.0f} fps); per frame us:",
          {k: round(v[1] / lap * 1e3, 2) for k, v in prof.items() if v[0]}, ps, f"map err {err:.3f} m sym {np.abs(S - S.T).max():.1e}", flush=True)
    ctx.close()


for nm in (sys.argv[1:] or ["cfg2", "cfg2_sliding", "cfg3"]):
    run(nm)
