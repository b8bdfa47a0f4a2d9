"""Headroom check: S independent camera streams (S contexts, each with its own detection / EKF streams) on ONE GPU.
The headline metric is one stream per GPU; this only shows how far the sequential EKF chain leaves the device idle."""
import sys, time
sys.path.insert(0, '.')
import numpy as np
from aruco_slam_amd import capi, synth

S = int(sys.argv[1]) if len(sys.argv) > 1 else 4
B = 200
cfg = synth.CONFIGS["cfg2"]; w = synth.PanelWorld(cfg)
lap = w.lap_length()
frs = [w.frame(i) for i in range(lap)]
turn = w.frame(lap)
ctxs = []
for s in range(S):
    c = capi.Context(max_rows=cfg.rows, max_cols=cfg.cols, max_batch=lap, max_landmarks=w.L + 8)
    c.set_camera(w.K, np.zeros(5))
    for i, f in enumerate(frs):
        c.synth_render(i, cfg.rows, cfg.cols, w.K, f.ids, f.poses, noise_amp=2, seed=1000 * s + i, download=False)
    c.stage_encoders([f.wl for f in frs], [f.wr for f in frs], [f.dt for f in frs])
    c.run_staged(0, lap, True); c.sync()
    c.stage_encoders([turn.wl], [turn.wr], [turn.dt], slot0=0)
    ctxs.append(c)
def step(pos):
    for c in ctxs:
        c.run_staged(pos, B, True)
for rep in range(3):
    for c in ctxs: c.sync()
    t0 = time.perf_counter()
    pos = 0
    for k in range(10):
        step(pos); pos = (pos + B) % lap
    for c in ctxs: c.sync()
    dt = time.perf_counter() - t0
    print(f"{S} streams: {S * 10 * B / dt:.0f} frames/s aggregate, {10 * B / dt:.0f} per stream")
for c in ctxs:
    mu, _ = c.get_state()
    assert (mu.size - 3) // 3 == w.L
