"""EKF-chain-only timing (observations already in the slots): isolates the sequential part of the path."""
import sys, time
sys.path.insert(0, '.')
import numpy as np
from aruco_slam_amd import capi, synth
cfg = synth.CONFIGS["cfg2"]; w = synth.PanelWorld(cfg)
lap = w.lap_length()
ctx = capi.Context(max_rows=cfg.rows, max_cols=cfg.cols, max_batch=lap, max_landmarks=w.L+8)
ctx.set_camera(w.K, np.zeros(5))
frs=[w.frame(i) for i in range(lap)]
for i,f in enumerate(frs): ctx.synth_render(i, cfg.rows, cfg.cols, w.K, f.ids, f.poses, noise_amp=2, seed=i, download=False)
ctx.stage_encoders([f.wl for f in frs],[f.wr for f in frs],[f.dt for f in frs])
ctx.run_staged(0, lap, True); ctx.sync()
t=w.frame(lap); ctx.stage_encoders([t.wl],[t.wr],[t.dt], slot0=0)
for rep in range(2):
    t0=time.perf_counter(); ctx.run_staged(0, lap, 2); ctx.sync(); dt=time.perf_counter()-t0
    print("EKF only:", lap/dt, "fps", dt/lap*1e6, "us/frame")
t0=time.perf_counter(); ctx.run_staged(0, lap, 0); ctx.sync(); dt=time.perf_counter()-t0
print("detect only (batch %d):"%lap, lap/dt, "fps", dt/lap*1e6, "us/frame")
