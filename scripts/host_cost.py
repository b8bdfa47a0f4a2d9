"""host-side cost of issuing the per-frame EKF chain (no sync inside the timed call)"""
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
for rep in range(3):
    t0=time.perf_counter(); ctx.run_staged(0, 200, 2); t1=time.perf_counter(); ctx.sync(); t2=time.perf_counter()
    print("EKF-only 200 frames: issue %.2f ms (%.1f us/frame), total %.2f ms (%.1f us/frame)" % ((t1-t0)*1e3, (t1-t0)/200*1e6, (t2-t0)*1e3, (t2-t0)/200*1e6))
for rep in range(2):
    t0=time.perf_counter(); ctx.run_staged(0, 200, 1); t1=time.perf_counter(); ctx.run_staged(200, 200, 1); t2=time.perf_counter(); ctx.sync(); t3=time.perf_counter()
    print("full 2x200: issue %.2f + %.2f ms, total %.2f ms (%.1f us/frame)" % ((t1-t0)*1e3, (t2-t1)*1e3, (t3-t0)*1e3, (t3-t0)/400*1e6))
