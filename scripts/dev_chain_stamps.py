"""development aid: one lap of cfg2 through a -DASLAM_WIN_STAMPS build (ARUCO_SLAM_LIB), prints the chain kernel's phase cycles"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from aruco_slam_amd import capi, synth
cfg = synth.CONFIGS["cfg2"]; w = synth.PanelWorld(cfg); lap = w.lap_length()
ctx = capi.Context(max_rows=cfg.rows, max_cols=cfg.cols, max_batch=lap, max_landmarks=w.L + 8)
ctx.set_camera(w.K, np.zeros(5)); synth.apply_detector(cfg, ctx)
frs = [w.frame(i) for i in range(lap)]
for i, f in enumerate(frs):
    ctx.synth_render(i, cfg.rows, cfg.cols, w.K, f.ids, f.poses, noise_amp=2, seed=i, download=False)
ctx.stage_encoders([f.wl for f in frs], [f.wr for f in frs], [f.dt for f in frs])
ctx.run_staged(0, lap, True); ctx.sync()
print("map built", (ctx.get_state()[0].size - 3) // 3, flush=True)
ctx.run_staged(0, 64, True); ctx.sync()
