import sys, os
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import conftest  # sets ARUCO_SLAM_LIB to the emulation library when no GPU
from aruco_slam_amd import capi, synth
import parity_common as pc
cfg = synth.CONFIGS["cfg2"]; w = synth.PanelWorld(cfg)
fi = int(sys.argv[1]) if len(sys.argv) > 1 else 191
fr = w.frame(fi)
ctx = capi.Context(max_rows=cfg.rows, max_cols=cfg.cols, max_batch=1, max_landmarks=16, persistent_waves=8)
ctx.set_camera(w.K, np.zeros(5))
img = ctx.synth_render(0, cfg.rows, cfg.cols, w.K, fr.ids, fr.poses, noise_amp=2, seed=fi)
ctx.run_staged(0, 1, with_ekf=False); ctx.sync()
pc.check_stages(ctx, 0, img, expect_ids=fr.ids)
print("ok")
