import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
from aruco_slam_amd import capi, synth
cfg = synth.CONFIGS["cfg2"]; w = synth.make_world(cfg); n = 60
ctx = capi.Context(max_rows=cfg.rows, max_cols=cfg.cols, max_batch=1, max_landmarks=w.L + 8)
ctx.set_camera(w.K, np.zeros(5))
frs = [w.frame(i) for i in range(n)]
imgs = [ctx.synth_render(0, cfg.rows, cfg.cols, w.K, f.ids, f.poses, noise_amp=2, seed=i) for i, f in enumerate(frs)]
bgr = [np.ascontiguousarray(np.repeat(g[:, :, None], 3, axis=2)) for g in imgs]
ctx.add_encoder(0.0, 0.0, 0.0)
t_now = 0.0; lat = []; parts = []
for rep in range(3):
  for i in range(n):
    t_now += frs[i].dt
    ctx.add_encoder(frs[i].wl, frs[i].wr, t_now)
    t0 = time.perf_counter(); ctx.add_image(bgr[i]); lat.append(time.perf_counter() - t0); parts.append(ctx.last_timing())
a = np.array(lat[10:]) * 1e6
print("add_image p50 %.1f p99 %.1f us; phases p50:" % (np.percentile(a, 50), np.percentile(a, 99)), {k: round(float(np.percentile([p[k] for p in parts[10:]], 50)), 1) for k in parts[0]})
