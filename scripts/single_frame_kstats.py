"""development aid: the single-frame drop-in call (aslam_add_encoder + aslam_add_image) 40 times on cfg2 frames - run it under
rocprofv3 --kernel-trace --stats to see the batch-1 duration of every kernel (scripts/gpu_single_kstats.sh)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from aruco_slam_amd import capi, synth
cfg = synth.CONFIGS["cfg2"]; w = synth.make_world(cfg); n = 40
ctx = capi.Context(max_rows=cfg.rows, max_cols=cfg.cols, max_batch=1, max_landmarks=w.L + 8)
ctx.set_camera(w.K, np.zeros(5))      # default DetectorParameters, as the reference node runs
frs = [w.frame(i) for i in range(n)]
imgs = [ctx.synth_render(0, cfg.rows, cfg.cols, w.K, f.ids, f.poses, noise_amp=2, seed=i) for i, f in enumerate(frs)]
bgr = [np.ascontiguousarray(np.repeat(g[:, :, None], 3, axis=2)) for g in imgs]
ctx.add_encoder(0.0, 0.0, 0.0)
t_now = 0.0; lat = []
for i in range(n):
    t_now += frs[i].dt
    ctx.add_encoder(frs[i].wl, frs[i].wr, t_now)
    t0 = time.perf_counter(); ctx.add_image(bgr[i]); lat.append(time.perf_counter() - t0)
print("add_image p50 %.1f us" % (np.percentile(np.array(lat[5:]) * 1e6, 50)))
