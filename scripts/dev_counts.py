"""development aid: work-list sizes of one cfg2 lap (start candidates, contours, quad candidates, identify work)"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from aruco_slam_amd import capi, synth
name = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
cfg = synth.CONFIGS[name]; w = synth.make_world(cfg); n = int(sys.argv[2]) if len(sys.argv) > 2 else (64 if cfg.rows <= 720 else 24)
ctx = capi.Context(max_rows=cfg.rows, max_cols=cfg.cols, max_batch=n, max_landmarks=w.L + 8)
ctx.set_camera(w.K, np.zeros(5)); synth.apply_detector(cfg, ctx)
frs = [w.frame(i) for i in range(n)]
for i, f in enumerate(frs):
    ctx.synth_render(i, cfg.rows, cfg.cols, w.K, f.ids, f.poses, noise_amp=2, seed=i, download=False)
ctx.run_staged(0, n, False); ctx.sync()
out = (C.c_uint * 8)()
ctx.lib.aslam_debug_get_counters.argtypes = [C.c_void_p, C.POINTER(C.c_uint)]
ctx.lib.aslam_debug_get_counters(ctx.h, out)
print("frames", n, "q_trace(start tickets)", out[0], "q_quads", out[1], "n_ident", out[2], "q_ident", out[3], "q_write", out[4])
print("per frame: starts %.0f, ident work %.1f" % (out[0] / n, out[2] / n))
fc = [ctx.debug_frame_counts(i) for i in range(n)]
for k in fc[0]:
    v = np.array([f[k] for f in fc]); print(f"per frame {k}: mean {v.mean():.0f} min {v.min()} max {v.max()}")
