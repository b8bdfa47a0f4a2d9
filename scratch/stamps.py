import sys, time, ctypes
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
out=(ctypes.c_ulonglong*16)()
for n in (1,2,3,50):
    ctx.run_staged(0, n, 2); ctx.sync()
    capi.load().aslam_debug_get_stamps(ctx.h, out)
    s=list(out)
    print(n, [ (s[i+1]-s[i])*0.01 for i in range(4)], "us (100MHz ticks)")
ctx.run_staged(0, 50, 2); ctx.sync()
capi.load().aslam_debug_get_stamps(ctx.h, out)
s=list(out)
print('cycles: publish %d, barrier %d, Sread %d, inv3 %d, ops+mfma %d' % (s[9]-s[8], s[10]-s[9], s[11]-s[10], s[12]-s[11], s[13]-s[12]))
