import sys, time, ctypes
sys.path.insert(0, '.')
import numpy as np
from aruco_slam_amd import capi, synth
cfg = synth.CONFIGS["cfg2"]; w = synth.PanelWorld(cfg)
lap = 200
ctx = capi.Context(max_rows=cfg.rows, max_cols=cfg.cols, max_batch=lap, max_landmarks=w.L+8)
ctx.set_camera(w.K, np.zeros(5))
frs=[w.frame(i) for i in range(lap)]
for i,f in enumerate(frs): ctx.synth_render(i, cfg.rows, cfg.cols, w.K, f.ids, f.poses, noise_amp=2, seed=i, download=False)
out=(ctypes.c_uint*8)()
lib=capi.load()
lib.aslam_debug_get_counters(ctx.h, out); a=list(out)
t0=time.perf_counter(); ctx.run_staged(0, lap, 0); ctx.sync(); dt=time.perf_counter()-t0
lib.aslam_debug_get_counters(ctx.h, out); b=list(out)
it=b[5]-a[5]; st=(b[6]-a[6])*16; wr=(b[7]-a[7])*16
print("frames", lap, "time/frame us", dt/lap*1e6)
print("wave-iterations", it, "lane-steps", st, "write-steps", wr, "steps/frame", st/lap, "lane utilisation", st/(it*64.0))
n=ctx.debug_get_contours(0,0,4000,400000) if hasattr(ctx,'debug_get_contours') else None
