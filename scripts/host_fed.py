"""PCIe-inclusive rate of the host-fed stream API (pinned ring + asynchronous upload) at the headline config; reported in
DESIGN.md next to bench.py's HBM-resident `value`, never as `value`."""
import sys, time
sys.path.insert(0, '.')
import numpy as np
from aruco_slam_amd import capi, synth

cfg = synth.CONFIGS["cfg2"]; w = synth.PanelWorld(cfg)
lap = w.lap_length()
H = 100
ctx = capi.Context(max_rows=cfg.rows, max_cols=cfg.cols, max_batch=lap, max_landmarks=w.L + 8)
ctx.set_camera(w.K, np.zeros(5))
frs = [w.frame(i) for i in range(lap)]
imgs = [ctx.synth_render(i, cfg.rows, cfg.cols, w.K, f.ids, f.poses, noise_amp=2, seed=i) for i, f in enumerate(frs)]
ctx.stage_encoders([f.wl for f in frs], [f.wr for f in frs], [f.dt for f in frs])
ctx.run_staged(0, lap, True); ctx.sync()                      # map built
turn = w.frame(lap)
enc = [(turn.wl, turn.wr, turn.dt)] + [(f.wl, f.wr, f.dt) for f in frs[1:]]
ctx.stream_open(cfg.rows, cfg.cols, 1, H)
for mode in ("push (host memcpy into the pinned ring)", "acquire/commit (producer writes the pinned slot; here: left as is)"):
    for rep in range(2):
        t0 = time.perf_counter()
        for i in range(lap):
            if mode.startswith("push"):
                ctx.stream_push(imgs[i], *enc[i])
            else:
                ctx.stream_slot(cfg.rows, cfg.cols)
                ctx.stream_commit(*enc[i])
        ctx.stream_flush()
        dt = time.perf_counter() - t0
    print(f"{mode}: {lap / dt:.0f} frames/s, {lap * cfg.rows * cfg.cols / dt / 1e9:.2f} GB/s over PCIe")
mu, _ = ctx.get_state()
print("landmarks", (mu.size - 3) // 3)
