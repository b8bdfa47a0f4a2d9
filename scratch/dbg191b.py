import sys, os
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from aruco_slam_amd import capi, synth
import parity_common as pc
from oracle import pyoracle as orc
cfg = synth.CONFIGS["cfg2"]; w = synth.PanelWorld(cfg)
ctx = capi.Context(max_rows=cfg.rows, max_cols=cfg.cols, max_batch=24, max_landmarks=16)
ctx.set_camera(w.K, np.zeros(5))
f0 = 168
frs = [w.frame(f0 + i) for i in range(24)]
imgs = [ctx.synth_render(i, cfg.rows, cfg.cols, w.K, fr.ids, fr.poses, noise_amp=2, seed=f0 + i) for i, fr in enumerate(frs)]
for rep in range(3):
    ctx.run_staged(0, 24, with_ekf=False); ctx.sync()
    bad = []
    for i in range(24):
        ids_o, c_o = orc.detect(imgs[i])
        ids_g, c_g = ctx.get_slot_detections(i)[:2]
        if not (np.array_equal(ids_o, ids_g) and np.array_equal(c_o, c_g)): bad.append(i)
    print("rep", rep, "bad slots", bad)
    for i in bad[:2]:
        try:
            pc.check_stages(ctx, i, imgs[i], expect_ids=frs[i].ids)
        except AssertionError as e:
            print("  slot", i, "->", str(e)[:200])
