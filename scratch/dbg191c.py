import sys, os
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from aruco_slam_amd import capi, synth
import parity_common as pc
from oracle import pyoracle as orc
cfg = synth.CONFIGS["cfg2"]; w = synth.PanelWorld(cfg)
batch = 24
for rep in range(2):
    ctx = capi.Context(max_rows=cfg.rows, max_cols=cfg.cols, max_batch=batch, max_landmarks=w.L + 8)
    ctx.set_camera(w.K, np.zeros(5))
    nbad = 0
    for f0 in range(0, 240, batch):
        frs = [w.frame(f0 + i) for i in range(batch)]
        imgs = [ctx.synth_render(i, cfg.rows, cfg.cols, w.K, fr.ids, fr.poses, noise_amp=2, seed=f0 + i) for i, fr in enumerate(frs)]
        ctx.stage_encoders([fr.wl for fr in frs], [fr.wr for fr in frs], [fr.dt for fr in frs])
        ctx.run_staged(0, batch, with_ekf=True); ctx.sync()
        for i in range(batch):
            ids_o, c_o = orc.detect(imgs[i])
            ids_g, c_g = ctx.get_slot_detections(i)[:2]
            if not (np.array_equal(ids_o, ids_g) and np.array_equal(c_o, c_g)):
                nbad += 1
                print("rep", rep, "frame", f0 + i, "slot", i, "oracle n", len(ids_o), "gpu n", len(ids_g))
                print("   only oracle:", sorted(set(ids_o.tolist()) - set(ids_g.tolist())), "only gpu:", sorted(set(ids_g.tolist()) - set(ids_o.tolist())))
                try:
                    pc.check_stages(ctx, i, imgs[i])
                except AssertionError as e:
                    print("   stage:", str(e)[:160])
    print("rep", rep, "bad", nbad)
