"""Ad-hoc GPU bring-up check: stage parity vs the oracle at full size + first timings."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from aruco_slam_amd import capi, synth
from oracle import pyoracle as orc

out = {}
def log(*a):
    print(*a, flush=True)

# ---- 1. cfg2 frame: stage parity -------------------------------------------------------------
cfg = synth.CONFIGS["cfg2"]
w = synth.PanelWorld(cfg)
B = 16
ctx = capi.Context(max_rows=cfg.rows, max_cols=cfg.cols, max_batch=B, max_landmarks=256)
D = np.zeros(5)
ctx.set_camera(w.K, D)
frames = [w.frame(i * 7) for i in range(B)]
imgs = []
t = time.time()
for i, fr in enumerate(frames):
    imgs.append(ctx.synth_render(i, cfg.rows, cfg.cols, w.K, fr.ids, fr.poses, noise_amp=2, seed=i))
log("render", B, "frames", time.time() - t)
t = time.time(); ctx.run_staged(0, B, with_ekf=False); ctx.sync(); log("first detect batch (incl. module load)", time.time() - t)
ok_all = True
for fi in (0, 5, 15):
    img = imgs[fi]
    for s, k in enumerate((3, 13, 23)):
        th = orc.threshold(img, k)
        sizes, keys, hole, pts = orc.find_contours(th)
        sel = (sizes >= 38) & (sizes <= 5120)
        gs, gk, gp = ctx.debug_contours(fi, s)
        offs = np.concatenate([[0], np.cumsum(sizes)])
        opts = np.concatenate([pts[offs[i]:offs[i+1]] for i in np.nonzero(sel)[0]]) if sel.any() else np.zeros((0, 2), np.int32)
        ok = np.array_equal(sizes[sel], gs) and np.array_equal(keys[sel], gk) and np.array_equal(opts, gp)
        ok_all &= ok
        log("frame", fi, "scale", s, "contours", int(sel.sum()), len(gs), "points", len(opts), "ok", ok)
    t = time.time(); ids_o, c_o = orc.detect(img); dt_o = time.time() - t
    ids_g, c_g, rv_g, tv_g = ctx.get_slot_detections(fi)
    ok = np.array_equal(ids_o, ids_g) and np.array_equal(c_o, c_g)
    ok_all &= ok
    perr = 0.0
    for j in range(len(ids_o)):
        rv, tv, it = orc.solve_pnp(c_o[j], 0.27, w.K, D)
        perr = max(perr, np.abs(rv - rv_g[j]).max(), np.abs(tv - tv_g[j]).max())
    log("frame", fi, "detections", len(ids_o), len(ids_g), "ok", ok, "expected", len(frames[fi].ids), "oracle detect s", round(dt_o, 3), "pose err", perr)
out["stage_parity"] = bool(ok_all)

# ---- 2. timing ---------------------------------------------------------------------------------
ctx.profile_enable(True); ctx.profile_reset()
for rep in range(5):
    ctx.run_staged(0, B, with_ekf=False)
ctx.sync()
prof = ctx.profile_get()
log("profile detect x5 batches of", B, json.dumps(prof))
ctx.profile_enable(False)
t = time.time()
for rep in range(10):
    ctx.run_staged(0, B, with_ekf=False)
ctx.sync()
dt = time.time() - t
log("detect-only fps", 10 * B / dt)
out["detect_fps"] = 10 * B / dt

# ---- 3. EKF sequence vs oracle (rank-3 form) ----------------------------------------------------
o = orc.Slam(literal=False); o.set_camera(w.K, D)
ctx2 = capi.Context(max_rows=cfg.rows, max_cols=cfg.cols, max_batch=B, max_landmarks=256)
ctx2.set_camera(w.K, D)
t_now = 0.0
nfr = 48
worst = 0.0
for f0 in range(0, nfr, B):
    fr = [w.frame(f0 + i) for i in range(B)]
    im = [ctx2.synth_render(i, cfg.rows, cfg.cols, w.K, f.ids, f.poses, noise_amp=2, seed=f0 + i) for i, f in enumerate(fr)]
    ctx2.stage_encoders([f.wl for f in fr], [f.wr for f in fr], [f.dt for f in fr])
    ctx2.run_staged(0, B, with_ekf=True); ctx2.sync()
    for i, f in enumerate(fr):
        t_now += f.dt
        o.add_encoder(f.wl, f.wr, t_now); o.add_image(im[i])
    mu_o, S_o = o.get_state(); mu_g, S_g = ctx2.get_state()
    same = mu_o.shape == mu_g.shape
    emu = np.abs(mu_o - mu_g).max() if same else -1
    eS = np.abs(S_o - S_g).max() / np.abs(S_o).max() if same else -1
    worst = max(worst, emu, eS) if same else 1e9
    gi, gx, ga, gz, gR = ctx2.get_observations()
    log("after frame", f0 + B - 1, "N", len(mu_o), len(mu_g), "mu_err", emu, "S_err", eS, "last n_obs", len(gi), "actions", np.bincount(ga, minlength=3).tolist())
out["ekf_worst"] = float(worst)
ctx2.profile_enable(True); ctx2.profile_reset()
t = time.time()
for rep in range(3):
    ctx2.run_staged(0, B, with_ekf=True)
ctx2.sync()
dt = time.time() - t
log("full pipeline fps (profiled)", 3 * B / dt, json.dumps(ctx2.profile_get()))
ctx2.profile_enable(False)
t = time.time()
for rep in range(10):
    ctx2.run_staged(0, B, with_ekf=True)
ctx2.sync()
dt = time.time() - t
log("full pipeline fps", 10 * B / dt)
out["full_fps"] = 10 * B / dt
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/gpu_check.json", "w"))
log("RESULT", json.dumps(out))
