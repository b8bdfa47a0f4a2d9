import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import os
os.environ.setdefault("ARUCO_SLAM_LIB", os.getcwd() + "/tests/hipemu/_build/libaruco_slam_emu.so")
from aruco_slam_amd import capi, synth
from oracle import pyoracle as orc
cfg = synth.CONFIGS["cfg2"]; w = synth.PanelWorld(cfg)
fr = w.frame(5)
ctx = capi.Context(max_rows=cfg.rows, max_cols=cfg.cols, max_batch=1, max_landmarks=16, persistent_waves=8)
ctx.set_camera(w.K, np.zeros(5))
img = ctx.synth_render(0, cfg.rows, cfg.cols, w.K, fr.ids, fr.poses, noise_amp=2, seed=5)
lo, hi = int(0.03 * 1280), int(4.0 * 1280)
for k in (3, 13, 23):
    th = orc.threshold(img, k)
    sizes, keys, hole, pts = orc.find_contours(th)
    sel = (sizes >= lo) & (sizes <= hi)
    print("win", k, "fg px", int((th > 0).sum()), "contours", len(sizes), "points all", int(sizes.sum()), "kept", int(sel.sum()), "points kept", int(sizes[sel].sum()),
          "small(<38)", int((sizes < lo).sum()), "pts small", int(sizes[sizes < lo].sum()), "holes", int(hole.sum()))

def shift(a, dy, dx):
    out = np.zeros_like(a)
    H, W = a.shape
    ys = slice(max(dy,0), H+min(dy,0)); xs = slice(max(dx,0), W+min(dx,0))
    yd = slice(max(-dy,0), H+min(-dy,0)); xd = slice(max(-dx,0), W+min(-dx,0))
    out[yd, xd] = a[ys, xs]
    return out
for k in (3, 13, 23):
    fg = orc.threshold(img, k) > 0
    fg[0,:]=fg[-1,:]=False; fg[:,0]=fg[:,-1]=False
    W_ = shift(fg,0,-1); NW = shift(fg,-1,-1); N = shift(fg,-1,0); NE = shift(fg,-1,1)
    anyn = np.zeros_like(fg)
    for dy in (-1,0,1):
        for dx in (-1,0,1):
            if dy or dx: anyn |= shift(fg,dy,dx)
    outer = fg & ~W_ & ~NW & ~N & ~NE & anyn
    hole = ~fg & W_ & N
    print("win", k, "outer weak", int(outer.sum()), "hole weak", int(hole.sum()))
    # strict with lookahead 8
    def strict_outer(y,x):
        for t in range(1,9):
            if not fg[y,x+t-1+1-1+0] : pass
        return True
    ys,xs = np.nonzero(outer); so=0
    for y,x in zip(ys,xs):
        ok=True; t=1
        while t<=8:
            if not fg[y,x+t]: break
            if N[y,x+t] or NE[y,x+t]: ok=False; break
            t+=1
        so+=ok
    ys,xs = np.nonzero(hole); sh=0
    for y,x in zip(ys,xs):
        ok=True; t=1
        while t<=8:
            if fg[y,x+t]: break
            if x+t>=fg.shape[1]: ok=False;break
            if not N[y,x+t]: ok=False; break
            t+=1
        sh+=ok
    print("      strict outer", so, "strict hole", sh)
