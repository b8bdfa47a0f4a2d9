"""development aid (CPU emulation): border-node statistics of one frame - nodes per type, nodes per border cycle, segment lengths"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("ARUCO_SLAM_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "hipemu", "_build", "libaruco_slam_emu.so"))
import numpy as np
from aruco_slam_amd import capi, synth
name = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 64
cfg = synth.CONFIGS[name]; w = synth.make_world(cfg)
ctx = capi.Context(max_rows=cfg.rows, max_cols=cfg.cols, max_batch=1, max_landmarks=w.L + 8, persistent_waves=4)
ctx.set_camera(w.K, np.zeros(5)); synth.apply_detector(cfg, ctx)
fr = w.frame(3)
ctx.synth_render(0, cfg.rows, cfg.cols, w.K, fr.ids, fr.poses, noise_amp=2, seed=3)
ctx.run_staged(0, 1, False); ctx.sync()
DX = [1, 1, 0, -1, -1, -1, 0, 1]; DY = [0, -1, -1, -1, 0, 1, 1, 1]
def first_outer(m): return 0 if m & 1 else 7 if m & 128 else 6 if m & 64 else 5
def first_hole(m): return (int(m) & 0xFE).bit_length() - 1
for sc in range(3):
    M = ctx.debug_nbr(0, sc, cfg.rows, cfg.cols).astype(int)
    rows, cols = M.shape
    fg = np.zeros_like(M, bool)      # a pixel is foreground iff some neighbour sees it: W neighbour's bit 0
    fg[:, 1:] |= (M[:, :-1] & 1) > 0; fg[:, :-1] |= (M[:, 1:] & 16) > 0; fg[1:, :] |= (M[:-1, :] & 64) > 0; fg[:-1, :] |= (M[1:, :] & 4) > 0
    fg |= False
    def top_outer(x, y):
        m = M[y, x]
        for t in range(1, 9):
            if not m & 1: return True
            m = M[y, x + t]
            if m & 6: return False
        return True
    def top_hole(x, y):
        m = M[y, x]
        for t in range(1, 9):
            if m & 1: return True
            if x + t >= cols: return False
            m = M[y, x + t]
            if not m & 4: return False
        return True
    def step(x, y, s):
        m = M[y, x]
        for k in range(8):
            d = (s + 1 + k) & 7
            if m >> d & 1: return x + DX[d], y + DY[d], (d + 4) & 7
        raise RuntimeError
    nodes = {}
    ys, xs = np.nonzero(fg & (M != 0))
    for x, y in zip(xs, ys):
        m = M[y, x]
        outer = (m & 0x1E) == 0; hole = (m & 3) == 2
        grid = x % K == 0 or y % K == 0
        if outer or hole:
            s0 = first_outer(m) if outer else first_hole(m)
            top = top_outer(x, y) if outer else top_hole(x + 1, y)
            cutok = grid and not (m >> ((s0 + 1) & 7) & 1)
            if top or cutok: nodes[(x, y, s0)] = ("O" if outer else "H") + ("t" if top else "-")
            elif True: nodes.setdefault(("dead", x, y), "dead")
        if grid:
            for s in range(8):
                if (m >> s & 1) and not (m >> ((s + 1) & 7) & 1) and (x, y, s) not in nodes: nodes[(x, y, s)] = "C"
    live = {k: v for k, v in nodes.items() if v != "dead"}
    ndead = len(nodes) - len(live)
    # walk segments
    nxt = {}; seglen = {}
    for k0 in live:
        x, y, s = k0; n = 0
        while True:
            x, y, s = step(x, y, s); n += 1
            if (x, y, s) in live or n > 6000: break
        nxt[k0] = (x, y, s) if (x, y, s) in live else None; seglen[k0] = n
    # cycles
    seen = set(); cyc = []
    for k0 in live:
        if k0 in seen: continue
        c = []; k = k0
        while k is not None and k not in seen:
            seen.add(k); c.append(k); k = nxt[k]
        cyc.append((len(c), sum(seglen[q] for q in c)))
    cyc.sort(reverse=True)
    types = {}
    for v in live.values(): types[v] = types.get(v, 0) + 1
    sl = np.array(list(seglen.values()))
    print(f"scale {sc}: live nodes {len(live)} {types}, dead candidates {ndead}; segments: mean {sl.mean():.1f} p99 {np.percentile(sl, 99):.0f} max {sl.max()}; total steps {sl.sum()}; "
          f"largest cycles (nodes, points): {cyc[:6]}")
