"""BASELINE config 5: batches of 64 independent 640x480 frames, 4 markers each, detection + pose only (no EKF).
One context = one in-order chain of latency-bound kernels per 64-frame step; independent batches can go to several contexts
(each with its own streams and work lists), which overlaps those chains."""
import sys, time
sys.path.insert(0, '.')
import numpy as np
from aruco_slam_amd import capi, synth

rows, cols, f, n = 480, 640, 450.0, 64
S = int(sys.argv[1]) if len(sys.argv) > 1 else 1
ids0, poses0, K = synth.simple_scene(rows, cols, f, 4, seed=0, tz=(1.0, 2.0))
ctxs = []
for s in range(S):
    ctx = capi.Context(max_rows=rows, max_cols=cols, max_batch=2 * n, max_landmarks=16)
    ctx.set_camera(K, np.zeros(5))
    for i in range(2 * n):
        ids, poses, _ = synth.simple_scene(rows, cols, f, 4, seed=i, tz=(1.0, 2.0))
        ctx.synth_render(i, rows, cols, K, ids, poses, noise_amp=2, seed=1000 * s + i, download=False)
    ctxs.append(ctx)
for rep in range(3):
    for c in ctxs: c.sync()
    t0 = time.perf_counter()
    steps = 50
    for st in range(steps):
        for c in ctxs:
            c.run_staged((st & 1) * n, n, with_ekf=False)
    for c in ctxs: c.sync()
    dt = time.perf_counter() - t0
    print(f"cfg5, {S} context(s): {S * steps * n / dt:.0f} frames/s ({dt / steps * 1e3:.3f} ms per round of {S} 64-frame steps, "
          f"{S * steps * n * rows * cols / dt / 1e9:.1f} GB/s of input pixels)")
tot = sum(len(ctxs[0].get_slot_detections(i)[0]) for i in range(2 * n))
print("markers found", tot, "of", 2 * n * 4)
