"""Randomised detector parity sweep on the GPU (development tool): many random scenes / sizes / noise levels / distortion, device
result against the oracle; prints every mismatch.  usage: python scripts/fuzz_detect.py [n_cases] [first_seed]"""
import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import torch  # noqa: F401  (before the library)
from aruco_slam_amd import capi, synth
from oracle import pyoracle as orc

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
bad = 0
over = 0
gated = 0
worst_all = 0.0
ctxs = {}
for seed in range(seed0, seed0 + n_cases):
    rng = np.random.RandomState(seed)
    rows = int(rng.choice([240, 360, 480, 601, 720, 1080]))
    cols = int(rng.choice([320, 487, 640, 853, 1280, 1920]))
    f = float(rng.uniform(0.5, 1.0)) * cols
    n = int(rng.randint(0, 13))
    ids, poses, K = synth.simple_scene(rows, cols, f, max(n, 1), seed=seed, tz=(0.8, 2.8), max_yaw_deg=50.0)
    if n == 0:
        ids, poses = ids[:0], poses[:0]
    D = np.zeros(5) if seed % 2 else np.array([rng.uniform(-0.2, 0.2), rng.uniform(-0.1, 0.1), rng.uniform(-0.005, 0.005), rng.uniform(-0.005, 0.005), 0.0])
    key = (rows, cols)
    if key not in ctxs:
        ctxs[key] = capi.Context(max_rows=rows, max_cols=cols, max_batch=1, max_landmarks=16)
    ctx = ctxs[key]
    ctx.set_camera(K, D)
    gray = ctx.synth_render(0, rows, cols, K, ids, poses, noise_amp=int(rng.randint(0, 12)), seed=seed, background=int(rng.randint(40, 256)),
                            supersample=int(rng.choice([1, 2, 4])))
    ctx.run_staged(0, 1, with_ekf=False)
    try:
        ctx.sync()
    except capi.AslamError as e:
        over += 1
        print("capacity reported, seed", seed, rows, cols, str(e)[:60])
        continue
    ids_o, c_o = orc.detect(gray)
    ids_g, c_g, rv, tv = ctx.get_slot_detections(0)
    ok = np.array_equal(ids_o, ids_g) and np.array_equal(c_o, c_g)
    worst = 0.0
    if ok and len(ids_g):
        # poses: only markers that pass the range / covariance gates ever reach the filter; a marker whose 20 LM iterations do not
        # converge (degenerate geometry) is chaotic in both implementations and is dropped by the covariance gate
        _, valid, _, _ = ctx.get_slot_raw_observations(0)
        for i in range(len(ids_g)):
            if not valid[i]:
                gated += 1
                continue
            r_o, t_o, _ = orc.solve_pnp(c_g[i], 0.27, K, D)
            worst = max(worst, float(np.abs(r_o - rv[i]).max() / max(np.abs(r_o).max(), 1e-12)), float(np.abs(t_o - tv[i]).max() / np.abs(t_o).max()))
        ok = worst <= 1e-6
        worst_all = max(worst_all, worst)
    if not ok:
        bad += 1
        print("MISMATCH seed", seed, rows, cols, "ids/corners equal:", np.array_equal(ids_o, ids_g) and np.array_equal(c_o, c_g), "worst relative pose difference:", worst)
print(f"{n_cases} cases, {bad} mismatches, {over} reported capacity overflows, {gated} gated markers skipped, worst relative pose difference of a gate-passing marker {worst_all:.2e}")
