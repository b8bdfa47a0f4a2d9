import sys, os
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
os.environ.setdefault("ARUCO_SLAM_LIB", os.getcwd() + "/tests/hipemu/_build/libaruco_slam_emu.so")
import numpy as np
from aruco_slam_amd import capi, synth
from oracle import pyoracle as orc
bits, maxcorr = synth.random_dictionary(6, 40, 11, seed=3)
orc.set_dictionary(bits, maxcorr)
rows, cols, f = 240, 320, 300.0
ids, poses, K = synth.simple_scene(rows, cols, f, 3, seed=4, tz=(0.9, 1.3))
ids = np.arange(3, dtype=np.int32) + 5
ctx = capi.Context(max_rows=rows, max_cols=cols, max_batch=1, persistent_waves=4, max_landmarks=16)
ctx.set_camera(K, np.zeros(5)); ctx.set_dictionary(bits, maxcorr)
img = ctx.synth_render(0, rows, cols, K, ids, poses, noise_amp=2, seed=2)
ids_o, c_o = orc.detect(img)
print("expected", ids, "oracle", ids_o)
co, so, _, _ = orc.candidates(img, 2)
print("final candidates", len(co))
for c in co:
    ok, idv, cc = orc.identify(img, c)
    print(ok, idv, c.reshape(-1)[:4])
np.save('/tmp/img.npy', img)
print(poses[:, 9:])
c0, s0, _, _ = orc.candidates(img, 0)
print("stage0 quads", len(c0))
for c, s in zip(c0, s0):
    xs = c.reshape(4, 2)
    if xs[:, 0].min() > 40 and xs[:, 0].max() < 130 and xs[:, 1].min() > 25 and xs[:, 1].max() < 115:
        print(s, xs.reshape(-1))
for k in (3, 13, 23):
    th = orc.threshold(img, k)
    sizes, keys, hole, pts = orc.find_contours(th)
    big = [(int(s), int(kk) % 320, int(kk) // 320) for s, kk in zip(sizes, keys) if s > 150]
    print(k, big)
