"""Child process of tests/test_gpu_parity.py::test_pipelined_map_gather_over_rccl: one rank, backend nccl (= RCCL)."""
import os
import sys

import torch
import torch.distributed as dist
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aruco_slam_amd import capi, synth  # noqa: E402
from aruco_slam_amd.dist import MapGather, MAP_DTYPE  # noqa: E402

torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", device_id=torch.device("cuda", 0))
cfg = synth.CONFIGS["cfg1"]
w = synth.PanelWorld(cfg)
n = 8
ctx = capi.Context(max_rows=cfg.rows, max_cols=cfg.cols, max_batch=n, max_landmarks=16)
ctx.set_camera(w.K, np.zeros(5))
frs = [w.frame(i) for i in range(n)]
for i, fr in enumerate(frs):
    ctx.synth_render(i, cfg.rows, cfg.cols, w.K, fr.ids, fr.poses, noise_amp=1, seed=i, download=False)
ctx.stage_encoders([f.wl for f in frs], [f.wr for f in frs], [f.dt for f in frs])
g = MapGather(ctx, device="cuda:0")
for first in (0, 2, 4, 6):
    ctx.run_staged(first, 2, with_ekf=True)
    g.gather_pipelined()
g.flush()
ctx.sync()
final = np.frombuffer(ctx.export_map().tobytes(), dtype=MAP_DTYPE)
assert np.array_equal(g.records()[0], final), "gathered map differs from the blocking export"
assert (final["id"] >= 0).sum() == len(ctx.get_landmark_ids()) > 0
dist.destroy_process_group()
print("nccl gather ok")
