"""Child process of tests/test_gpu_parity.py::test_library_first_then_torch_share_one_hip_runtime: the C-ABI library is loaded
and USED before torch is imported; both must then see the GPU (one HIP runtime in the process, whatever the import order)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
assert "torch" not in sys.modules
from aruco_slam_amd import capi, synth  # noqa: E402

cfg = synth.CONFIGS["cfg1"]
w = synth.PanelWorld(cfg)
n = 4
ctx = capi.Context(max_rows=cfg.rows, max_cols=cfg.cols, max_batch=n, max_landmarks=16)
ctx.set_camera(w.K, np.zeros(5))
frs = [w.frame(i) for i in range(n)]
for i, fr in enumerate(frs):
    ctx.synth_render(i, cfg.rows, cfg.cols, w.K, fr.ids, fr.poses, noise_amp=1, seed=i, download=False)
ctx.stage_encoders([f.wl for f in frs], [f.wr for f in frs], [f.dt for f in frs])
ctx.run_staged(0, n, with_ekf=True)
ctx.sync()
assert "torch" not in sys.modules

import torch  # noqa: E402
assert torch.cuda.is_available(), "torch lost the GPU: a second HIP runtime was mapped"
x = torch.arange(1024, device="cuda:0", dtype=torch.float32)
assert float((x * 2).sum().item()) == 1023 * 1024.0
from aruco_slam_amd.dist import MapGather, MAP_DTYPE  # noqa: E402
g = MapGather(ctx, device="cuda:0")
rec = np.frombuffer(g.gather().cpu().numpy().tobytes(), dtype=MAP_DTYPE)
ref = np.frombuffer(ctx.export_map().tobytes(), dtype=MAP_DTYPE)
assert np.array_equal(rec, ref) and (ref["id"] >= 0).sum() == len(ctx.get_landmark_ids()) > 0
maps = [ln for ln in open("/proc/self/maps") if "libamdhip64" in ln]
paths = sorted({ln.split()[-1] for ln in maps})
assert len(paths) == 1, paths
print("one hip runtime ok:", paths[0])
