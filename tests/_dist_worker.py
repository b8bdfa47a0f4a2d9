"""world_size-2 worker for tests/test_dist_gloo.py (launched with torch.distributed.run, gloo backend, CPU)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from aruco_slam_amd import capi, synth  # noqa: E402
from aruco_slam_amd.dist import MapGather, rank_info, stream_for_rank  # noqa: E402


def main():
    rank, local_rank, world = rank_info()
    dist.init_process_group(backend="gloo")
    assert stream_for_rank(4, rank, world) == [s for s in range(4) if s % world == rank]
    cfg = synth.SceneConfig(rows=240, cols=320, f=225.0, grid=(2, 2), n_panels=3, col_spacing=0.9, row_spacing=0.7, step=0.05,
                            tz_far=2.4, tz_near=1.9, seed=1 + rank)               # a different stream per rank
    w = synth.PanelWorld(cfg)
    n = 3 + 2 * rank
    ctx = capi.Context(max_rows=cfg.rows, max_cols=cfg.cols, max_batch=n, persistent_waves=4, max_landmarks=16)
    ctx.set_camera(w.K, np.zeros(5))
    frs = [w.frame(i) for i in range(n)]
    for i, f in enumerate(frs):
        ctx.synth_render(i, cfg.rows, cfg.cols, w.K, f.ids, f.poses, noise_amp=1, seed=10 * rank + i, download=False)
    ctx.stage_encoders([f.wl for f in frs], [f.wr for f in frs], [f.dt for f in frs])
    ctx.run_staged(0, n, with_ekf=True)
    ctx.sync()
    g = MapGather(ctx)
    g.gather()
    rec = g.records()
    assert rec.shape == (world, 16)
    mine = np.frombuffer(ctx.export_map().tobytes(), dtype=rec.dtype)
    assert np.array_equal(rec[rank], mine)
    mu, S = ctx.get_state()
    L = (mu.size - 3) // 3
    assert L > 0 and (rec[rank]["id"][:L] >= 0).all() and (rec[rank]["id"][L:] == -1).all()
    # every rank sees every other rank's map, bit for bit
    counts = torch.tensor([int((rec[r]["id"] >= 0).sum()) for r in range(world)])
    ref = counts.clone()
    dist.broadcast(ref, src=0)
    assert torch.equal(ref, counts)
    # max-over-ranks timing reduction used by bench.py
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    assert t.item() == world
    dist.barrier()
    dist.destroy_process_group()
    print(f"rank {rank} ok: {L} landmarks, gathered {counts.tolist()}")


if __name__ == "__main__":
    main()
