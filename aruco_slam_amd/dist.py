"""One camera stream per GPU: each rank owns an independent `ArucoSlam` stream — its own EKF
state, its own frames — so the data path needs no collective.  The only exchange is the landmark-map gather:
every rank contributes a fixed-size block of MAP_RECORD_BYTES-byte records (id, index, x, y, theta, Sigma_ll) and
receives everyone's.  On GPUs this is one RCCL all-gather over xGMI straight from the library's device buffer
(torch.distributed backend "nccl"); the CPU tests use "gloo".  The gathered maps are read-only: nothing is
fused back into a stream's filter, so per-stream results stay identical to the single-stream reference.
"""
import os

import numpy as np

from . import capi

MAP_DTYPE = np.dtype([("id", "<i4"), ("index", "<i4"), ("x", "<f8"), ("y", "<f8"), ("theta", "<f8"), ("S", "<f8", (9,))])
assert MAP_DTYPE.itemsize == capi.MAP_RECORD_BYTES


def rank_info():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def stream_for_rank(n_streams, rank, world):
    """streams s with s % world == rank (stream -> GPU `s mod G`)"""
    return [s for s in range(n_streams) if s % world == rank]


class MapGather:
    """all-gather of the landmark-map records of one context per rank"""

    def __init__(self, ctx, device=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.ctx = torch, dist, ctx
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.nbytes = int(ctx.init.max_landmarks) * capi.MAP_RECORD_BYTES
        self.on_gpu = device is not None and str(device).startswith("cuda")
        dev = device if self.on_gpu else "cpu"
        self.mine = torch.zeros(self.nbytes, dtype=torch.uint8, device=dev)
        self.all = torch.zeros(self.nbytes * self.world, dtype=torch.uint8, device=dev)

    def gather(self):
        if self.on_gpu:
            self.ctx.export_map_to_device(self.mine.data_ptr())          # device -> device, no host hop
        else:
            self.mine.copy_(self.torch.from_numpy(self.ctx.export_map()))
        if self.world > 1:
            self.dist.all_gather_into_tensor(self.all, self.mine)
        else:
            self.all.copy_(self.mine)
        return self.all

    # -- pipelined variant (GPU only): export step k behind its EKF chain, gather step k-1 meanwhile ----------------
    def gather_pipelined(self):
        """Enqueue the export of the map as of the EKF steps submitted so far and all-gather the PREVIOUS call's export, so
        the host never waits for the step it has just submitted (the stream keeps its detection / EKF overlap).  The result
        returned lags one call behind; flush() delivers the last one."""
        assert self.on_gpu
        if not hasattr(self, "_bufs"):
            self._bufs = [self.mine, self.torch.zeros_like(self.mine)]
            self._k = 0
            self._work = None
        cur = self._k & 1
        self._finish_collective()                               # the one enqueued a whole step ago: done by now
        self.ctx.export_map_async(self._bufs[cur].data_ptr(), cur)
        if self._k > 0:
            self._gather_buffer(cur ^ 1)
        self._k += 1
        return self.all

    def _gather_buffer(self, b):
        self.ctx.export_wait(b)
        if self.dist.is_initialized():                           # also with a single rank: the same RCCL path
            self._work = self.dist.all_gather_into_tensor(self.all, self._bufs[b], async_op=True)
        else:
            self.all.copy_(self._bufs[b])

    def _finish_collective(self):
        # The library writes the export buffers on its own HIP stream, which torch knows nothing about: before a buffer is
        # reused the collective that read it must have completed on the device, not merely be ordered on torch's stream.
        if self._work is not None:
            self._work.wait()
            self.torch.cuda.current_stream().synchronize()
            self._work = None

    def flush(self):
        if getattr(self, "_k", 0) > 0:
            self._finish_collective()
            self._gather_buffer((self._k - 1) & 1)
            self._finish_collective()
            self.torch.cuda.current_stream().synchronize()
        return self.all

    def records(self):
        """gathered maps as a (world, max_landmarks) structured array (host copy)"""
        buf = self.all.cpu().numpy().tobytes()
        return np.frombuffer(buf, dtype=MAP_DTYPE).reshape(self.world, -1)
