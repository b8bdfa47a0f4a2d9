"""Pinned host -> device copy bandwidth of the box (hipMemcpyAsync through torch), next to which the PCIe-inclusive rate of the
host-fed stream (bench.py extra.host_fed_stream) is to be read.  usage: python scripts/microbench/pinned_copy.py > profiles/<tag>_pinned_copy.txt"""
import time
import torch

assert torch.cuda.is_available()
for nbytes, label in ((921600, "one 1280x720 gray frame"), (2764800, "one 1280x720 bgr8 frame"), (92160000, "100 gray frames (one submit of the ring)")):
    h = torch.empty(nbytes, dtype=torch.uint8).pin_memory()
    d = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    for _ in range(3):
        d.copy_(h, non_blocking=True)
    torch.cuda.synchronize()
    reps = max(5, int(2e9 // nbytes))
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        d.copy_(h, non_blocking=True)
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / reps
    print(f"{label}: {nbytes} B pinned H2D {ms * 1e3:.1f} us per copy = {nbytes / ms / 1e6:.2f} GB/s")
h = torch.empty(2764800, dtype=torch.uint8)          # pageable, as aslam_add_image receives it
d = torch.empty(2764800, dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50):
    d.copy_(h)
torch.cuda.synchronize()
print(f"one bgr8 frame from PAGEABLE memory (blocking): {(time.perf_counter() - t0) / 50 * 1e6:.1f} us per copy")
