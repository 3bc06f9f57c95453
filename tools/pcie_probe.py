#!/usr/bin/env python3
"""What the PCIe link gives: 512 MiB page-locked -> device, device -> page-locked, and both at once (torch copies on two
streams): the floor of the host-buffer entry point."""
import torch, time
n = 512 << 20
h = torch.empty(n, dtype=torch.uint8).pin_memory()
d = torch.empty(n, dtype=torch.uint8, device="cuda")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
h2 = torch.empty(n, dtype=torch.uint8).pin_memory()
d2 = torch.empty(n, dtype=torch.uint8, device="cuda")
def t(f, k=5):
    f(); torch.cuda.synchronize()
    b = 1e9
    for _ in range(k):
        t0 = time.perf_counter(); f(); torch.cuda.synchronize(); b = min(b, time.perf_counter() - t0)
    return b
a = t(lambda: d.copy_(h, non_blocking=True))
b = t(lambda: h2.copy_(d2, non_blocking=True))
def both():
    with torch.cuda.stream(s1): d.copy_(h, non_blocking=True)
    with torch.cuda.stream(s2): h2.copy_(d2, non_blocking=True)
c = t(both)
print(f"H2D {n/a/1e9:.1f} GB/s  D2H {n/b/1e9:.1f} GB/s  both at once: {n/c/1e9:.1f} GB/s each way ({c*1e3:.1f} ms for 512 MiB each way)")
import os
print("cpu affinity", len(os.sched_getaffinity(0)), "numa nodes:", os.listdir("/sys/devices/system/node") if os.path.exists("/sys/devices/system/node") else None)
