#!/usr/bin/env python3
"""Where do page-locked host buffers live relative to the GPU?  Prints the NUMA node of every AMD GPU function on the
PCI bus, the node(s) of the pages of a hutk_host_alloc buffer (move_pages with no target = query), this process's CPU
affinity, and the host path's rate on this box: do the boxes whose host path gives 26 instead of 37 GB/s keep their
buffers on the far node?"""
import ctypes, glob, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
for d in sorted(glob.glob("/sys/bus/pci/devices/*")):
    try:
        if open(d + "/vendor").read().strip() == "0x1002" and open(d + "/class").read().startswith(("0x0302", "0x0380", "0x0300", "0x1200")):
            print(os.path.basename(d), "numa_node", open(d + "/numa_node").read().strip(), "class", open(d + "/class").read().strip())
    except OSError:
        pass
print("cpus allowed:", len(os.sched_getaffinity(0)), "first/last", min(os.sched_getaffinity(0)), max(os.sched_getaffinity(0)), "running on cpu", os.sched_getcpu() if hasattr(os, "sched_getcpu") else "?")
try:
    for n in sorted(glob.glob("/sys/devices/system/node/node*")):
        print(os.path.basename(n), "cpus", open(n + "/cpulist").read().strip())
except OSError:
    pass
from hutoken_amd import _capi
import torch
print("this process's GPU:", torch.cuda.get_device_properties(0).name, "pci", getattr(torch.cuda.get_device_properties(0), "pci_bus_id", "?"), getattr(torch.cuda.get_device_properties(0), "pci_device_id", "?"))
pa = _capi.PinnedArray(64 << 20, np.uint8)
pa.array[:] = 1
libc = ctypes.CDLL(None, use_errno=True)
PAGE = 4096
npg = 64
pages = (ctypes.c_void_p * npg)(*[pa.array.ctypes.data + i * (1 << 20) for i in range(npg)])
status = (ctypes.c_int * npg)()
rc = libc.syscall(279, 0, npg, pages, None, status, 0)  # move_pages(pid 0, count, pages, nodes = NULL: query, status, flags)
print("move_pages rc", rc, "nodes of 64 sample pages of a hutk_host_alloc buffer:", sorted(set(status)))
os.system(f"{sys.executable} tools/pcie_probe.py 2>/dev/null | head -1")
os.system(f"{sys.executable} tools/host_path_sweep.py 1000000 0 2>/dev/null | tail -1")
