#!/usr/bin/env python3
"""Can the tile kernel read its input straight from page-locked HOST memory (no copy up)?  Times the device-resident
pipeline with d_bytes in HBM, then with d_bytes = the page-locked host buffer's own address (hipHostMalloc memory is
mapped into the device's address space), ids and offsets in HBM either way, and checks that the ids are the same."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from hutoken_amd import _capi, data, synth
vp, sp, kw = data.vocab_files("VG")
ctx = _capi.Context(vp, sp, kw["prefix"], kw["is_byte_encoder"])
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
d, o = synth.corpus("C3", n)
dev = torch.device("cuda", 0)
pb = _capi.PinnedArray(len(d) + 64, np.uint8)
pb.array[: len(d)] = d
db, do = torch.from_numpy(d).to(dev), torch.from_numpy(o).to(dev)
cap = ctx.ids_capacity(len(d), n)
ids = torch.empty(cap, dtype=torch.int32, device=dev)
oo = torch.empty(n + 1, dtype=torch.int64, device=dev)
err = torch.zeros(1, dtype=torch.int32, device=dev)
st = torch.cuda.current_stream().cuda_stream
def run(ptr):
    ctx.encode_device(ptr, do.data_ptr(), n, len(d), ids.data_ptr(), cap, oo.data_ptr(), 0, err.data_ptr(), st)
res = {}
for name, ptr in (("HBM", db.data_ptr()), ("page-locked host", pb.array.ctypes.data), ("HBM", db.data_ptr())):
    run(ptr); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(3): run(ptr)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / 3
    h = hash(ids[: int(oo[-1])].cpu().numpy().tobytes())
    res.setdefault("h", h)
    assert h == res["h"] and int(err.item()) == 0
    print(f"input in {name}: {dt*1e3:.2f} ms per batch, {len(d)/dt/1e9:.1f} GB/s (tile kernel {ctx.last_timing()[0]:.2f} ms)", flush=True)
