#!/usr/bin/env python3
"""hutk_encode_batch on page-locked host buffers (C3, 1 M documents by default): ms and GB/s per chunk size of the
pipelined path (HUTK_PIPE_CHUNK_MB), to choose pipe_chunk_bytes() in hutk_api.cpp.  usage: host_path_sweep.py [n_docs] [MB ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hutoken_amd import _capi, data, synth

n_docs = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
sizes = [int(x) for x in sys.argv[2:]] or [0, 8, 12, 16, 24, 32, 48, 64]
vp, sp, kw = data.vocab_files("VG")
ctx = _capi.Context(vp, sp, kw["prefix"], kw["is_byte_encoder"])
d, o = synth.corpus("C3", n_docs)
L = _capi.load()
cap = ctx.ids_capacity(len(d), n_docs)
pb, po = _capi.PinnedArray(len(d), np.uint8), _capi.PinnedArray(n_docs + 1, np.int64)
pi, poo = _capi.PinnedArray(cap, np.int32), _capi.PinnedArray(n_docs + 1, np.int64)
pb.array[:] = d
po.array[:] = o


def once():
    rc = L.hutk_encode_batch(ctx.handle, pb.array.ctypes.data, po.array.ctypes.data, n_docs, pi.array.ctypes.data, cap,
                             poo.array.ctypes.data, None)
    assert rc == 0, _capi.last_error()


ref = None
for mb in sizes:
    if mb:
        os.environ["HUTK_PIPE_CHUNK_MB"] = str(mb)
    else:
        os.environ.pop("HUTK_PIPE_CHUNK_MB", None)
    once()
    ts = []
    for _ in range(5):
        t = time.perf_counter()
        once()
        ts.append(time.perf_counter() - t)
    h = hash(pi.array[: int(poo.array[n_docs])].tobytes())
    ref = ref if ref is not None else h
    assert h == ref
    print(f"chunk {mb or 'default':>7} MB: best {min(ts)*1e3:6.2f} ms  median {sorted(ts)[2]*1e3:6.2f} ms  {len(d)/min(ts)/1e9:6.2f} GB/s", flush=True)
