#!/usr/bin/env python3
"""One warm-up and two calls of hutk_encode_batch on page-locked buffers (C3, 1 M documents), to be run under
`rocprofv3 --memory-copy-trace --kernel-trace`: do the copies up and down overlap?  HUTK_PIPE_CHUNK_MB selects the chunking."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hutoken_amd import _capi, data, synth
vp, sp, kw = data.vocab_files("VG")
ctx = _capi.Context(vp, sp, kw["prefix"], kw["is_byte_encoder"])
n_docs = 1_000_000
d, o = synth.corpus("C3", n_docs)
L = _capi.load()
cap = ctx.ids_capacity(len(d), n_docs)
pb, po = _capi.PinnedArray(len(d), np.uint8), _capi.PinnedArray(n_docs + 1, np.int64)
pi, poo = _capi.PinnedArray(cap, np.int32), _capi.PinnedArray(n_docs + 1, np.int64)
pb.array[:] = d; po.array[:] = o
for i in range(3):
    t = time.perf_counter()
    rc = L.hutk_encode_batch(ctx.handle, pb.array.ctypes.data, po.array.ctypes.data, n_docs, pi.array.ctypes.data, cap, poo.array.ctypes.data, None)
    assert rc == 0
    print(f"call {i}: {(time.perf_counter() - t) * 1e3:.2f} ms", flush=True)
