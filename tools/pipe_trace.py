#!/usr/bin/env python3
"""Host-side time stamps of the chunked host path (HUTK_PIPE_TRACE=1 makes hutk_encode_batch print them on stderr): per
chunk the loop start, the end of the wait for the chunk buffers, and the wait for the chunk's id total -- default
chunking and 64 MB chunks one after the other."""
import os, sys, time
sys.path.insert(0, '.')
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
def once():
    rc = L.hutk_encode_batch(ctx.handle, pb.array.ctypes.data, po.array.ctypes.data, n_docs, pi.array.ctypes.data, cap, poo.array.ctypes.data, None)
    assert rc == 0
for mb in (None, "64", None):
    if mb: os.environ["HUTK_PIPE_CHUNK_MB"] = mb
    else: os.environ.pop("HUTK_PIPE_CHUNK_MB", None)
    os.environ.pop("HUTK_PIPE_TRACE", None)
    once(); once()
    os.environ["HUTK_PIPE_TRACE"] = "1"
    sys.stderr.write(f"--- chunk {mb or 'default'}\n"); sys.stderr.flush()
    t = time.perf_counter(); once(); sys.stderr.write(f"call {1e3*(time.perf_counter()-t):.2f} ms\n")
