#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer entry point hutk_encode_batch (numpy in, numpy out): the same
kernels as bench.py plus the H2D copy of bytes/offsets and the D2H copy of ids/offsets.  Never bench.py's
`value`; quoted in DESIGN.md section 5."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hutoken_amd import _capi, data, synth

name = sys.argv[1] if len(sys.argv) > 1 else "C3"
n_docs = int(sys.argv[2]) if len(sys.argv) > 2 else 200000
vp, sp, kw = data.vocab_files("VG")
ctx = _capi.Context(vp, sp, kw["prefix"], kw["is_byte_encoder"])
d, o = synth.corpus(name, n_docs)
import ctypes as C
import numpy as np
L = _capi.load()
cap = ctx.ids_capacity(len(d), n_docs)


def run(bytes_a, offs_a, ids_a, oo_a, label):
    def once():
        rc = L.hutk_encode_batch(ctx.handle, bytes_a.ctypes.data, offs_a.ctypes.data, n_docs, ids_a.ctypes.data, cap,
                                 oo_a.ctypes.data, None)
        assert rc == 0, _capi.last_error()
    once()
    best = 1e9
    for _ in range(4):
        t = time.perf_counter()
        once()
        best = min(best, time.perf_counter() - t)
    print(f"{name} {n_docs} docs {len(d)/1e6:.1f} MB, hutk_encode_batch host in/out, {label}: {best*1e3:.1f} ms, "
          f"{len(d)/best/1e9:.2f} GB/s, ids {int(oo_a[n_docs])}", flush=True)
    return ids_a[: int(oo_a[n_docs])].copy(), oo_a.copy()


ids_p = np.empty(cap, dtype=np.int32)
oo_p = np.empty(n_docs + 1, dtype=np.int64)
r_page = run(d, o, ids_p, oo_p, "pageable numpy arrays")
pb, po = _capi.PinnedArray(len(d), np.uint8), _capi.PinnedArray(n_docs + 1, np.int64)
pi, poo = _capi.PinnedArray(cap, np.int32), _capi.PinnedArray(n_docs + 1, np.int64)
pb.array[:] = d
po.array[:] = o
r_pin = run(pb.array, po.array, pi.array, poo.array, "page-locked buffers (hutk_host_alloc)")
assert np.array_equal(r_page[0], r_pin[0]) and np.array_equal(r_page[1], r_pin[1])
os.environ["HUTK_NO_PIPELINE"] = "1"
r_simple = run(pb.array, po.array, pi.array, poo.array, "page-locked buffers, unchunked path")
assert np.array_equal(r_simple[0], r_pin[0]) and np.array_equal(r_simple[1], r_pin[1])
