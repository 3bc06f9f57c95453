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
ctx.encode_packed(d, o)
best = 1e9
for _ in range(3):
    t = time.perf_counter()
    ids, oo, st, rc = ctx.encode_packed(d, o)
    best = min(best, time.perf_counter() - t)
print(f"{name} {n_docs} docs {len(d)/1e6:.1f} MB, host buffers in/out (pageable): {best*1e3:.1f} ms, "
      f"{len(d)/best/1e9:.2f} GB/s, ids {int(oo[-1])}")
