#!/usr/bin/env python3
"""Latency of the small calls (one document, small batches): launch-bound, quoted in DESIGN.md."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hutoken_amd as hutoken
from hutoken_amd import data, synth

vp, sp, kw = data.vocab_files("VG")
hutoken.initialize(vp, sp, **kw)
s = "How can the net amount of entropy of the universe be massively decreased?"
hutoken.encode(s)
t = time.perf_counter()
for _ in range(200):
    hutoken.encode(s)
print(f"encode(73-byte sentence): {(time.perf_counter() - t) / 200 * 1e6:.0f} us per call")
d, o = synth.corpus("C3", 20000)
docs = synth.docs_as_str(d, o)
for n in (1, 16, 256, 4096, 20000):
    hutoken.batch_encode(docs[:n], 1)
    t = time.perf_counter()
    reps = 20 if n <= 4096 else 5
    for _ in range(reps):
        hutoken.batch_encode(docs[:n], 1)
    dt = (time.perf_counter() - t) / reps
    nb = int(o[n])
    print(f"batch_encode({n} docs, {nb/1e3:.0f} kB, Python lists in and out): {dt*1e3:.2f} ms, {nb/dt/1e6:.1f} MB/s")
