#!/usr/bin/env python3
"""Times the reference (oracle/_ref) and the C restatement (oracle/) on the frozen corpora in THIS
machine.  Output goes into BASELINE.md section 4."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hutoken_amd import data, synth
from oracle import oracle as O, ref

vp, sp, kw = data.vocab_files("VG")
tok = ref.RefTokenizer(vp, sp, kw["prefix"], kw["is_byte_encoder"])
orc = O.Oracle(vp, sp, kw["prefix"], kw["is_byte_encoder"])
print("cpu_count", os.cpu_count())
for name, n in (("C2", 100000), ("C3", 100000)):
    d, o = synth.corpus(name, n)
    docs = synth.docs_as_str(d, o)
    mb = int(o[-1]) / 1e6
    for k in (1, 8):
        best = 1e9
        tok.batch_encode(docs[:2000], k)
        for _ in range(3 if k > 1 else 1):
            t = time.perf_counter(); tok.batch_encode(docs, k); best = min(best, time.perf_counter() - t)
        print(f"reference batch_encode {name} {n} docs {mb:.1f} MB threads={k}: {mb / best:.2f} MB/s")
    for k in (1, 8):
        best = 1e9
        for _ in range(3 if k > 1 else 1):
            t = time.perf_counter(); orc.encode_packed(d, o, k); best = min(best, time.perf_counter() - t)
        print(f"oracle (C restatement, packed I/O) {name} threads={k}: {mb / best:.2f} MB/s")
