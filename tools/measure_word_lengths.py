#!/usr/bin/env python3
"""Robustness probe: device-resident throughput on texts made of random-letter words of a given length range
(hardly any is a vocabulary key, so every word takes the merge loop; beyond 32 units a word leaves the tile
kernel for the exception kernels).  Not a benchmark line; quoted in DESIGN.md."""
import os, random, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from hutoken_amd import _capi, data

vp, sp, kw = data.vocab_files("VG")
ctx = _capi.Context(vp, sp, kw["prefix"], kw["is_byte_encoder"])
rng = random.Random(1)


def corpus(lo, hi, total=50_000_000, alphabet=b"etaoinshrdlucmfw"):
    words = [bytes(rng.choice(alphabet) for _ in range(rng.randint(lo, hi))) for _ in range(20000)]
    docs, n = [], 0
    while n < total:
        d = b" ".join(rng.choice(words) for _ in range(80))
        docs.append(d)
        n += len(d)
    offs = np.zeros(len(docs) + 1, dtype=np.int64)
    np.cumsum([len(d) for d in docs], out=offs[1:])
    return np.frombuffer(b"".join(docs), dtype=np.uint8), offs


dev = torch.device("cuda", 0)
ranges = [(3, 9), (10, 16), (17, 31), (33, 48), (49, 62), (70, 120), (300, 900)]
if len(sys.argv) > 2:
    ranges = [(int(sys.argv[1]), int(sys.argv[2]))]
for lo, hi in ranges:
    d, o = corpus(lo, hi)
    n_docs = len(o) - 1
    db, do = torch.from_numpy(d.copy()).to(dev), torch.from_numpy(o).to(dev)
    cap = ctx.ids_capacity(len(d), n_docs)
    ids = torch.empty(cap, dtype=torch.int32, device=dev)
    oo = torch.empty(n_docs + 1, dtype=torch.int64, device=dev)
    err = torch.zeros(1, dtype=torch.int32, device=dev)
    st = torch.cuda.current_stream().cuda_stream

    def run():
        ctx.encode_device(db.data_ptr(), do.data_ptr(), n_docs, len(d), ids.data_ptr(), cap, oo.data_ptr(), 0,
                          err.data_ptr(), st)
    run()
    torch.cuda.synchronize()
    t = time.perf_counter()
    run()
    run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / 2
    print(f"random words of {lo}-{hi} letters: {len(d)/1e6:.0f} MB in {dt*1e3:.1f} ms = {len(d)/dt/1e9:.1f} GB/s, "
          f"ids/byte {int(oo[-1])/len(d):.3f}", flush=True)
