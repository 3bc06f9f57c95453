#!/usr/bin/env python3
"""One device-resident batch of MORE THAN 4 GiB of text through hutk_encode_batch_device (64-bit offsets, tile and id
indices beyond 2^32 bytes): the ids of the first and the last documents against the oracle, the id total against the
sum over the batch cut into ordinary pieces.  usage: big_batch_check.py [N_DOCS=8600000]   (GPU, ~60 GB of HBM)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from hutoken_amd import _capi, data, synth
from oracle import oracle as O

n_docs = int(sys.argv[1]) if len(sys.argv) > 1 else 8_600_000
vp, sp, kw = data.vocab_files("VG")
ctx = _capi.Context(vp, sp, kw["prefix"], kw["is_byte_encoder"])
O.build()
orc = O.Oracle(vp, sp, kw["prefix"], kw["is_byte_encoder"])
dev = torch.device("cuda", 0)
t = time.time()
piece = 1_000_000
parts, offs_parts, base = [], [np.zeros(1, dtype=np.int64)], 0
for first in range(0, n_docs, piece):
    d, o = synth.corpus("C3", min(piece, n_docs - first), first_doc=first, threads=16)
    parts.append(d)
    offs_parts.append(o[1:] + base)
    base += int(o[-1])
offs = np.concatenate(offs_parts)
n_bytes = int(offs[-1])
print(f"{n_docs} docs, {n_bytes} bytes ({n_bytes / 2**32:.2f} x 2^32), generated in {time.time() - t:.0f} s", flush=True)
d_bytes = torch.empty(n_bytes, dtype=torch.uint8, device=dev)
pos = 0
for p in parts:
    d_bytes[pos:pos + len(p)].copy_(torch.from_numpy(p))
    pos += len(p)
d_offs = torch.from_numpy(offs).to(dev)
cap = ctx.ids_capacity(n_bytes, n_docs)
d_ids = torch.empty(cap, dtype=torch.int32, device=dev)
d_oo = torch.empty(n_docs + 1, dtype=torch.int64, device=dev)
d_err = torch.zeros(1, dtype=torch.int32, device=dev)
st = torch.cuda.current_stream(dev).cuda_stream
for rep in range(2):
    torch.cuda.synchronize()
    t = time.time()
    ctx.encode_device(d_bytes.data_ptr(), d_offs.data_ptr(), n_docs, n_bytes, d_ids.data_ptr(), cap, d_oo.data_ptr(), 0,
                      d_err.data_ptr(), st)
    torch.cuda.synchronize()
    dt = time.time() - t
assert int(d_err.item()) == 0, int(d_err.item())
oo = d_oo.cpu().numpy()
total = int(oo[-1])
print(f"one launch sequence: {dt * 1e3:.1f} ms, {n_bytes / dt / 1e9:.1f} GB/s, {total} ids", flush=True)
K = 3000
for name, a, b in (("first", 0, K), ("last", n_docs - K, n_docs), ("across 2^32", int(np.searchsorted(offs, 2**32)) - K // 2,
                                                                    int(np.searchsorted(offs, 2**32)) + K // 2)):
    a, b = max(a, 0), min(b, n_docs)
    lo, hi = int(offs[a]), int(offs[b])
    seg = np.concatenate([p for p in [d_bytes[lo:hi].cpu().numpy()]])
    ids_o, oo_o, _ = orc.encode_packed(seg, offs[a:b + 1] - lo, 16)
    got = d_ids[int(oo[a]):int(oo[b])].cpu().numpy()
    assert np.array_equal(oo[a:b + 1] - oo[a], oo_o), name
    assert np.array_equal(got, ids_o), name
    print(f"  {name} {b - a} documents: ids equal the oracle's", flush=True)
# the id total against ordinary pieces (each well below 2^32 bytes)
s = 0
for i, p in enumerate(parts):
    a = i * piece
    b = min(a + piece, n_docs)
    s += int(oo[b] - oo[a])
    if i in (0, len(parts) - 1, len(parts) // 2):
        ids_h, oo_h, _, rc = ctx.encode_packed(p, offs[a:b + 1] - offs[a])
        assert rc == 0 and int(oo_h[-1]) == int(oo[b] - oo[a]), i
        assert np.array_equal(ids_h, d_ids[int(oo[a]):int(oo[b])].cpu().numpy()), i
        print(f"  piece {i}: the chunked host path gives the same {int(oo_h[-1])} ids", flush=True)
assert s == total
print("OK")
