#!/usr/bin/env python3
"""Diagnostic: mean shader cycles per phase of k_tiles (clock64 stamps), GPU only."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from hutoken_amd import _capi, data, synth

name = sys.argv[1] if len(sys.argv) > 1 else "C3"
n_docs = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
vocab = sys.argv[3] if len(sys.argv) > 3 else "VG"
vp, sp, kw = data.vocab_files(vocab)
ctx = _capi.Context(vp, sp, kw["prefix"], kw["is_byte_encoder"])
if name.startswith("words:"):  # words:LO:HI -- random words of LO..HI letters (every word through the merge loop / exception kernels)
    lo, hi = (int(x) for x in name.split(":")[1:3])
    d, o = synth.random_words(lo, hi, n_docs, 8)
elif name == "cjk":  # paragraphs of CJK characters: under the reference's splitter a few words of ~1 KB per document
    d, o = synth.cjk_paragraphs(n_docs)
else:
    d, o = synth.corpus(name, n_docs)
dev = torch.device("cuda", 0)
db, do = torch.from_numpy(d).to(dev), torch.from_numpy(o).to(dev)
cap = ctx.ids_capacity(len(d), n_docs)
ids = torch.empty(cap, dtype=torch.int32, device=dev)
oo = torch.empty(n_docs + 1, dtype=torch.int64, device=dev)
err = torch.zeros(1, dtype=torch.int32, device=dev)
st = torch.cuda.current_stream().cuda_stream
def run():
    ctx.encode_device(db.data_ptr(), do.data_ptr(), n_docs, len(d), ids.data_ptr(), cap, oo.data_ptr(), 0, err.data_ptr(), st)
for _ in range(3): run()
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(5): run()
torch.cuda.synchronize()
dt = (time.perf_counter() - t) / 5
print(f"{name} {n_docs} docs {len(d)/1e6:.1f} MB vocab {vocab}: {dt*1e3:.3f} ms/step  {len(d)/dt/1e9:.2f} GB/s  tile-kernel {ctx.last_timing()[0]:.3f} ms  ids {int(oo[-1])}")
print("  tables:", ctx.table_stats())
ctx.profile(True)
run(); torch.cuda.synchronize()
tb = _capi.load().hutk_debug_tile_bytes()
n_tiles = (len(d) + tb - 1) // tb
if not os.environ.get("HUTK_MERGE_STAMPS_BUILD"):  # (a merge-stamps build spends the ten stamps inside the merge phase: no phase table)
    ph = ctx.profile_read(n_tiles)
    names = ["total", "1 stage+docs", "2 classify", "3 byte pairs", "4 bucket words", "5 merge", "6 scan+meta", "7 write ids", "8 docpos", "9 -"]
    for nme, v in zip(names, ph):
        print(f"  {nme:18s} {v:10.0f} cyc  {100*v/ph[0]:5.1f}%")
if os.environ.get("HUTK_MERGE_STAMPS_BUILD"):
    # a -DHUTK_MERGE_STAMPS=1 build: the stamps are inside the merge phase; per wavefront of the workgroup
    raw = ctx.profile_raw(n_tiles)[: n_tiles // 4 * 4].reshape(-1, 4, 10)
    names = ["barrier 1 (wait)", "pool fill", "barrier 2 (wait)", "set-up", "trips", "barrier 3 (wait)"]
    for wv in range(4):
        dlt = np.diff(raw[:, wv, :7], axis=1).mean(axis=0)
        print(f"  wavefront {wv}: " + "  ".join(f"{n} {v:.0f}" for n, v in zip(names, dlt)))
    span = (raw[:, :, 6].max(axis=1) - raw[:, :, 0].min(axis=1)).mean()
    print(f"  workgroup: first arrival to last exit {span:.0f} cycles")
    simd = raw[:, :, 0] & 3
    print("  SIMD of wavefront 0..3 of a workgroup, share of workgroups: " +
          "  ".join("wv%d %s" % (w, np.round(np.bincount(simd[:, w], minlength=4) / len(simd), 2).tolist()) for w in range(4)))
    w0 = raw[:, 0, :]  # the wavefront with the pool's longest words: lane 0's clock inside the trips
    trips = (w0[:, 9] >> 40).astype(np.float64)
    res = (w0[:, 9] & ((1 << 40) - 1)).astype(np.float64)
    ok = trips > 0
    print(f"  wavefront 0 trips: {trips[ok].mean():.1f} per workgroup; per trip: apply+issue {(w0[ok, 7] / trips[ok]).mean():.0f}  "
          f"rescan {(w0[ok, 8] / trips[ok]).mean():.0f}  wait+resolve+store {(res[ok] / trips[ok]).mean():.0f} cycles")
ctx.profile(False)
