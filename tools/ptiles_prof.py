#!/usr/bin/env python3
"""What the wavefronts of k_ptiles spend their cycles on (a -DHUTK_PT_PROF=1 build: tools/build_variant.sh ptprof -DHUTK_PT_PROF=1;
HUTOKEN_AMD_LIB=hutoken_amd/lib/ab/ptprof.so).  GPU only.   ptiles_prof.py [CORPUS N_DOCS]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["HUTK_PTILES"] = "1"
import numpy as np, torch
from hutoken_amd import _capi, data, synth

name = sys.argv[1] if len(sys.argv) > 1 else "C3"
n_docs = int(sys.argv[2]) if len(sys.argv) > 2 else 200000
vp, sp, kw = data.vocab_files("VG")
ctx = _capi.Context(vp, sp, kw["prefix"], kw["is_byte_encoder"])
if name.startswith("words:"):
    lo, hi = (int(x) for x in name.split(":")[1:3])
    d, o = synth.random_words(lo, hi, n_docs, 8)
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
print(f"{name} {n_docs} docs {len(d)/1e6:.1f} MB: tile kernel {ctx.last_timing()[0]:.3f} ms (counters off)")
ctx.profile(True)
run(); run(); torch.cuda.synchronize()
print(f"  with counters: tile kernel {ctx.last_timing()[0]:.3f} ms")
tb = _capi.load().hutk_debug_tile_bytes()
n_tiles = (len(d) + tb - 1) // tb
raw = ctx.profile_raw(n_tiles).reshape(-1)
n_waves = int(os.environ.get("PT_GRID", "256")) * int(os.environ.get("PT_WAVES", "16"))
r = raw[: n_waves * 16].reshape(n_waves, 16).astype(np.float64)
names = ["total", "front end", "merge", "epilogue", "idle", "tiles", "merge calls", "trips", "refills", "words merged", "enqueue waits"]
tot = r[:, 0].mean()
print(f"  wavefronts {n_waves}; mean life {tot:.0f} cycles, longest {r[:,0].max():.0f}, shortest {r[:,0].min():.0f}")
for k in range(1, 5):
    print(f"  {names[k]:12s} {r[:,k].mean():12.0f} cycles  {100*r[:,k].mean()/tot:5.1f} %")
print(f"  rest (loop, enqueue) {100*(tot - r[:,1:5].sum(axis=1).mean())/tot:5.1f} %")
tiles = r[:, 5].sum()
print(f"  tiles {tiles:.0f}; per tile: front end {r[:,1].sum()/tiles:.0f} cycles, epilogue {r[:,3].sum()/tiles:.0f}, merge {r[:,2].sum()/tiles:.0f}")
fe = ["stage + documents", "classify", "one-byte words + lists", "short words", "long words + tail"]
print("  front end per tile: " + ", ".join(f"{fe[k]} {r[:,11+k].sum()/tiles:.0f}" for k in range(5)))
print(f"  merge calls {r[:,6].sum():.0f}, trips {r[:,7].sum():.0f} ({r[:,2].sum()/max(r[:,7].sum(),1):.0f} cycles each), refills {r[:,8].sum():.0f}, words {r[:,9].sum():.0f} "
      f"({r[:,9].sum()/tiles:.1f} per tile, {r[:,9].sum()/max(r[:,8].sum(),1):.1f} per refill), enqueue waits {r[:,10].sum():.0f}")
