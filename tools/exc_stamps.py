#!/usr/bin/env python3
"""Where a trip of d_exc_group_fast<2> spends its cycles (a build with -DHUTK_LAB_EXC_STAMPS=1; s_memtime sums per wavefront).
usage: HUTOKEN_AMD_LIB=hutoken_amd/lib/ab/NAME.so exc_stamps.py [lo hi docs]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np, torch
from hutoken_amd import _capi, data, synth
lo, hi, n = (int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (70, 120, 100000)))
vp, sp, kw = data.vocab_files("VG")
ctx = _capi.Context(vp, sp, kw["prefix"], kw["is_byte_encoder"])
d, o = synth.random_words(lo, hi, n, 8)
dev = torch.device("cuda", 0)
db, do = torch.from_numpy(d).to(dev), torch.from_numpy(o).to(dev)
cap = ctx.ids_capacity(len(d), n)
ids = torch.empty(cap, dtype=torch.int32, device=dev); oo = torch.empty(n + 1, dtype=torch.int64, device=dev); err = torch.zeros(1, dtype=torch.int32, device=dev)
def run(): ctx.encode_device(db.data_ptr(), do.data_ptr(), n, len(d), ids.data_ptr(), cap, oo.data_ptr(), 0, err.data_ptr(), torch.cuda.current_stream().cuda_stream)
run(); run(); torch.cuda.synchronize()
ctx.profile(True)
run(); torch.cuda.synchronize()
L = _capi.load()
tb = L.hutk_debug_tile_bytes()
n_tiles = (len(d) + tb - 1) // tb
raw = np.zeros((n_tiles, 10), dtype=np.int64)
assert L.hutk_debug_profile_raw(ctx._h, n_tiles, raw.ctypes.data) == 0
g = int(os.environ.get("EXB_QUAD", "5120"))
r = raw[:g, :10].astype(np.float64)
r = r[r[:, 4] > 0]
tr = r[:, 4].sum()
print(f"words {lo}-{hi}: {len(r)} wavefronts with trips; per wavefront: {r[:,0].mean():.0f} cycles alive, {r[:,5].mean():.1f} lots, {r[:,4].mean():.0f} trips")
print(f"  per trip: neighbours + issue {r[:,1].sum()/tr:.0f}   row search {r[:,2].sum()/tr:.0f}   resolve + minimum {r[:,3].sum()/tr:.0f}   sum {r[:,1:4].sum()/tr:.0f}")
print(f"  per lot: cursor + list entry + record {r[:,6].sum()/r[:,5].sum():.0f}   bytes, byte-pair entries, row {r[:,7].sum()/r[:,5].sum():.0f}   first search {r[:,8].sum()/r[:,5].sum():.0f}   output {r[:,9].sum()/r[:,5].sum():.0f}")
print(f"  outside the trips (set-up, output, cursor): {(r[:,0].sum() - r[:,1:4].sum())/r[:,5].sum():.0f} cycles per lot; trips are {r[:,1:4].sum()/r[:,0].sum()*100:.1f} % of the wavefronts' life")
