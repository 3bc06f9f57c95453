#!/usr/bin/env python3
"""Per-kernel GPU time of one device-resident batch of a named workload (run under rocprofv3 --kernel-trace --stats):
  kernel_times.py C3 200000 | kernel_times.py words 70 120 [docs] | kernel_times.py cjk [docs] (HUTK_NO_SEAM=1: every paragraph one word) | kernel_times.py cjktext [docs] (vocab VC)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hutoken_amd import _capi, data, synth

vp, sp, kw = data.vocab_files("VC" if sys.argv[1] in ("cjktext", "cjkvc") else "VG")
ctx = _capi.Context(vp, sp, kw["prefix"], kw["is_byte_encoder"])
if sys.argv[1] == "words":
    d, o = synth.random_words(int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]) if len(sys.argv) > 4 else 100000, 8)
elif sys.argv[1] == "cjktext":
    d, o = synth.cjk_text(int(sys.argv[2]) if len(sys.argv) > 2 else 20000)
elif sys.argv[1] == "cjkvc":  # characters drawn at random under the CJK-dense vocabulary
    d, o = synth.cjk_paragraphs(int(sys.argv[2]) if len(sys.argv) > 2 else 20000)
elif sys.argv[1] == "cjk":
    d, o = synth.cjk_paragraphs(int(sys.argv[2]) if len(sys.argv) > 2 else 50000)
else:
    d, o = synth.corpus(sys.argv[1], int(sys.argv[2]))
n = len(o) - 1
dev = torch.device("cuda", 0)
db, do = torch.from_numpy(d).to(dev), torch.from_numpy(o).to(dev)
cap = ctx.ids_capacity(len(d), n)
ids = torch.empty(cap, dtype=torch.int32, device=dev)
oo = torch.empty(n + 1, dtype=torch.int64, device=dev)
err = torch.zeros(1, dtype=torch.int32, device=dev)
for _ in range(5):
    ctx.encode_device(db.data_ptr(), do.data_ptr(), n, len(d), ids.data_ptr(), cap, oo.data_ptr(), 0, err.data_ptr(),
                      torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
print(len(d), "bytes", n, "docs", int(oo[-1]), "ids")
