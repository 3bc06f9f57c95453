#!/usr/bin/env python3
"""Long-word corpora (exception kernels), whole-step time, for several library builds on one box; a sample against the oracle.
usage: exc_ab.py lib1.so lib2.so ..."""
import os, subprocess, sys, re
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cases = [("words:17:31", "100000", "VG"), ("words:33:62", "100000", "VG"), ("words:300:900", "20000", "VG"), ("words:70:120", "100000", "VG"), ("words:130:250", "50000", "VG"), ("words:70:250", "20000", "VL"), ("cjk", "20000", "VG:noseam"), ("cjktext", "20000", "VC"), ("cjk", "20000", "VC")]
if os.environ.get("EXC_AB_CASES"):
    cases = [c for c in cases if c[0] in os.environ["EXC_AB_CASES"].split(",")]
code = r'''
import os, sys, time
sys.path.insert(0, %r)
import numpy as np, torch
from hutoken_amd import _capi, data, synth
from oracle import oracle as O
name, n, vocab = sys.argv[1], int(sys.argv[2]), sys.argv[3]
if vocab.endswith(":noseam"):
    os.environ["HUTK_NO_SEAM"] = "1"; vocab = vocab.split(":")[0]
vp, sp, kw = data.vocab_files(vocab)
ctx = _capi.Context(vp, sp, kw["prefix"], kw["is_byte_encoder"])
if name.startswith("words:"):
    lo, hi = (int(x) for x in name.split(":")[1:3]); d, o = synth.random_words(lo, hi, n, 8)
elif name == "cjk": d, o = synth.cjk_paragraphs(n)
else: d, o = synth.cjk_text(n)
dev = torch.device("cuda", 0)
db, do = torch.from_numpy(d).to(dev), torch.from_numpy(o).to(dev)
cap = ctx.ids_capacity(len(d), n)
ids = torch.empty(cap, dtype=torch.int32, device=dev); oo = torch.empty(n + 1, dtype=torch.int64, device=dev); err = torch.zeros(1, dtype=torch.int32, device=dev)
def run(): ctx.encode_device(db.data_ptr(), do.data_ptr(), n, len(d), ids.data_ptr(), cap, oo.data_ptr(), 0, err.data_ptr(), torch.cuda.current_stream().cuda_stream)
run(); torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(3): run()
torch.cuda.synchronize()
dt = (time.perf_counter() - t) / 3
k = min(n, 300)
orc = O.Oracle(vp, sp, kw["prefix"], kw["is_byte_encoder"])
ids_o, oo_o, _ = orc.encode_packed(d[: o[k]], o[: k + 1], 8)
ok = np.array_equal(oo[: k + 1].cpu().numpy(), oo_o) and np.array_equal(ids[: int(oo_o[-1])].cpu().numpy(), ids_o)
print(f"{name} {n} {sys.argv[3]}: {dt*1e3:.2f} ms  {len(d)/dt/1e9:.2f} GB/s  err {int(err.item())}  first {k} documents vs oracle: {'equal' if ok else 'DIFFERENT'}")
''' % root
for lib in sys.argv[1:]:
    print("==", lib, flush=True)
    for name, n, vocab in cases:
        env = dict(os.environ, HUTOKEN_AMD_LIB=os.path.join(root, lib))
        out = subprocess.run([sys.executable, "-c", code, name, n, vocab], env=env, capture_output=True, text=True)
        print((out.stdout.strip().split("\n") or [""])[-1] or out.stderr[-300:], flush=True)
