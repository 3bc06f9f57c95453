import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from hutoken_amd import _capi, data, synth
vp, sp, kw = data.vocab_files("VG")
dev = torch.device("cuda", 0)
n = 200000
ctxs = [_capi.Context(vp, sp, kw["prefix"], kw["is_byte_encoder"]) for _ in range(2)]
bat = []
for i, c in enumerate(ctxs):
    d, o = synth.corpus("C3", n, first_doc=i * n)
    db, do = torch.from_numpy(d).to(dev), torch.from_numpy(o).to(dev)
    cap = c.ids_capacity(len(d), n)
    ids = torch.empty(cap, dtype=torch.int32, device=dev); oo = torch.empty(n + 1, dtype=torch.int64, device=dev)
    err = torch.zeros(1, dtype=torch.int32, device=dev)
    st = torch.cuda.Stream(dev)
    bat.append((c, db, do, len(d), ids, cap, oo, err, st))
def run(b):
    c, db, do, nb, ids, cap, oo, err, st = b
    c.encode_device(db.data_ptr(), do.data_ptr(), n, nb, ids.data_ptr(), cap, oo.data_ptr(), 0, err.data_ptr(), st.cuda_stream)
for b in bat: run(b); run(b)
torch.cuda.synchronize()
def t_one(reps=6):
    t = time.perf_counter()
    for _ in range(reps): run(bat[0])
    torch.cuda.synchronize(); return (time.perf_counter() - t) / reps
def t_two(reps=6):
    t = time.perf_counter()
    for _ in range(reps): run(bat[0]); run(bat[1])
    torch.cuda.synchronize(); return (time.perf_counter() - t) / reps
a = t_one(); b = t_two()
nb = bat[0][3]
print(f"one stream: {a*1e3:.3f} ms/batch {nb/a/1e9:.1f} GB/s; two streams: {b*1e3:.3f} ms per pair = {b/2*1e3:.3f} ms/batch {2*nb/b/1e9:.1f} GB/s")
