#!/usr/bin/env python3
"""k_ptiles (HUTK_PTILES=1) against k_tiles (HUTK_PTILES=0) on the same device buffers: every id and offset equal, a
sample of the documents against the oracle, and the time of both.  GPU only.
  ptiles_check.py [CORPUS N_DOCS [VOCAB]] ...   CORPUS: C2 | C3 | C5 | words:LO:HI | cjk"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("HUTK_PTILES_MIN_TILES", "1")
import numpy as np, torch
from hutoken_amd import _capi, data, synth

def corpus(name, n):
    if name.startswith("words:"):
        lo, hi = (int(x) for x in name.split(":")[1:3])
        return synth.random_words(lo, hi, n, 8)
    if name == "cjk":
        return synth.cjk_paragraphs(n)
    return synth.corpus(name, n)

def main():
    args = sys.argv[1:] or ["C3", "2000", "C3", "200000"]
    jobs = []
    while args:
        name, n = args[0], int(args[1]); args = args[2:]
        vocab = "VG"
        if args and not args[0][0].isdigit() and args[0] in ("VG", "VL", "VGM"):
            vocab = args[0]; args = args[1:]
        jobs.append((name, n, vocab))
    dev = torch.device("cuda", 0)
    ok_all = True
    for name, n_docs, vocab in jobs:
        vp, sp, kw = data.vocab_files("VG" if vocab == "VGM" else vocab)
        mp = data.merges_file("VG") if vocab == "VGM" and hasattr(data, "merges_file") else None
        ctx = _capi.Context(vp, sp, kw["prefix"], kw["is_byte_encoder"], merges_path=mp) if mp else _capi.Context(vp, sp, kw["prefix"], kw["is_byte_encoder"])
        d, o = corpus(name, n_docs)
        db, do = torch.from_numpy(d).to(dev), torch.from_numpy(o).to(dev)
        cap = ctx.ids_capacity(len(d), n_docs)
        oo = torch.empty(n_docs + 1, dtype=torch.int64, device=dev)
        err = torch.zeros(1, dtype=torch.int32, device=dev)
        stt = torch.zeros(n_docs, dtype=torch.int32, device=dev)
        st = torch.cuda.current_stream().cuda_stream
        res = {}
        for mode in ("0", "1", "auto"):
            if mode == "auto":
                os.environ.pop("HUTK_PTILES", None)  # both kernels are enqueued, the batch's bytes decide on the device
            else:
                os.environ["HUTK_PTILES"] = mode
            ids = torch.full((cap,), -7, dtype=torch.int32, device=dev)
            def run():
                ctx.encode_device(db.data_ptr(), do.data_ptr(), n_docs, len(d), ids.data_ptr(), cap, oo.data_ptr(), stt.data_ptr(), err.data_ptr(), st)
            run(); torch.cuda.synchronize()
            e = int(err.item())
            for _ in range(2): run()
            torch.cuda.synchronize()
            t = time.perf_counter()
            reps = 5
            tk = 0.0
            for _ in range(reps):
                run()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t) / reps
            tk = ctx.last_timing()[0]
            total = int(oo[-1])
            res[mode] = (ids[:total].cpu().numpy().copy(), oo.cpu().numpy().copy(), stt.cpu().numpy().copy(), e, dt, tk)
            print(f"{name} {n_docs} {vocab} PTILES={mode}: err {e}  {dt*1e3:.3f} ms/step  {len(d)/dt/1e9:.2f} GB/s  tile kernel {tk:.3f} ms  ids {total}", flush=True)
        a, b = res["0"], res["1"]
        c = res["auto"]
        if not (a[3] == c[3] and np.array_equal(a[1], c[1]) and np.array_equal(a[0], c[0]) and np.array_equal(a[2], c[2])):
            ok_all = False
            print("  MISMATCH between k_tiles and the automatic choice")
        same = a[3] == b[3] and np.array_equal(a[1], b[1]) and np.array_equal(a[0], b[0]) and np.array_equal(a[2], b[2])
        if not same:
            ok_all = False
            print("  MISMATCH between the two kernels:", "err", a[3], b[3], "offsets equal", np.array_equal(a[1], b[1]))
            if not np.array_equal(a[1], b[1]):
                bad = np.nonzero(a[1] != b[1])[0]
                print("  first differing offset at doc", bad[0], "of", len(bad), ":", a[1][bad[0]], b[1][bad[0]])
            elif len(a[0]) == len(b[0]):
                bad = np.nonzero(a[0] != b[0])[0]
                dd = np.searchsorted(a[1], bad[0], side="right") - 1
                print("  first differing id at", bad[0], "doc", dd, "n differing", len(bad), a[0][bad[0]:bad[0]+8], b[0][bad[0]:bad[0]+8])
                print("  text:", bytes(d[o[dd]:o[dd+1]])[:200])
        else:
            print("  kernels agree: ids, offsets, status, error word")
        # a sample against the oracle
        from oracle import oracle as O
        orc = O.Oracle(vp, sp, kw["prefix"], kw["is_byte_encoder"])
        ns = min(n_docs, 3000)
        ids_o, oo_o, _ = orc.encode_packed(d[: o[ns]], o[: ns + 1], 8)
        okk = np.array_equal(b[1][: ns + 1], oo_o) and np.array_equal(b[0][: oo_o[-1]], ids_o)
        print("  first", ns, "documents against the oracle:", "equal" if okk else "DIFFERENT")
        ok_all = ok_all and okk
        ctx.close()
    print("ALL OK" if ok_all else "FAILED")
    sys.exit(0 if ok_all else 1)

main()
