#!/usr/bin/env python3
"""bench.py -- GB/s of input text encoded (GPT-2-shaped 50257-entry vocab, bit-exact
ids) on N MI355X, the metric BASELINE.json names.

  python bench.py --gpus 1 --steps 10 --warmup 3
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

A step = one pass of the batch encode path (hutk_encode_batch_device: every kernel
of the pipeline) over one batch that is ALREADY RESIDENT IN HBM: packed UTF-8 bytes
+ int64 offsets in, int32 ids + int64 offsets out, all device buffers.
Workload (`value`, scaling "weak"): corpus C3, 1,000,000 synthetic documents per GPU,
mean 512 B, mixed UTF-8 (hutoken_amd/csrc/hutk_synth.c), vocabulary VG (data/vg50257_*).
Documents shard trivially: rank r encodes documents [r*1M, (r+1)*1M) of the generator
with its own context; the only collective is an all-gather of the per-rank id totals
(RCCL), inside every timed step.

The JSON line also carries
  roofline      algorithmic HBM bytes of the dominant kernel (k_tiles) / its mean duration from HIP
                events recorded around it on the launch stream in every timed step, against 8 TB/s;
                `wavefront_cycles`: where a resident wavefront's cycles go (committed PMC profile; shares of one
                counter's unit, none can pass 1); `latency`: what does bind the kernel (occupancy A/B, L2 latency counters)
  strong        (N > 1) BASELINE config 4: the SAME 1,000,000 documents cut into N byte-balanced
                contiguous shards (hutoken_amd.sharding.shard_by_bytes), rank r generating only its range
  end_to_end    (N = 1) the drop-in's host entry points on the same workload: page-locked host buffers in
                and out through hutk_encode_batch (PCIe inclusive), and the Python list API on a sample
  secondary     (N = 1) other configurations (config 5's single-GPU half and the merges path at 1 M documents) and
                off-distribution text (200k documents each), device-resident
  cpu_baseline  (N = 1) the reference itself (oracle/_ref, compiled from the reference sources; kind
                "reference") timed on the host cores on a bounded sample of the same workload: its list API
                (`value`) and, marshalling-free, its C core through the internal encode() seam (`seam`).
Everything measured on the GPU is checked against the oracle on its first 2000 documents.
"""
import argparse
import glob
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8.0 TB/s spec
N_SIMD = 1024           # 256 CUs x 4 SIMDs
VERIFY_DOCS = 2000


def cpu_baseline(vp, sp, kw, corpus_name, n_sample, cores, mp=None):
    """Times the CPU path on the first n_sample documents of the workload."""
    from hutoken_amd import synth
    data, offs = synth.corpus(corpus_name, n_sample)
    nbytes = int(offs[-1])
    out = {"cores": cores, "sample": f"first {n_sample} documents of {corpus_name} ({nbytes / 1e6:.1f} MB)"}
    try:
        from oracle import ref
        if ref.available():
            docs = synth.docs_as_str(data, offs)
            tok = ref.RefTokenizer(vp, sp, kw["prefix"], kw["is_byte_encoder"], mp)
            tok.batch_encode(docs[:1000], cores)
            t = time.perf_counter()
            res = tok.batch_encode(docs, cores)
            dt = time.perf_counter() - t
            out.update(kind="reference", value=nbytes / dt / 1e9, unit="GB/s",
                       note="hutoken.batch_encode(list[str], num_threads=cores) of the reference "
                            "compiled from its own sources, list marshalling included",
                       n_ids=int(sum(len(r) for r in res)))
            try:  # the C core alone: `cores` pthreads on the reference's internal encode() seam, no Python objects
                n_ids, sec = tok.seam_batch(data, offs, cores)
                out["seam"] = {"value": nbytes / sec / 1e9, "unit": "GB/s", "cores": cores, "n_ids": int(n_ids),
                               "note": "void encode(struct EncodeTask*) of the compiled reference (core.h:11) called once "
                                       "per document from `cores` pthreads (oracle/ref_seam.c): no list marshalling, no GIL"}
            except Exception as e:
                out["seam_error"] = repr(e)
            return out
    except Exception as e:  # fall through to the port
        out["reference_error"] = repr(e)
    from oracle import oracle as O
    orc = O.Oracle(vp, sp, kw["prefix"], kw["is_byte_encoder"], mp)
    orc.encode_packed(data[: int(offs[1000])], offs[:1001], cores)
    t = time.perf_counter()
    ids, oo, st = orc.encode_packed(data, offs, cores)
    dt = time.perf_counter() - t
    out.update(kind="port", value=nbytes / dt / 1e9, unit="GB/s",
               note="this repo's C restatement of the reference algorithm (oracle/), packed I/O",
               n_ids=int(oo[-1]))
    return out


def committed_profile(n_bytes, suffix):
    """The last committed profiles/<tag>_<suffix>.json of THIS workload (written by tools/summarize_rocprof.py from
    separate `rocprofv3 --pmc` runs of the same bench command: counters cannot be collected inside a timed run)."""
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", f"*_{suffix}.json"))):
        try:
            t = json.load(open(f))
        except Exception:
            continue
        if t.get("workload_bytes") == n_bytes:
            best = (t, os.path.relpath(f, ROOT))
    return best if best else (None, None)


class DeviceBatch:
    """A packed batch resident in HBM with its output buffers; run() enqueues one pass of the pipeline."""

    def __init__(self, ctx, data, offs, dev):
        import torch
        self.ctx, self.dev = ctx, dev
        self.n_docs, self.n_bytes = len(offs) - 1, int(offs[-1])
        self.d_bytes = torch.from_numpy(data).to(dev)
        self.d_offs = torch.from_numpy(offs).to(dev)
        self.cap = ctx.ids_capacity(self.n_bytes, self.n_docs)
        self.d_ids = torch.empty(self.cap, dtype=torch.int32, device=dev)
        self.d_oo = torch.empty(self.n_docs + 1, dtype=torch.int64, device=dev)
        self.d_err = torch.zeros(1, dtype=torch.int32, device=dev)
        self.stream = torch.cuda.current_stream(dev).cuda_stream

    def run(self):
        self.ctx.encode_device(self.d_bytes.data_ptr(), self.d_offs.data_ptr(), self.n_docs, self.n_bytes,
                               self.d_ids.data_ptr(), self.cap, self.d_oo.data_ptr(), 0, self.d_err.data_ptr(),
                               self.stream)

    def check_err(self):
        code = int(self.d_err.item())
        if code != 0:
            raise SystemExit(f"device-side error {code}")

    def n_ids(self):
        return int(self.d_oo[self.n_docs].item())

    def verify(self, orc, data, offs, cores):
        """First VERIFY_DOCS documents against the oracle: offsets and every id."""
        import numpy as np
        k = min(VERIFY_DOCS, self.n_docs)
        ids_o, oo_o, _ = orc.encode_packed(data[: int(offs[k])], offs[: k + 1], min(cores, 8))
        oo_g = self.d_oo[: k + 1].cpu().numpy()
        ids_g = self.d_ids[: int(oo_g[k])].cpu().numpy()
        if not (np.array_equal(oo_o, oo_g) and np.array_equal(ids_o, ids_g)):
            raise SystemExit("PARITY FAILURE: GPU ids differ from the oracle")
        return True


def timed(batch, steps, warmup, sync):
    """-> (seconds per step, mean k_tiles ms) of `steps` passes after `warmup` untimed ones."""
    for _ in range(warmup):
        batch.run()
    sync()
    batch.check_err()
    tile_ms = []
    t0 = time.perf_counter()
    for _ in range(steps):
        batch.run()
        tile_ms.append(batch.ctx.last_timing()[0])
    sync()
    return (time.perf_counter() - t0) / steps, sum(tile_ms) / len(tile_ms)


class _Env:
    """Environment settings for the duration of a block (the library reads HUTK_NO_SEAM when a context is created and
    HUTK_PTILES at every call)."""

    def __init__(self, kv):
        self.kv, self.old = kv or {}, {}

    def __enter__(self):
        for k, v in self.kv.items():
            self.old[k] = os.environ.get(k)
            os.environ[k] = v

    def __exit__(self, *exc):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def c3_shard(n_shards, which=0):
    """Shard `which` of BASELINE config 4's byte-balanced cut of the 1 M documents of C3 into n_shards (what ONE GPU of an
    n_shards-GPU run encodes per step): generated from the lengths alone, as bench.py's strong mode does."""
    import numpy as np
    from hutoken_amd import sharding, synth
    lens = synth.lengths("C3", 1_000_000)
    offs_all = np.zeros(1_000_001, dtype=np.int64)
    np.cumsum(lens, out=offs_all[1:])
    first, count = sharding.shard_by_bytes(offs_all, n_shards)[which]
    return synth.corpus("C3", int(count), first_doc=int(first))


def secondary_runs(dev, dev_index, cores, no_verify):
    """Other configurations and off-distribution text, device-resident, checked against the oracle."""
    import torch
    from hutoken_amd import _capi, data as hdata, synth
    from oracle import oracle as O
    sync = lambda: torch.cuda.synchronize(dev)  # noqa: E731
    out = []
    # (label, vocabulary, merges file?, generator, environment while the context is created, environment while it runs)
    cases = [("C2 x VG (BASELINE config 2: ASCII)", "VG", False, lambda: synth.corpus("C2", 200_000), None, None),
             ("C3 x VG, 1/8 byte-balanced shard of the 1 M documents (the per-GPU share of BASELINE config 4 at N = 8; "
              "projected: no 8-GPU run)", "VG", False, lambda: c3_shard(8), None, None),
             ("C3 x VG, 1/2 byte-balanced shard (config 4's per-GPU share at N = 2)", "VG", False, lambda: c3_shard(2), None, None),
             ("C5 x VL (config 5's single-GPU half at its full size: 1 M documents of Hungarian text, Llama-shaped vocab, "
              "prefix, non-byte mode)", "VL", False, lambda: synth.corpus("C5", 1_000_000), None, None),
             ("C5 x VL, 200 k documents", "VL", False, lambda: synth.corpus("C5", 200_000), None, None),
             ("C3 x VG + merges file (id-keyed merge path), 1 M documents", "VG", True, lambda: synth.corpus("C3", 1_000_000), None, None),
             ("C3 x VG, 1 M documents, the persistent tile kernel (HUTK_PTILES=1: hutk_ptiles.hip; not the default)", "VG", False,
              lambda: synth.corpus("C3", 1_000_000), None, {"HUTK_PTILES": "1"}),
             ("random words of 17-31 letters x VG (every word through the merge loop)", "VG", False,
              lambda: synth.random_words(17, 31, 200_000, 20), None, None),
             ("random words of 33-62 letters x VG (every word through the exception kernels)", "VG", False,
              lambda: synth.random_words(33, 62, 200_000, 10), None, None),
             ("random words of 70-120 letters x VG (exception kernels: two lanes per word, 32 words per wavefront)", "VG", False,
              lambda: synth.random_words(70, 120, 100_000, 8), None, None),
             ("random words of 130-250 letters x VG (exception kernels: four lanes per word)", "VG", False,
              lambda: synth.random_words(130, 250, 50_000, 8), None, None),
             ("random words of 300-900 letters x VG (exception kernels: eight / sixteen lanes per word)", "VG", False,
              lambda: synth.random_words(300, 900, 20_000, 8), None, None),
             ("CJK paragraphs x VG (words of 300-1200 bytes under the reference's splitter; seams cut them)", "VG", False,
              lambda: synth.cjk_paragraphs(60_000), None, None),
             ("CJK paragraphs x VG, the persistent tile kernel (HUTK_PTILES=1)", "VG", False,
              lambda: synth.cjk_paragraphs(60_000), None, {"HUTK_PTILES": "1"}),
             ("CJK paragraphs x VG WITHOUT the seam map (HUTK_NO_SEAM=1: every paragraph one word, as the reference sees it)", "VG",
              False, lambda: synth.cjk_paragraphs(20_000), {"HUTK_NO_SEAM": "1"}, None),
             ("ONE document of 100 MB x VG (the reference's own benchmark shape, scripts/benchmark.py:51-104), whole", "VG", False,
              lambda: synth.big_document(100_000_000), None, None),
             ("the same 100 MB document as 64 whitespace-aligned pieces (scripts/benchmark.py:26-48, threaded_benchmark.sh)", "VG",
              False, lambda: (lambda d, o: (d, synth.whitespace_chunks(d, 64)))(*synth.big_document(100_000_000)), None, None),
             ("ONE document of 1 GB x VG, whole (ids checked in tests/test_gpu_bigdoc.py, not here)", "VG", False,
              lambda: synth.big_document(1_000_000_000), None, {"_no_verify": "1"}),
             ("CJK text x VC (a vocabulary trained on CJK-dense text: merges across every frequent pair of neighbouring "
              "characters saturate the seam map's first level; data/vc12257_*)", "VC", False, lambda: synth.cjk_text(20_000), None, None),
             ("CJK paragraphs of characters drawn at random x VC (pairs the vocabulary never learnt: the seam map's second level cuts "
              "nearly everywhere)", "VC", False, lambda: synth.cjk_paragraphs(20_000), None, None),
             ("the same with HUTK_NO_SEAM2=1 (first level only)", "VC", False, lambda: synth.cjk_paragraphs(20_000),
              {"HUTK_NO_SEAM2": "1"}, None)]
    ctxs = {}
    for label, vocab, merges, gen, env_ctx, env_run in cases:
        key = (vocab, merges, tuple(sorted((env_ctx or {}).items())))
        vp, sp, kw = hdata.vocab_files(vocab)
        mp = hdata.merges_file(vocab) if merges else None
        if key not in ctxs:
            with _Env(env_ctx):
                ctxs[key] = (_capi.Context(vp, sp, kw["prefix"], kw["is_byte_encoder"], device=dev_index, merges_path=mp),
                             O.Oracle(vp, sp, kw["prefix"], kw["is_byte_encoder"], mp))
        ctx, orc = ctxs[key]
        d, o = gen()
        b = DeviceBatch(ctx, d, o, dev)
        skip_verify = bool(env_run and env_run.get("_no_verify"))
        with _Env({k: v for k, v in (env_run or {}).items() if not k.startswith("_")}):
            sec, tile_ms = timed(b, 5, 2, sync)
        ok = None if (no_verify or skip_verify) else b.verify(orc, d, o, cores)
        out.append({"workload": label, "docs": b.n_docs, "bytes": b.n_bytes, "ids": b.n_ids(),
                    "value": round(b.n_bytes / sec / 1e9, 2), "unit": "GB/s", "ms_per_step": round(sec * 1e3, 4),
                    "k_tiles_ms": round(tile_ms, 4), "verified_vs_oracle": ok})
        del b
    return out


def numa_info(arr, dev_index):
    """NUMA node(s) of a page-locked buffer's pages (move_pages(2) with no target = query) and of the GPU's PCI function:
    on a two-socket host a buffer on the far node halves the host path's rate; hutk_host_alloc asks
    for the device's node (HUTK_HOST_ALLOC_NUMA=0: does not)."""
    import ctypes
    info = {"host_alloc_numa": os.environ.get("HUTK_HOST_ALLOC_NUMA", "default (prefer the device's node)")}
    try:
        import torch
        bus = torch.cuda.get_device_properties(dev_index).pci_bus_id
        dom = getattr(torch.cuda.get_device_properties(dev_index), "pci_domain_id", 0)
        devn = torch.cuda.get_device_properties(dev_index).pci_device_id
        path = "/sys/bus/pci/devices/%04x:%02x:%02x.0/numa_node" % (dom, bus, devn)
        info["device_node"] = int(open(path).read().strip())
    except Exception as e:
        info["device_node"] = None
        info["device_node_error"] = str(e)[:80]
    try:
        npg = 64
        step = max(4096, (arr.nbytes // npg) & ~4095)
        pages = (ctypes.c_void_p * npg)(*[arr.ctypes.data + i * step for i in range(npg)])
        status = (ctypes.c_int * npg)()
        rc = ctypes.CDLL(None, use_errno=True).syscall(279, 0, npg, pages, None, status, 0)
        info["buffer_nodes"] = sorted(set(int(x) for x in status)) if rc == 0 else None
    except Exception as e:
        info["buffer_nodes"] = None
        info["buffer_nodes_error"] = str(e)[:80]
    return info


def end_to_end(ctx, orc, data, offs, vp, sp, kw, dev_index, cores, no_verify):
    """The host entry points on the same workload (PCIe inclusive; never `value`)."""
    import numpy as np
    from hutoken_amd import _capi, synth
    import hutoken_amd as hutoken
    L = _capi.load()
    n_docs, n_bytes = len(offs) - 1, int(offs[-1])
    cap = ctx.ids_capacity(n_bytes, n_docs)
    pb, po = _capi.PinnedArray(n_bytes, np.uint8), _capi.PinnedArray(n_docs + 1, np.int64)
    pi, poo = _capi.PinnedArray(cap, np.int32), _capi.PinnedArray(n_docs + 1, np.int64)
    pb.array[:] = data
    po.array[:] = offs

    def once():
        rc = L.hutk_encode_batch(ctx.handle, pb.array.ctypes.data, po.array.ctypes.data, n_docs, pi.array.ctypes.data,
                                 cap, poo.array.ctypes.data, None)
        if rc:
            raise SystemExit("hutk_encode_batch: " + _capi.last_error())
    once()
    best = 1e9
    for _ in range(3):
        t = time.perf_counter()
        once()
        best = min(best, time.perf_counter() - t)
    k = min(VERIFY_DOCS, n_docs)
    ok = None
    if not no_verify:
        ids_o, oo_o, _ = orc.encode_packed(data[: int(offs[k])], offs[: k + 1], min(cores, 8))
        ok = bool(np.array_equal(oo_o, poo.array[: k + 1]) and np.array_equal(ids_o, pi.array[: int(oo_o[k])]))
        if not ok:
            raise SystemExit("PARITY FAILURE: hutk_encode_batch ids differ from the oracle")
    out = {"packed_pinned": {"value": round(n_bytes / best / 1e9, 2), "unit": "GB/s", "ms": round(best * 1e3, 2),
                             "docs": n_docs, "verified_vs_oracle": ok, "numa": numa_info(pb.array, dev_index),
                             "note": "hutk_encode_batch, page-locked host bytes+offsets in, host ids+offsets out "
                                     "(hutk_host_alloc); chunks of whole documents, H2D / kernels / D2H overlapped"}}
    # every GPU this process sees behind ONE context (hutk_ctx_add_device): only where there is more than one, and never
    # under torchrun (there each rank has its own GPU)
    try:
        import torch
        n_vis = torch.cuda.device_count() if int(os.environ.get("WORLD_SIZE", "1")) == 1 else 1
        if os.environ.get("HUTK_BENCH_SAME_DEVICE_TWICE"):  # rehearsal on a one-GPU box
            n_vis = 2
        if n_vis > 1:
            others = [dev_index] * (n_vis - 1) if os.environ.get("HUTK_BENCH_SAME_DEVICE_TWICE") else \
                [d for d in range(n_vis) if d != dev_index]
            for d in others:
                ctx.add_device(d)
            once()
            bm = 1e9
            for _ in range(3):
                t = time.perf_counter()
                once()
                bm = min(bm, time.perf_counter() - t)
            okm = None
            if not no_verify:
                okm = bool(np.array_equal(oo_o, poo.array[: k + 1]) and np.array_equal(ids_o, pi.array[: int(oo_o[k])]))
                if not okm:  # (the line is still printed, with a top-level parity_failures entry, and the run exits non-zero)
                    out.setdefault("parity_failures", []).append("packed_pinned_all_devices")
            out["packed_pinned_all_devices"] = {
                "value": round(n_bytes / bm / 1e9, 2), "unit": "GB/s", "ms": round(bm * 1e3, 2), "devices": n_vis,
                "verified_vs_oracle": okm,
                "note": "the same call on a context with hutk_ctx_add_device for every visible GPU: byte-balanced runs of "
                        "whole documents, one host thread per device"}
    except SystemExit:
        raise
    except Exception as e:  # (reported, never fatal for the line)
        out["packed_pinned_all_devices"] = {"error": str(e)[:200]}
    for a in (pb, po, pi, poo):
        a.close()
    # the reference's own surface: list[str] -> list[list[int]]
    n_list = min(100_000, n_docs)
    docs = synth.docs_as_str(data[: int(offs[n_list])], offs[: n_list + 1])
    hutoken.initialize(vp, sp, device=dev_index, **kw)
    res = hutoken.batch_encode(docs, cores)  # (untimed: the new context's staging buffers and workspace grow to the batch's size here)
    dt = 1e9
    for _ in range(2):
        del res
        t = time.perf_counter()
        res = hutoken.batch_encode(docs, cores)
        dt = min(dt, time.perf_counter() - t)
    okl = None
    if not no_verify:
        want = orc.batch_encode(docs[:VERIFY_DOCS], min(cores, 8))
        okl = res[:VERIFY_DOCS] == want
        if not okl:
            raise SystemExit("PARITY FAILURE: batch_encode ids differ from the oracle")
    nb = int(offs[n_list])
    out["list_api"] = {"value": round(nb / dt / 1e9, 4), "unit": "GB/s", "ms": round(dt * 1e3, 1), "docs": n_list,
                       "verified_vs_oracle": okl,
                       "note": "hutoken_amd.batch_encode(list[str]) -> list[list[int]], CPython object traffic included; best of two "
                               "calls after one untimed call of the same size (buffers grown)"}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--corpus", default="C3", choices=["C2", "C3", "C5"])
    ap.add_argument("--vocab", default="VG", choices=["VG", "VL"],
                    help="VG: GPT-2 shape (the BASELINE metric); VL: Llama/SentencePiece shape -- with --corpus C5 this is "
                         "BASELINE config 5")
    ap.add_argument("--docs", type=int, default=None, help="documents per GPU (default: the corpus size)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="what `value` is at N > 1: weak = the corpus per GPU; strong = ONE corpus cut into N byte-balanced "
                         "shards.  The default run reports weak as `value` and, at N > 1, strong beside it")
    ap.add_argument("--cpu-docs", type=int, default=100_000)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip end_to_end and secondary (N = 1)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only to rehearse the N > 1 code "
                         "path where RCCL cannot run, e.g. two ranks sharing one GPU with HUTK_BENCH_DEVICE=0)")
    ap.add_argument("--force-dist", action="store_true",
                    help="N = 1 only: initialise torch.distributed all the same (world size 1) and run the per-step all-gather "
                         "of the id total, so that the RCCL branch executes on a one-GPU box (profiles/r03_n1_nccl_rehearsal.json)")
    ap.add_argument("--merges", action="store_true",
                    help="the id-keyed merge path (the vocabulary with its merges file, SURVEY 8 f-1) as the main workload")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hutoken_amd encode path has no CPU fallback")
    dev_index = int(os.environ.get("HUTK_BENCH_DEVICE", local_rank))
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    cdev = dev if args.backend == "nccl" else torch.device("cpu")  # where the collectives' tensors live
    dist_on = world > 1 or args.force_dist  # the collectives run (with --force-dist also in a world of one)
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    from hutoken_amd import _capi, data as hdata, sharding, synth
    vp, sp, kw = hdata.vocab_files(args.vocab)
    mp = hdata.merges_file(args.vocab) if args.merges else None
    ctx = _capi.Context(vp, sp, kw["prefix"], kw["is_byte_encoder"], device=dev_index, merges_path=mp)

    n_corpus = args.docs or synth.KINDS[args.corpus][2]
    cores = os.cpu_count() or 1
    gen_threads = max(2, min(32, cores // max(world, 1)))
    # two sets of buffers for the per-step all-gather: it is issued asynchronously and waited for one step later, so that
    # the next step's kernels do not queue behind an 8-byte collective (the id totals are not an input of the encode)
    d_tot = [torch.zeros(1, dtype=torch.int64, device=cdev) for _ in range(2)]
    gathered = [[torch.zeros(1, dtype=torch.int64, device=cdev) for _ in range(world)] for _ in range(2)]

    def sync():
        torch.cuda.synchronize(dev)

    def run_mode(first_doc, n_docs):
        """Timed region of one scaling mode on this rank's documents [first_doc, first_doc + n_docs)."""
        t_gen = time.perf_counter()
        data, offs = synth.corpus(args.corpus, n_docs, first_doc=first_doc, threads=gen_threads)
        t_gen = time.perf_counter() - t_gen
        batch = DeviceBatch(ctx, data, offs, dev)

        pending = []  # handles of the all-gathers in flight (at most two)

        def step():
            batch.run()
            if dist_on:  # the path's one exchange: per-rank id totals
                k = len(pending) & 1
                d_tot[k].copy_(batch.d_oo[n_docs:n_docs + 1])
                pending.append(dist.all_gather(gathered[k], d_tot[k], async_op=True))
                if len(pending) >= 2:
                    pending[-2].wait()  # (its buffers are the next step's)

        def drain():
            if pending:
                pending[-1].wait()
            pending.clear()

        for _ in range(args.warmup):
            step()
        drain()
        sync()
        batch.check_err()
        tile_ms = []
        if dist_on:
            dist.barrier()
        sync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
            tile_ms.append(ctx.last_timing()[0])  # HIP events around k_tiles on the launch stream
        drain()  # (every step's exchange is complete inside the timed region)
        sync()
        if dist_on:
            dist.barrier()
        elapsed = time.perf_counter() - t0
        total_bytes, total_ids = batch.n_bytes, batch.n_ids()
        if dist_on:
            tmax = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            elapsed = float(tmax.item())
            tb = torch.tensor([batch.n_bytes, total_ids], dtype=torch.int64, device=cdev)
            dist.all_reduce(tb)
            total_bytes, total_ids = int(tb[0].item()), int(tb[1].item())
        verified = None
        if not args.no_verify:
            from oracle import oracle as O
            orc = O.Oracle(vp, sp, kw["prefix"], kw["is_byte_encoder"], mp)
            verified = batch.verify(orc, data, offs, cores)
        return dict(batch=batch, data=data, offs=offs, elapsed=elapsed, total_bytes=total_bytes, total_ids=total_ids,
                    tile_ms=sum(tile_ms) / len(tile_ms), verified=verified, t_gen=t_gen)

    # ---- weak: the whole corpus on every rank ------------------------------------------------------------
    modes = {}
    if args.scaling == "weak" or world > 1:
        modes["weak"] = run_mode(rank * n_corpus, n_corpus)
    # ---- strong (BASELINE config 4): ONE corpus, N byte-balanced contiguous shards -------------------------
    if args.scaling == "strong" or world > 1:
        if world == 1:
            modes["strong"] = modes.get("weak") or run_mode(0, n_corpus)
        else:
            if "weak" in modes:  # free the weak batch's buffers first
                modes["weak"].pop("batch"); modes["weak"].pop("data"); modes["weak"].pop("offs")
                torch.cuda.empty_cache()
            lens = synth.lengths(args.corpus, n_corpus, threads=gen_threads)  # lengths only: no rank holds the corpus
            offs_all = np.zeros(n_corpus + 1, dtype=np.int64)
            np.cumsum(lens, out=offs_all[1:])
            first, count = sharding.shard_by_bytes(offs_all, world)[rank]
            modes["strong"] = run_mode(first, count)
            modes["strong"]["shard"] = (int(first), int(count))

    if rank == 0:
        main_mode = args.scaling if args.scaling in modes else "weak"
        m = modes[main_mode]

        def mode_line(mm):
            return {"value": round(mm["total_bytes"] * args.steps / mm["elapsed"] / 1e9, 3), "unit": "GB/s",
                    "ms_per_step": round(mm["elapsed"] / args.steps * 1e3, 4), "total_bytes": mm["total_bytes"],
                    "total_ids": mm["total_ids"], "verified_vs_oracle": mm["verified"]}

        ml = mode_line(m)
        t_tile = m["tile_ms"] / 1e3
        # algorithmic bytes of THIS RANK's k_tiles launch: input + offsets read, ids + per-doc counts written
        rb = modes[main_mode]
        r_bytes = rb["batch"].n_bytes if "batch" in rb else rb["total_bytes"] // world
        r_docs = rb["batch"].n_docs if "batch" in rb else n_corpus
        r_ids = rb["batch"].n_ids() if "batch" in rb else rb["total_ids"] // world
        b_alg = r_bytes + 8 * (r_docs + 1) + 4 * r_ids + 4 * r_docs
        achieved = b_alg / t_tile / 1e9
        tr, tr_src = committed_profile(r_bytes, "traffic")
        iss, iss_src = committed_profile(r_bytes, "issue")
        shape = "mixed UTF-8" if args.corpus == "C3" else "ASCII" if args.corpus == "C2" else "Hungarian-like UTF-8"
        vdesc = "VG (GPT-2 shape, 50257 entries)" if args.vocab == "VG" else \
                "VL (Llama/SentencePiece shape, 32000 entries, prefix U+2581, is_byte_encoder=False)"
        line = {
            "metric": "GB/s input text encoded (GPT-2 vocab) at 1/2/4/8 GPUs; bit-exact ids",
            "value": ml["value"], "unit": "GB/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ml["ms_per_step"], "higher_is_better": True,
            "scaling": main_mode, "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": f"{args.corpus}: {n_corpus} synthetic docs "
                                   f"{'per GPU' if main_mode == 'weak' else 'in all, cut into byte-balanced shards'} "
                                   f"({r_bytes / 1e6:.1f} MB on rank 0, mean {r_bytes / max(r_docs, 1):.0f} B, {shape}), "
                                   f"vocab {vdesc}"
                                   f"{', id-keyed merge path (merges file)' if args.merges else ''}, "
                                   "device-resident packed I/O",
                       "docs_per_gpu": r_docs, "bytes_per_gpu": r_bytes, "ids_per_gpu": r_ids,
                       "parallelism": f"documents sharded over {world} GPU(s), all-gather of id totals"},
            "roofline": {"bound": "hbm", "kernel": "k_tiles", "achieved": round(achieved, 2),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
                         "traffic": tr["traffic_bytes_per_launch"] if tr else None, "traffic_source": tr_src,
                         "kernel_ms": round(t_tile * 1e3, 4), "algorithmic_bytes": b_alg},
            "verified_vs_oracle": m["verified"],
            "gen_s": round(m["t_gen"], 2),
        }
        # Where a resident wavefront's cycles go (committed PMC profile of this workload).  Every share is a quotient of two
        # SQ counters of the same unit (quad-cycles of one wavefront, summed over the wavefronts) and cannot exceed 1.  No
        # "issue roofline": round 3's priced SQ_INSTS_VALU at an assumed 2..4 cycles and its upper end passed 1; round 4
        # measured instead that 40 % fewer front-end instructions leave the kernel's time where it was
        # (profiles/r04_slim_rounds_ab.txt): issue does not bind it.
        if iss:
            line["roofline"]["wavefront_cycles"] = {
                "valu_wave_insts": iss.get("valu_insts_per_launch"), "salu_wave_insts": iss.get("salu_insts_per_launch"),
                "lds_wave_insts": iss.get("lds_insts_per_launch"),
                "parked_at_waitcnt_or_barrier": iss.get("wait_any_frac"), "stalled_at_issue": iss.get("wait_inst_frac"),
                "executing_valu": iss.get("valu_active_frac"),
                "lds_bank_conflict_frac": iss.get("lds_bank_conflict_frac"), "ta_busy_frac": iss.get("ta_busy_frac"),
                "unit": "shares of SQ_WAVE_CYCLES (per-wavefront quad-cycles, summed); the last two: of the LDS-active cycles / of the launch",
                "source": iss_src}
        # What binds: the latency of chains of dependent table gathers (measured, profiles/r03_latency.json).
        try:
            lat = json.load(open(os.path.join(ROOT, "profiles", "r03_latency.json")))
            line["roofline"]["latency"] = {
                "bound": lat["bound"], "occupancy_ab_gbs": lat["occupancy_ab_gbs"],
                "wait_frac_profiled": iss.get("wait_any_frac") if iss else None,
                "mean_l1_to_l2_read_latency_cycles": lat["memory_latency_pmc"]["mean_l1_to_l2_read_latency_cycles"],
                "mean_l2_miss_latency_cycles": lat["memory_latency_pmc"]["mean_l2_miss_to_fabric_latency_cycles"],
                "l2_hit_rate": lat["memory_latency_pmc"]["l2_hit_rate"], "source": "profiles/r03_latency.json"}
        except Exception:
            pass
        if world > 1:
            line["weak"] = mode_line(modes["weak"])
            line["strong"] = dict(mode_line(modes["strong"]),
                                  workload=f"{args.corpus}: {n_corpus} docs in all, {world} byte-balanced shards "
                                           "(BASELINE config 4)",
                                  rank0_shard_docs=modes["strong"].get("shard", (0, n_corpus))[1])
        if world == 1 and not args.no_extras and "batch" in m:
            from oracle import oracle as O
            orc = O.Oracle(vp, sp, kw["prefix"], kw["is_byte_encoder"], mp)
            try:
                line["end_to_end"] = end_to_end(ctx, orc, m["data"], m["offs"], vp, sp, kw, dev_index, cores, args.no_verify)
            except SystemExit:
                raise
            except Exception as e:  # a measurement beside the metric must not lose the line
                line["end_to_end"] = {"error": repr(e)}
            if isinstance(line["end_to_end"], dict) and line["end_to_end"].get("parity_failures"):
                line["parity_failures"] = line["end_to_end"].pop("parity_failures")
            m.pop("batch"); torch.cuda.empty_cache()
            try:
                line["secondary"] = secondary_runs(dev, dev_index, cores, args.no_verify)
            except SystemExit:
                raise
            except Exception as e:
                line["secondary"] = {"error": repr(e)}
        if world == 1 and not args.no_cpu:
            line["cpu_baseline"] = cpu_baseline(vp, sp, kw, args.corpus, min(args.cpu_docs, n_corpus), cores, mp)
        print(json.dumps(line), flush=True)
        if line.get("parity_failures"):
            sys.exit(3)
    if dist_on:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
