#!/usr/bin/env python3
"""bench.py -- GB/s of input text encoded (GPT-2-shaped 50257-entry vocab, bit-exact
ids) on N MI355X, the metric BASELINE.json names.

  python bench.py --gpus 1 --steps 10 --warmup 3
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

A step = one pass of the batch encode path (hutk_encode_batch_device: every kernel
of the pipeline) over one batch that is ALREADY RESIDENT IN HBM: packed UTF-8 bytes
+ int64 offsets in, int32 ids + int64 offsets out, all device buffers.
Workload at every N: corpus C3, 1,000,000 synthetic documents per GPU, mean 512 B,
mixed UTF-8 (hutoken_amd/csrc/hutk_synth.c), vocabulary VG (data/vg50257_*).
Documents shard trivially: rank r encodes documents [r*1M, (r+1)*1M) of the
generator with its own context; the only collective is an all-gather of the
per-rank id totals (RCCL), inside every timed step.  Scaling is therefore weak.

The JSON line also carries
  roofline      algorithmic HBM bytes of the dominant kernel (k_tiles) / its
                mean duration from HIP events recorded around it on the launch
                stream in every timed step, against 8 TB/s
  cpu_baseline  the reference itself (oracle/_ref, compiled from the reference
                sources; kind "reference") or this repo's C restatement (kind
                "port") timed on the host cores, on a bounded sample of the same
                workload, rank 0 at N=1 only.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec


def cpu_baseline(vp, sp, kw, corpus_name, n_sample, cores, mp=None):
    """Times the CPU path on the first n_sample documents of the workload."""
    import numpy as np
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


def committed_traffic(n_bytes):
    """HBM-side bytes of one k_tiles launch from the PMC passes of the last committed profile of THIS workload
    (profiles/<tag>_traffic.json, written by tools/summarize_rocprof.py from separate `rocprofv3 --pmc` runs of
    the same bench command: counters cannot be collected inside a timed run).  None when there is no profile
    of a batch of this size."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_traffic.json"))):
        try:
            t = json.load(open(f))
        except Exception:
            continue
        if t.get("workload_bytes") == n_bytes:
            best = (float(t["traffic_bytes_per_launch"]), os.path.relpath(f, ROOT))
    return best if best else (None, None)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--corpus", default="C3", choices=["C2", "C3", "C5"])
    ap.add_argument("--docs", type=int, default=None, help="documents per GPU (default: the corpus size)")
    ap.add_argument("--cpu-docs", type=int, default=100_000)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only to rehearse the N > 1 code "
                         "path where RCCL cannot run, e.g. two ranks sharing one GPU with HUTK_BENCH_DEVICE=0)")
    ap.add_argument("--merges", action="store_true",
                    help="secondary configuration: the id-keyed merge path (VG with its merges file, SURVEY 8 f-1); "
                         "the default and BASELINE metric is the string-keyed path")
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
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")

    from hutoken_amd import _capi, data as hdata, synth
    vp, sp, kw = hdata.vocab_files("VG")
    mp = hdata.merges_file("VG") if args.merges else None
    ctx = _capi.Context(vp, sp, kw["prefix"], kw["is_byte_encoder"], device=dev_index, merges_path=mp)

    n_docs = args.docs or synth.KINDS[args.corpus][2]
    cores = os.cpu_count() or 1
    t_gen = time.perf_counter()
    data, offs = synth.corpus(args.corpus, n_docs, first_doc=rank * n_docs,
                              threads=max(2, min(32, cores // max(world, 1))))
    t_gen = time.perf_counter() - t_gen
    n_bytes = int(offs[-1])

    d_bytes = torch.from_numpy(data).to(dev)
    d_offs = torch.from_numpy(offs).to(dev)
    cap = ctx.ids_capacity(n_bytes, n_docs)
    d_ids = torch.empty(cap, dtype=torch.int32, device=dev)
    d_oo = torch.empty(n_docs + 1, dtype=torch.int64, device=dev)
    d_err = torch.zeros(1, dtype=torch.int32, device=dev)
    d_tot = torch.zeros(1, dtype=torch.int64, device=cdev)
    gathered = [torch.zeros(1, dtype=torch.int64, device=cdev) for _ in range(world)]
    stream = torch.cuda.current_stream(dev).cuda_stream

    def step():
        ctx.encode_device(d_bytes.data_ptr(), d_offs.data_ptr(), n_docs, n_bytes, d_ids.data_ptr(), cap,
                          d_oo.data_ptr(), 0, d_err.data_ptr(), stream)
        if world > 1:  # the path's one exchange: per-rank id totals
            d_tot.copy_(d_oo[n_docs:n_docs + 1])
            dist.all_gather(gathered, d_tot)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize(dev)
    if int(d_err.item()) != 0:
        raise SystemExit(f"device-side error {int(d_err.item())}")

    tile_ms = []
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        tile_ms.append(ctx.last_timing()[0])  # HIP events around k_tiles on the launch stream
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        tb = torch.tensor([n_bytes], dtype=torch.int64, device=cdev)
        dist.all_reduce(tb)
        total_bytes = int(tb.item())
    else:
        total_bytes = n_bytes
    n_ids = int(d_oo[n_docs].item())

    # parity spot check against the oracle on the first documents of this rank's shard
    verified = None
    if not args.no_verify:
        from oracle import oracle as O
        orc = O.Oracle(vp, sp, kw["prefix"], kw["is_byte_encoder"], mp)
        k = min(2000, n_docs)
        ids_o, oo_o, _ = orc.encode_packed(data[: int(offs[k])], offs[: k + 1], min(cores, 8))
        oo_g = d_oo[: k + 1].cpu().numpy()
        ids_g = d_ids[: int(oo_g[k])].cpu().numpy()
        verified = bool(np.array_equal(oo_o, oo_g) and np.array_equal(ids_o, ids_g))
        if not verified:
            raise SystemExit("PARITY FAILURE: GPU ids differ from the oracle")

    if rank == 0:
        ms_step = elapsed / args.steps * 1e3
        value = total_bytes * args.steps / elapsed / 1e9
        t_tile = sum(tile_ms) / len(tile_ms) / 1e3
        b_alg = n_bytes + 8 * (n_docs + 1) + 4 * n_ids + 4 * n_docs
        achieved = b_alg / t_tile / 1e9
        traffic, traffic_src = committed_traffic(n_bytes)
        line = {
            "metric": "GB/s input text encoded (GPT-2 vocab) at 1/2/4/8 GPUs; bit-exact ids",
            "value": round(value, 3), "unit": "GB/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_step, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": f"{args.corpus}: {n_docs} synthetic docs per GPU "
                                   f"({n_bytes / 1e6:.1f} MB, mean {n_bytes / n_docs:.0f} B, "
                                   f"{'mixed UTF-8' if args.corpus != 'C2' else 'ASCII'}), "
                                   "vocab VG (GPT-2 shape, 50257 entries)"
                                   f"{', id-keyed merge path (merges file)' if args.merges else ''}, "
                                   "device-resident packed I/O",
                       "docs_per_gpu": n_docs, "bytes_per_gpu": n_bytes, "ids_per_gpu": n_ids,
                       "parallelism": f"documents sharded over {world} GPU(s), all-gather of id totals"},
            "roofline": {"bound": "hbm", "kernel": "k_tiles", "achieved": round(achieved, 2),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
                         "traffic": traffic, "traffic_source": traffic_src, "kernel_ms": round(t_tile * 1e3, 4),
                         "algorithmic_bytes": b_alg},
            "verified_vs_oracle": verified,
            "gen_s": round(t_gen, 2),
        }
        if world == 1 and not args.no_cpu:
            line["cpu_baseline"] = cpu_baseline(vp, sp, kw, args.corpus, min(args.cpu_docs, n_docs), cores, mp)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
