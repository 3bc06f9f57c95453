#!/usr/bin/env python3
"""One byte-balanced shard of C3's 1 M documents (BASELINE config 4's per-GPU share), device-resident, 20 steps -- to be run under
`rocprofv3 --kernel-trace` (tools/shard_trace.sh).   shard_steps.py N_SHARDS"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from hutoken_amd import _capi, data
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
vp, sp, kw = data.vocab_files("VG")
ctx = _capi.Context(vp, sp, kw["prefix"], kw["is_byte_encoder"])
d, o = bench.c3_shard(n) if n > 1 else __import__("hutoken_amd.synth", fromlist=["x"]).corpus("C3", 1_000_000)
dev = torch.device("cuda", 0)
b = bench.DeviceBatch(ctx, d, o, dev)
for _ in range(25):
    b.run()
torch.cuda.synchronize()
print(len(d), "bytes", len(o) - 1, "docs", b.n_ids(), "ids")
