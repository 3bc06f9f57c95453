"""The N>1 path on CPU: two gloo ranks shard one batch by documents, encode their shard
(the oracle stands in for the GPU path, which needs a GPU), all-gather the id totals, and
the stitched result equals the single-process encode."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

import helpers as H


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, vp, sp, out_dir):
    sys.path.insert(0, H.ROOT)
    import torch.distributed as dist
    from hutoken_amd import sharding, synth
    from oracle import oracle as O
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    data, offs = synth.corpus("C3", 400)
    first, count = sharding.shard_by_bytes(offs, world)[rank]
    ld, lo = sharding.local_view(data, offs, first, count)
    orc = O.Oracle(vp, sp, None, True)
    ids, oo, st = orc.encode_packed(ld, lo, 2)
    totals = sharding.gather_id_totals(int(oo[-1]))
    base = sharding.global_id_base(totals, rank)
    np.savez(os.path.join(out_dir, f"r{rank}.npz"), ids=ids, oo=oo + base, first=first, count=count,
             totals=np.array(totals))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_stitch_to_the_single_process_result(tmp_path, vg_files, oracle_mod):
    from hutoken_amd import sharding, synth
    vp, sp, kw = vg_files
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, vp, sp, str(tmp_path)), nprocs=world, join=True)
    data, offs = synth.corpus("C3", 400)
    ids_ref, oo_ref, _ = oracle_mod.Oracle(vp, sp, None, True).encode_packed(data, offs, 4)
    parts = [np.load(os.path.join(str(tmp_path), f"r{r}.npz")) for r in range(world)]
    assert int(sum(p["count"] for p in parts)) == 400
    assert list(parts[0]["totals"]) == list(parts[1]["totals"])
    assert int(sum(parts[0]["totals"])) == int(oo_ref[-1])
    assert np.array_equal(np.concatenate([p["ids"] for p in parts]), ids_ref)
    stitched = np.concatenate([parts[0]["oo"][:-1], parts[1]["oo"]])
    assert np.array_equal(stitched, oo_ref)


def test_shard_helpers():
    from hutoken_amd import sharding
    offs = np.array([0, 10, 10, 30, 100, 101, 200], dtype=np.int64)
    for world in (1, 2, 3, 4, 8):
        parts = sharding.shard_by_bytes(offs, world)
        assert sum(c for _f, c in parts) == 6 and parts[0][0] == 0
        assert all(parts[i][0] + parts[i][1] == parts[i + 1][0] for i in range(world - 1))
        parts = sharding.shard_by_docs(6, world)
        assert sum(c for _f, c in parts) == 6
