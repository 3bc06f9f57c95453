"""The reference's own benchmark shape (scripts/benchmark.py:51-104, scripts/threaded_benchmark.sh:3-15): ONE document of
100 MB / 1 GB, whole and cut into 64 whitespace-aligned pieces -- a handful of giant documents instead of a million small
ones.  Every id of the pieces against the oracle; the whole document's ids = the pieces' ids back to back (the cuts are
chosen so that this holds, synth.whitespace_chunks).  GPU only."""
import hashlib
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _encode_device(ctx, d, o):
    import torch
    dev = torch.device("cuda", 0)
    n_docs = len(o) - 1
    db, do = torch.from_numpy(d).to(dev), torch.from_numpy(o).to(dev)
    cap = ctx.ids_capacity(len(d), n_docs)
    ids = torch.empty(cap, dtype=torch.int32, device=dev)
    oo = torch.empty(n_docs + 1, dtype=torch.int64, device=dev)
    err = torch.zeros(1, dtype=torch.int32, device=dev)
    st = torch.zeros(n_docs, dtype=torch.int32, device=dev)
    ctx.encode_device(db.data_ptr(), do.data_ptr(), n_docs, len(d), ids.data_ptr(), cap, oo.data_ptr(), st.data_ptr(),
                      err.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert int(err.item()) == 0
    oo_h = oo.cpu().numpy()
    out = ids[: int(oo_h[-1])].cpu().numpy()
    del ids, db
    torch.cuda.empty_cache()
    return out, oo_h


@pytest.mark.parametrize("mbytes", [100, 1000])
def test_one_giant_document_whole_and_in_64_pieces(mbytes, vg_files, oracle_mod):
    from hutoken_amd import _capi, synth
    vp, sp, kw = vg_files
    ctx = _capi.Context(vp, sp, kw["prefix"], kw["is_byte_encoder"], device=0)
    d, o1 = synth.big_document(mbytes * 1_000_000)
    assert len(d) >= mbytes * 1_000_000 and len(o1) == 2
    o64 = synth.whitespace_chunks(d, 64)
    assert len(o64) == 65
    ids_whole, oo_whole = _encode_device(ctx, d, o1)
    ids_64, oo_64 = _encode_device(ctx, d, o64)
    # the pieces against the oracle: every id (hashed per piece), the offsets
    orc = oracle_mod.Oracle(vp, sp, kw["prefix"], kw["is_byte_encoder"])
    ids_o, oo_o, _ = orc.encode_packed(d, o64, min(64, os.cpu_count() or 8))
    assert np.array_equal(oo_o, oo_64)
    for k in range(64):
        a, b = int(oo_o[k]), int(oo_o[k + 1])
        assert hashlib.sha256(ids_o[a:b].tobytes()).digest() == hashlib.sha256(ids_64[a:b].tobytes()).digest(), f"piece {k}"
    # the whole document: the same ids, one document
    assert int(oo_whole[1]) == int(oo_64[-1])
    assert np.array_equal(ids_whole, ids_64)
    ctx.close()
