"""The FULL-SIZE BASELINE configurations, every document, against the reference's own outputs
(tests/golden/g7_full.json: counts and sha256 per block of 100k documents, produced by the compiled reference in
tools/make_golden.py) -- through ONE hutk_encode_batch_device launch sequence on device-resident buffers, which is
exactly what bench.py times, and through the chunked host path.  Needs a real MI355X."""
import hashlib
import json
import os

import numpy as np
import pytest

import helpers as H

pytestmark = pytest.mark.gpu


def g7_cases():
    with open(os.path.join(H.GOLDEN_DIR, "g7_full.json")) as f:
        return json.load(f)


def case_id(g):
    return "%s-%s%s" % (g["vocab"], g["corpus"], "-merges" if g["merges"] else "")


def encode_one_launch(ctx, d, o):
    """Device-resident buffers, one call of hutk_encode_batch_device on torch's current stream."""
    import torch
    dev = torch.device("cuda", 0)
    n_docs, n_bytes = len(o) - 1, int(o[-1])
    d_bytes = torch.from_numpy(d).to(dev)
    d_offs = torch.from_numpy(o).to(dev)
    cap = ctx.ids_capacity(n_bytes, n_docs)
    d_ids = torch.empty(cap, dtype=torch.int32, device=dev)
    d_oo = torch.empty(n_docs + 1, dtype=torch.int64, device=dev)
    d_st = torch.zeros(n_docs, dtype=torch.int32, device=dev)
    d_err = torch.zeros(1, dtype=torch.int32, device=dev)
    ctx.encode_device(d_bytes.data_ptr(), d_offs.data_ptr(), n_docs, n_bytes, d_ids.data_ptr(), cap,
                      d_oo.data_ptr(), d_st.data_ptr(), d_err.data_ptr(), torch.cuda.current_stream(dev).cuda_stream)
    torch.cuda.synchronize(dev)
    assert int(d_err.item()) == 0
    oo = d_oo.cpu().numpy()
    ids = d_ids[: int(oo[-1])].cpu().numpy()
    assert not d_st.any().item()
    return ids, oo


def check_against(g, ids, oo):
    assert int(oo[-1]) == g["n_ids"] == len(ids)
    blocks = H.block_hashes(ids, oo, g["block_docs"])
    bad = [i for i, (a, b) in enumerate(zip(blocks, g["blocks"])) if a != b]
    assert not bad, "blocks of %d documents that differ from the reference: %s" % (g["block_docs"], bad)
    assert len(blocks) == len(g["blocks"])
    assert hashlib.sha256(ids.astype("<i4", copy=False).tobytes()).hexdigest() == g["sha256"]


@pytest.mark.parametrize("g", g7_cases(), ids=case_id)
def test_full_size_one_launch_equals_the_reference(g):
    from hutoken_amd import _capi, data, synth
    vp, sp, kw = data.vocab_files(g["vocab"])
    mp = data.merges_file(g["vocab"]) if g["merges"] else None
    ctx = _capi.Context(vp, sp, kw["prefix"], kw["is_byte_encoder"], merges_path=mp)
    d, o = synth.corpus(g["corpus"], g["n_docs"])
    assert int(o[-1]) == g["n_bytes"]
    assert hashlib.sha256(d.tobytes()).hexdigest() == g["corpus_sha256"]
    ids, oo = encode_one_launch(ctx, d, o)
    check_against(g, ids, oo)
    # the chunked host path (hutk_encode_batch: copies and kernels overlapped chunk by chunk) gives the same
    ids_h, oo_h, st, rc = ctx.encode_packed(d, o)
    assert rc == 0 and not st.any()
    assert np.array_equal(oo_h, oo) and np.array_equal(ids_h, ids)
    ctx.close()


@pytest.mark.parametrize("g", [g for g in g7_cases() if g["vocab"] == "VG"], ids=case_id)
def test_full_size_without_the_whole_word_table(g, monkeypatch):
    """HUTK_NO_WORD_TABLE=1: every word of the real-shape vocabulary goes through the merge loop (with the table
    87 % of C3's words never reach it)."""
    from hutoken_amd import _capi, data, synth
    monkeypatch.setenv("HUTK_NO_WORD_TABLE", "1")
    vp, sp, kw = data.vocab_files(g["vocab"])
    mp = data.merges_file(g["vocab"]) if g["merges"] else None
    ctx = _capi.Context(vp, sp, kw["prefix"], kw["is_byte_encoder"], merges_path=mp)
    assert ctx.table_stats()["n_word_entries"] == 0
    d, o = synth.corpus(g["corpus"], g["n_docs"])
    ids, oo = encode_one_launch(ctx, d, o)
    check_against(g, ids, oo)
    ctx.close()


@pytest.mark.parametrize("no_table", [False, True], ids=["table", "no-table"])
@pytest.mark.parametrize("g", g7_cases(), ids=case_id)
def test_full_size_without_the_seam_map(g, no_table, monkeypatch):
    """HUTK_NO_SEAM=1: words are not cut where no merge can span (hutk_loader.cpp, seam_from_pairs): every run of CJK
    characters is one long merge-loop word again.  The same ids -- the cut is invisible by construction -- also
    without the whole-word table."""
    from hutoken_amd import _capi, data, synth
    monkeypatch.setenv("HUTK_NO_SEAM", "1")
    if no_table:
        if g["vocab"] != "VG":
            pytest.skip("one vocabulary is enough for the combination")
        monkeypatch.setenv("HUTK_NO_WORD_TABLE", "1")
    vp, sp, kw = data.vocab_files(g["vocab"])
    mp = data.merges_file(g["vocab"]) if g["merges"] else None
    ctx = _capi.Context(vp, sp, kw["prefix"], kw["is_byte_encoder"], merges_path=mp)
    assert ctx.seam_map()[1] is False
    d, o = synth.corpus(g["corpus"], g["n_docs"])
    ids, oo = encode_one_launch(ctx, d, o)
    check_against(g, ids, oo)
    ctx.close()
