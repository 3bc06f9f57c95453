"""The reference ends a document at a word of more than 262144 bytes and reports nothing (src/core.c:402-407 sets
error_msg, src/core.c:503 clears it).  On the device that is k_cut, and it has to hold whatever the over-long word is made
of: one run of letters (an exception word whose end the ends pass looks for) or a run of three-byte characters that the
seam map cuts into thousands of short words.  Both entry points -- hutk_encode_batch on host buffers and
hutk_encode_batch_device, the form bench.py times -- give the oracle's ids, offsets and status.  Needs a real MI355X."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

LIMIT = 262144


def device_form(ctx, d, o):
    import torch
    dev = torch.device("cuda", 0)
    n_docs, n_bytes = len(o) - 1, int(o[-1])
    d_bytes = torch.from_numpy(np.array(d, copy=True)).to(dev)
    d_offs = torch.from_numpy(np.ascontiguousarray(o)).to(dev)
    cap = ctx.ids_capacity(n_bytes, n_docs)
    d_ids = torch.empty(cap, dtype=torch.int32, device=dev)
    d_oo = torch.empty(n_docs + 1, dtype=torch.int64, device=dev)
    d_st = torch.zeros(n_docs, dtype=torch.int32, device=dev)
    d_err = torch.zeros(1, dtype=torch.int32, device=dev)
    ctx.encode_device(d_bytes.data_ptr(), d_offs.data_ptr(), n_docs, n_bytes, d_ids.data_ptr(), cap,
                      d_oo.data_ptr(), d_st.data_ptr(), d_err.data_ptr(), torch.cuda.current_stream(dev).cuda_stream)
    torch.cuda.synchronize(dev)
    oo = d_oo.cpu().numpy()
    return d_ids[: int(oo[-1])].cpu().numpy(), oo, d_st.cpu().numpy(), int(d_err.item())


def quiet_char(orc, seam):
    """A three-byte character that the vocabulary neither merges inside nor joins to a copy of itself: a run of it costs
    the oracle's quadratic loop nothing, and every character boundary of the run is a seam."""
    for cp in list(range(0x4E00, 0x4E00 + 3000)) + list(range(0x3040, 0x30FF)):
        c = chr(cp).encode("utf-8")
        if len(orc.encode_bytes(b"\n" + c + c)[0]) == 7 and not (int(seam[c[2]]) >> (c[0] & 31)) & 1:
            return c
    pytest.skip("the vocabulary merges every three-byte character tried")


def test_over_long_words_cut_their_documents_on_both_entry_points():
    from hutoken_amd import _capi, data
    from oracle import oracle as O
    vp, sp, kw = data.vocab_files("VG")
    ctx = _capi.Context(vp, sp, kw["prefix"], kw["is_byte_encoder"])
    orc = O.Oracle(vp, sp, kw["prefix"], kw["is_byte_encoder"])
    seam, on = ctx.seam_map()
    assert on
    c = quiet_char(orc, seam)
    n_ok = (LIMIT - 1) // 3          # a leading space and n_ok characters: at most 262144 bytes
    n_cut = LIMIT // 3 + 1           # more than 262144 bytes without the space
    filler = b"the quick brown fox jumps over the lazy dog. " * 30
    docs = [
        b"before the long ones",
        b"cjk just below the limit: " + b" " + c * n_ok + b" and on it goes",          # passes
        b"cjk over the limit: " + c * n_cut + b" never seen",                           # cut in front of the run
        filler,
        c * n_cut + b" the run is the document's first word",                           # nothing is kept
        b"letters over the limit " + b"x" * (LIMIT + 1) + b" dropped",                  # one exception word
        b"two of them: " + c * n_cut + b" middle " + b"y" * (LIMIT + 7) + b" end",      # the first one decides
        b"mixed signs and characters " + (b"." + c) * (LIMIT // 4 + 1) + b" gone",      # one run of class OTHER
        filler + b" after",
        b"",
        b"last",
    ]
    data_, offs = O.pack(docs)
    ids_o, oo_o, st_o = orc.encode_packed(data_, offs, 8)
    assert st_o.tolist() == [0, 0, 1, 0, 1, 1, 1, 1, 0, 0, 0]
    ids_h, oo_h, st_h, rc = ctx.encode_packed(data_, offs)
    assert rc == 0
    assert st_h.tolist() == st_o.tolist()
    assert np.array_equal(oo_h, oo_o) and np.array_equal(ids_h, ids_o)
    ids_d, oo_d, st_d, err = device_form(ctx, data_, offs)
    assert err in (0, 9)  # HUTK_E_WORD_TOO_LARGE is a note
    assert st_d.tolist() == st_o.tolist()
    assert np.array_equal(oo_d, oo_o) and np.array_equal(ids_d, ids_o)
    # without the seam map every one of them is an exception word: the same
    ctx.close()


def test_the_same_without_seams(monkeypatch):
    from hutoken_amd import _capi, data
    from oracle import oracle as O
    monkeypatch.setenv("HUTK_NO_SEAM", "1")
    vp, sp, kw = data.vocab_files("VG")
    ctx = _capi.Context(vp, sp, kw["prefix"], kw["is_byte_encoder"])
    orc = O.Oracle(vp, sp, kw["prefix"], kw["is_byte_encoder"])
    docs = [b"a b", b"letters over the limit " + b"x" * (LIMIT + 1) + b" dropped", b"kept " + b"x" * (LIMIT - 1) + b" kept", b"z"]
    data_, offs = O.pack(docs)
    ids_o, oo_o, st_o = orc.encode_packed(data_, offs, 4)
    ids_d, oo_d, st_d, err = device_form(ctx, data_, offs)
    assert st_d.tolist() == st_o.tolist() == [0, 1, 0, 0]
    assert np.array_equal(oo_d, oo_o) and np.array_equal(ids_d, ids_o)
    ctx.close()
