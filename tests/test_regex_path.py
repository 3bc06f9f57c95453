"""The regex pre-token path (initialize(pattern=...), reference src/core.c:350-360, 372-378, 392-400): the oracle's
restatement and the product (host regexec -> word-boundary bitmaps -> GPU pretokenizer + merge loop) against
tests/golden/g9_regex_path.json, the outputs of the compiled reference (tools/make_golden_g8.py --g9)."""
import hashlib
import json
import locale
import os
import random
import sys

import numpy as np
import pytest

import helpers as H

sys.path.insert(0, os.path.join(H.ROOT, "tools"))
import make_golden_g8 as G  # noqa: E402  (patterns and the seeded texts only)


@pytest.fixture(scope="module")
def g9():
    with open(os.path.join(H.GOLDEN_DIR, "g9_regex_path.json")) as f:
        g = json.load(f)
    if locale.setlocale(locale.LC_CTYPE, None) != g["lc_ctype"]:
        pytest.skip("POSIX regex matching depends on LC_CTYPE; the fixture was made under " + g["lc_ctype"])
    return g


def digest(res):
    h = hashlib.sha256()
    for ids in res:
        h.update(json.dumps(ids).encode())
    return h.hexdigest()


def check(case, res):
    assert res[:len(case["first"])] == case["first"], case["pattern"]
    assert sum(len(x) for x in res) == case["n_ids"], case["pattern"]
    assert digest(res) == case["sha256"], case["pattern"]


def test_oracle_equals_the_reference(g9, tmp_path, oracle_mod):
    ents, sp = H.random_byte_vocab(11, n_merges=2000)
    vp, spath = H.write_vocab(tmp_path, "g9", ents, sp)
    texts = G.g9_texts(9000, 600)
    assert [m["pattern"] for m in g9["mid"]] == G.G9_PATTERNS
    for case in g9["mid"]:
        orc = oracle_mod.Oracle(vp, spath, None, True, pattern=case["pattern"])
        check(case, orc.batch_encode(texts, 4))
    with pytest.raises(ValueError):
        oracle_mod.Oracle(vp, spath, None, True, pattern="(")


def test_oracle_on_the_vg_corpus(g9, vg_files, oracle_mod):
    from hutoken_amd import synth
    vp, sp, kw = vg_files
    orc = oracle_mod.Oracle(vp, sp, kw["prefix"], kw["is_byte_encoder"], pattern=g9["vg"]["pattern"])
    d, o = synth.corpus("C3", g9["vg"]["n_docs"])
    ids, oo, st = orc.encode_packed(d, o, 8)
    check(g9["vg"], [ids[oo[i]:oo[i + 1]].tolist() for i in range(len(o) - 1)])


@pytest.mark.gpu
def test_gpu_equals_the_reference(g9, tmp_path, vg_files):
    from hutoken_amd import _capi, synth
    from oracle import oracle as O
    ents, sp = H.random_byte_vocab(11, n_merges=2000)
    vp, spath = H.write_vocab(tmp_path, "g9", ents, sp)
    texts = G.g9_texts(9000, 600)
    data, offs = O.pack(texts)
    ctx = _capi.Context(vp, spath, None, True)
    for case in g9["mid"]:
        ctx.set_pattern(case["pattern"])
        ids, oo, st, rc = ctx.encode_packed(data, offs)
        assert rc == 0
        check(case, [ids[oo[i]:oo[i + 1]].tolist() for i in range(len(texts))])
        for t in texts[:5]:  # one document per call (hutk_encode)
            assert ctx.encode_one(t.encode("utf-8"))[0] == case["first"][texts.index(t)]
    ctx.set_pattern(None)  # back to the hand-written splitter
    orc = O.Oracle(vp, spath, None, True)
    ids, oo, st, rc = ctx.encode_packed(data, offs)
    assert [ids[oo[i]:oo[i + 1]].tolist() for i in range(len(texts))] == orc.batch_encode(texts, 4)
    # the device-resident entry point brings the bytes down for libc's regexec and encodes on the device buffers
    import torch
    dev = torch.device("cuda", 0)
    ctx.set_pattern("[a-z]+")
    orc2 = O.Oracle(vp, spath, None, True, pattern="[a-z]+")
    n = len(texts)
    d_b, d_o = torch.from_numpy(np.array(data, copy=True)).to(dev), torch.from_numpy(np.array(offs, copy=True)).to(dev)
    cap = ctx.ids_capacity(len(data), n)
    d_ids = torch.empty(cap, dtype=torch.int32, device=dev)
    d_oo = torch.empty(n + 1, dtype=torch.int64, device=dev)
    d_st = torch.zeros(n, dtype=torch.int32, device=dev)
    d_err = torch.zeros(1, dtype=torch.int32, device=dev)
    ctx.encode_device(d_b.data_ptr(), d_o.data_ptr(), n, len(data), d_ids.data_ptr(), cap, d_oo.data_ptr(), d_st.data_ptr(),
                      d_err.data_ptr(), torch.cuda.current_stream(dev).cuda_stream)
    torch.cuda.synchronize(dev)
    oo2 = d_oo.cpu().numpy()
    ids2 = d_ids[: int(oo2[-1])].cpu().numpy()
    assert int(d_err.item()) == 0
    assert [ids2[oo2[i]:oo2[i + 1]].tolist() for i in range(n)] == orc2.batch_encode(texts, 4)
    # VG x C3 with the POSIX form of the GPT-2 pattern
    vp, sp, kw = vg_files
    ctx = _capi.Context(vp, sp, kw["prefix"], kw["is_byte_encoder"])
    ctx.set_pattern(g9["vg"]["pattern"])
    d, o = synth.corpus("C3", g9["vg"]["n_docs"])
    ids, oo, st, rc = ctx.encode_packed(d, o)
    assert rc == 0
    check(g9["vg"], [ids[oo[i]:oo[i + 1]].tolist() for i in range(len(o) - 1)])


@pytest.mark.gpu
def test_gpu_against_the_oracle_long_words_and_gaps(tmp_path, g9):
    """Patterns that make very long words (whole documents: the exception kernels find their ends in the bitmap), drop
    most of the text, or match nothing at all."""
    from hutoken_amd import _capi
    from oracle import oracle as O
    ents, sp = H.random_byte_vocab(12, n_merges=1500)
    vp, spath = H.write_vocab(tmp_path, "rx", ents, sp)
    rng = random.Random(77)
    texts = [H.random_text(rng, max_words=rng.choice([5, 40, 400])) for _ in range(300)] + ["", " ", "zzz", "a" * 5000]
    data, offs = O.pack(texts)
    ctx = _capi.Context(vp, spath, None, True)
    for pat in [".+", "[^.]+", "[a-z]{3}", "q", "[ ]?[[:alpha:]]+", "(.|\n)+", "[a-z ]+"]:
        ctx.set_pattern(pat)
        orc = O.Oracle(vp, spath, None, True, pattern=pat)
        ids, oo, st, rc = ctx.encode_packed(data, offs)
        want = orc.batch_encode(texts, 4)
        got = [ids[oo[i]:oo[i + 1]].tolist() for i in range(len(texts))]
        assert rc == 0 and got == want, pat


@pytest.mark.gpu
def test_gpu_pattern_together_with_a_prefix(tmp_path):
    """pattern= and prefix= (core.c:362-366, 420-451): the prefix goes with a document's FIRST MATCH -- in front of its units,
    or, when the document begins with a space, as a word of its own before it -- wherever in the document that match is."""
    from hutoken_amd import _capi
    from oracle import oracle as O
    rng = random.Random(91)
    texts = [H.random_text(rng, max_words=rng.choice([3, 12, 60])) for _ in range(400)]
    texts += ["", " ", "   x", "...abc", " ...abc def", "\n\nhello", "a" * 300, " " + "b" * 300, "123 abc", " 123 abc"]
    for kind in ("char", "byte"):
        if kind == "char":
            ents, sp = H.random_char_vocab(5, n_merges=400)
            prefix, is_byte = "▁", False
        else:
            ents, sp = H.random_byte_vocab(15, n_merges=1200)
            prefix, is_byte = "Ġ", True
        vp, spath = H.write_vocab(tmp_path, "rp" + kind, ents, sp)
        ctx = _capi.Context(vp, spath, prefix, is_byte)
        data, offs = O.pack(texts)
        for pat in ["[a-z]+", "[ ]?[[:alpha:]]+|[ ]?[[:digit:]]+", ".+", "[^ ]+", "x"]:
            ctx.set_pattern(pat)
            orc = O.Oracle(vp, spath, prefix, is_byte, pattern=pat)
            ids, oo, st, rc = ctx.encode_packed(data, offs)
            want = orc.batch_encode(texts, 4)
            got = [ids[oo[i]:oo[i + 1]].tolist() for i in range(len(texts))]
            assert rc == 0 and got == want, (kind, pat, [i for i in range(len(texts)) if got[i] != want[i]][:5])
        ctx.close()


@pytest.mark.gpu
def test_python_surface_with_a_pattern(tmp_path, g9):
    import hutoken_amd as hutoken
    from oracle import oracle as O
    ents, sp = H.random_byte_vocab(11, n_merges=2000)
    vp, spath = H.write_vocab(tmp_path, "g9", ents, sp)
    hutoken.initialize(vp, spath, is_byte_encoder=True, pattern=G.GPT2_POSIX)
    texts = G.g9_texts(9000, 50)
    assert hutoken.batch_encode(texts, 2) == g9["mid"][0]["first"][:25] + hutoken.batch_encode(texts, 2)[25:]
    assert [hutoken.encode(t) for t in texts[:10]] == g9["mid"][0]["first"][:10]
    with pytest.raises(ValueError, match="Regex could not be compiled"):
        hutoken.initialize(vp, spath, is_byte_encoder=True, pattern="(")
    hutoken.initialize(vp, spath, is_byte_encoder=True)
    assert hutoken.encode("hello world") == O.Oracle(vp, spath, None, True).encode("hello world")
