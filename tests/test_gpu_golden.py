"""GPU path against the committed golden vectors (the reference's own outputs) and, at the
full BASELINE sizes, size-independent properties.  Needs a real MI355X."""
import hashlib
import json
import os
import random

import numpy as np
import pytest

import helpers as H
from hutoken_amd import vocab_files as vf

pytestmark = pytest.mark.gpu


def sha_ids(list_of_lists):
    h = hashlib.sha256()
    for ids in list_of_lists:
        h.update(np.asarray(ids, dtype="<i4").tobytes())
        h.update(b"|")
    return h.hexdigest()


def load(name):
    with open(os.path.join(H.GOLDEN_DIR, name)) as f:
        return json.load(f)


def ctx_for(vp, sp, prefix, is_byte, merges=None):
    from hutoken_amd import _capi
    return _capi.Context(vp, sp, prefix, is_byte, merges_path=merges)


def encode_texts(ctx, texts):
    from oracle import oracle as O  # only its packer (str -> bytes, cut at NUL)
    data, offs = O.pack(texts)
    ids, oo, st, rc = ctx.encode_packed(data, offs)
    assert rc == 0
    return [ids[oo[i]:oo[i + 1]].tolist() for i in range(len(texts))]


def test_g1_handpicked(tmp_path):
    g = load("g1_handpicked.json")
    t = vf.bytes_to_unicode()
    raw = [bytes([b]) for b in vf.byte_token_order()] + [bytes.fromhex(m) for m in g["byte_vocab"]["merges_hex"]]
    vp, sp = H.write_vocab(tmp_path, "g1", [(vf.encode_visible(tok, t), i) for i, tok in enumerate(raw)],
                           vf.gpt2_special_mapping())
    ctx = ctx_for(vp, sp, None, True)
    cases = g["byte_vocab"]["cases"]
    assert encode_texts(ctx, [c["text"] for c in cases]) == [c["ids"] for c in cases]
    for c in cases[:20]:  # one document per call as well (hutk_encode)
        assert ctx.encode_one(c["text"].encode("utf-8"))[0] == c["ids"]
    ents, spm = H.random_char_vocab(5, n_merges=300, drop_chars="qző漢")
    vp, sp = H.write_vocab(tmp_path, "g1c", ents, spm)
    ctx = ctx_for(vp, sp, "▁", False)
    cases = g["char_vocab"]["cases"]
    assert encode_texts(ctx, [c["text"] for c in cases]) == [c["ids"] for c in cases]


def test_g2_mid_vocabs(tmp_path):
    for g in load("g2_mid_vocabs.json"):
        ents, sp = H.random_byte_vocab(g["seed"], n_merges=2000, proper=g["proper"], dup_ids=g["dup_ids"])
        vp, spath = H.write_vocab(tmp_path, "g2_%d" % g["seed"], ents, sp)
        ctx = ctx_for(vp, spath, None, True)
        rng = random.Random(g["seed"] * 1000)
        texts = [H.random_text(rng, max_words=30) for _ in range(1500)]
        res = encode_texts(ctx, texts)
        assert res[:40] == g["first"]
        assert sum(len(x) for x in res) == g["n_ids"]
        assert sha_ids(res) == g["sha256"]


@pytest.mark.parametrize("fixture,vocab", [("g3_vg_corpora.json", "VG"), ("g4_vl_corpora.json", "VL")])
def test_corpora(fixture, vocab):
    from hutoken_amd import data, synth
    vp, sp, kw = data.vocab_files(vocab)
    ctx = ctx_for(vp, sp, kw["prefix"], kw["is_byte_encoder"])
    for g in load(fixture):
        if "text" in g:
            assert ctx.encode_one(g["text"].encode())[0] == g["ids"]
            continue
        d, o = synth.corpus(g["corpus"], g["n_docs"])
        assert hashlib.sha256(d.tobytes()).hexdigest() == g["corpus_sha256"]
        ids, oo, st, rc = ctx.encode_packed(d, o)
        assert rc == 0
        res = [ids[oo[i]:oo[i + 1]].tolist() for i in range(g["n_docs"])]
        assert res[:len(g["first"])] == g["first"]
        assert int(oo[-1]) == g["n_ids"]
        assert sha_ids(res) == g["sha256"]


def test_full_size_c2_against_oracle(vg_files, oracle_mod):
    """BASELINE config 2: 100k docs, mean 256 B ASCII -- every id compared."""
    from hutoken_amd import synth
    vp, sp, kw = vg_files
    ctx = ctx_for(vp, sp, kw["prefix"], kw["is_byte_encoder"])
    orc = oracle_mod.Oracle(vp, sp, kw["prefix"], kw["is_byte_encoder"])
    d, o = synth.corpus("C2")
    ids_g, oo_g, st, rc = ctx.encode_packed(d, o)
    ids_o, oo_o, _ = orc.encode_packed(d, o, os.cpu_count() or 8)
    assert rc == 0 and np.array_equal(oo_g, oo_o) and np.array_equal(ids_g, ids_o)


def test_full_size_c3_properties(vg_files, oracle_mod):
    """BASELINE config 3: 1M docs, mean 512 B mixed UTF-8.  Checked through properties that do not
    need the oracle at full size: (1) a batch equals the concatenation of its shards (any cut);
    (2) reversing the document order permutes the per-document id lists; (3) a seeded sample of
    documents equals the oracle; (4) documents re-packed with empty documents interleaved."""
    from hutoken_amd import sharding, synth
    vp, sp, kw = vg_files
    ctx = ctx_for(vp, sp, kw["prefix"], kw["is_byte_encoder"])
    orc = oracle_mod.Oracle(vp, sp, kw["prefix"], kw["is_byte_encoder"])
    n = 1_000_000
    d, o = synth.corpus("C3", n)
    ids, oo, st, rc = ctx.encode_packed(d, o)
    assert rc == 0 and (st == 0).all()
    # (1) shards
    parts = sharding.shard_by_bytes(o, 3)
    cat_ids, bases = [], []
    for first, count in parts:
        ld, lo = sharding.local_view(d, o, first, count)
        i2, o2, _, rc2 = ctx.encode_packed(ld, lo)
        assert rc2 == 0
        cat_ids.append(i2)
        bases.append(int(o2[-1]))
    assert np.array_equal(np.concatenate(cat_ids), ids)
    assert sum(bases) == int(oo[-1])
    # (3) sample vs oracle
    rng = np.random.default_rng(7)
    pick = np.sort(rng.choice(n, size=3000, replace=False))
    sd = np.concatenate([d[o[i]:o[i + 1]] for i in pick])
    so = np.zeros(len(pick) + 1, dtype=np.int64)
    so[1:] = np.cumsum([o[i + 1] - o[i] for i in pick])
    ids_o, oo_o, _ = orc.encode_packed(sd, so, 8)
    for k, i in enumerate(pick):
        assert np.array_equal(ids[oo[i]:oo[i + 1]], ids_o[oo_o[k]:oo_o[k + 1]]), int(i)
    # (2) + (4) on a 50k-document slice: reversed order with empty documents interleaved
    m = 50_000
    docs = [d[o[i]:o[i + 1]] for i in range(m)]
    rev = []
    for x in reversed(docs):
        rev.append(x)
        rev.append(d[0:0])
    rd = np.concatenate(rev)
    ro = np.zeros(len(rev) + 1, dtype=np.int64)
    ro[1:] = np.cumsum([len(x) for x in rev])
    ids_r, oo_r, _, rc3 = ctx.encode_packed(rd, ro)
    assert rc3 == 0
    for i in range(0, m, 97):
        j = 2 * (m - 1 - i)
        assert np.array_equal(ids_r[oo_r[j]:oo_r[j + 1]], ids[oo[i]:oo[i + 1]])
        assert oo_r[j + 1] == oo_r[j + 2]  # the interleaved empty document


def test_errors_and_limits(tmp_path, oracle_mod):
    ents, sp = H.random_byte_vocab(4, n_merges=200)
    vp, spath = H.write_vocab(tmp_path, "e", ents, sp)
    ctx = ctx_for(vp, spath, None, True)
    orc = oracle_mod.Oracle(vp, spath, None, True)
    # a 0x00 byte inside a document is an error of the packed interface
    with pytest.raises(ValueError, match="0x00"):
        ctx.encode_packed(np.frombuffer(b"ab\0cd", dtype=np.uint8), np.array([0, 5], dtype=np.int64))
    # the reference's word limit: 262144 bytes pass, 262145 silently end the document (core.c:402-407, 503)
    ok = b"hi " + b"x" * 262143 + b" yo"      # " x..." is 262144 bytes with its leading space
    cut = b"hi " + b"x" * 262144 + b" yo"
    docs = [b"before", ok, cut, b"after"]
    from oracle import oracle as O
    data, offs = O.pack(docs)
    ids_o, oo_o, st_o = orc.encode_packed(data, offs, 4)
    ids_g, oo_g, st_g, rc = ctx.encode_packed(data, offs)
    assert rc == 0
    assert st_o.tolist() == [0, 0, 1, 0] == st_g.tolist()
    assert np.array_equal(oo_o, oo_g) and np.array_equal(ids_o, ids_g)
    assert ids_g[oo_g[2]:oo_g[3]].tolist() == orc.encode(b"hi")


def test_g10_replacements_of_several_units(tmp_path):
    """Special-character replacements of several units (the reference's pretokenizer emits any string, src/pretokenizer.c:
    102-168; its own tests pin 'a' -> "Alpha", tests/test_pretokenizer.c:38-41): the words that hold such an item go
    through the exception kernels, which expand it.  Vectors: the compiled reference (tools/make_golden_g10.py)."""
    import test_g10_cpu as G
    with open(os.path.join(H.GOLDEN_DIR, "g10_pretokenizer.json")) as f:
        g10 = json.load(f)
    for case in g10["pretokenizer"]:
        if "ids" not in case:
            continue
        vp, sp = G.case_files(tmp_path, case)
        ctx = ctx_for(vp, sp, case["prefix"] or None, case["is_byte_encoder"])
        assert ctx.encode_one(case["text"].encode("utf-8"))[0] == case["ids"], case["name"]
        assert ctx.encode_one(("x " + case["text"] + " y " + case["text"]).encode("utf-8"))[0] == case["ids_in_sentence"], case["name"]
        ctx.close()
    for g in g10["files"]:
        vp, sp, mp, texts = G.file_case(tmp_path, g)
        ctx = ctx_for(vp, sp, g["prefix"], g["kind"] == "byte", mp)
        res = encode_texts(ctx, texts)
        assert res[:len(g["first"])] == g["first"], g["seed"]
        assert sum(len(x) for x in res) == g["n_ids"] and G.sha_ids(res) == g["sha256"], g["seed"]
        ctx.close()


def test_g5_merges_path(tmp_path):
    """The id-keyed merge path (merges file) against the reference's outputs."""
    from hutoken_amd import data, synth
    for g in load("g5_merges_path.json"):
        if g.get("vocab") == "VG+merges":
            vp, sp, kw = data.vocab_files("VG")
            ctx = ctx_for(vp, sp, kw["prefix"], kw["is_byte_encoder"], data.merges_file("VG"))
            d, o = synth.corpus(g["corpus"], g["n_docs"])
            ids, oo, st, rc = ctx.encode_packed(d, o)
            assert rc == 0
            res = [ids[oo[i]:oo[i + 1]].tolist() for i in range(g["n_docs"])]
        else:
            ents, sp = H.random_byte_vocab(g["seed"], n_merges=2000, proper=g["proper"], dup_ids=g["dup_ids"])
            vp, spath = H.write_vocab(tmp_path, "g5_%d" % g["seed"], ents, sp)
            mp = H.write_merges(tmp_path, "g5_%d" % g["seed"], H.random_merges_text(ents, g["seed"] * 3, keep=0.8))
            ctx = ctx_for(vp, spath, g["prefix"], True, mp)
            assert ctx.uses_merges
            rng = random.Random(g["seed"] * 1000)
            res = encode_texts(ctx, [H.random_text(rng, max_words=30) for _ in range(1500)])
        assert res[:len(g["first"])] == g["first"]
        assert sum(len(x) for x in res) == g["n_ids"]
        assert sha_ids(res) == g["sha256"]


def test_chunked_host_path_equals_the_simple_one(vg_files, monkeypatch):
    """hutk_encode_batch on a large batch overlaps copies and kernels chunk by chunk (pageable and page-locked
    buffers); same ids, offsets and status as the unchunked path, including a document cut at an over-long
    word, which sends the whole batch through the simple path's trimming."""
    from hutoken_amd import _capi, synth
    vp, sp, kw = vg_files
    ctx = ctx_for(vp, sp, kw["prefix"], kw["is_byte_encoder"])
    d, o = synth.corpus("C3", 150000)  # 76 MB: above the 48 MB threshold, five chunks
    monkeypatch.setenv("HUTK_NO_PIPELINE", "1")
    ids_s, oo_s, st_s, rc = ctx.encode_packed(d, o)
    assert rc == 0
    monkeypatch.delenv("HUTK_NO_PIPELINE")
    ids_c, oo_c, st_c, rc = ctx.encode_packed(d, o)
    assert rc == 0
    assert np.array_equal(oo_s, oo_c) and np.array_equal(ids_s, ids_c) and np.array_equal(st_s, st_c)
    # page-locked buffers through the raw C ABI
    n = len(o) - 1
    cap = ctx.ids_capacity(len(d), n)
    pb, po = _capi.PinnedArray(len(d), np.uint8), _capi.PinnedArray(n + 1, np.int64)
    pi, poo, pst = _capi.PinnedArray(cap, np.int32), _capi.PinnedArray(n + 1, np.int64), _capi.PinnedArray(n, np.int32)
    pb.array[:] = d
    po.array[:] = o
    rc = _capi.load().hutk_encode_batch(ctx.handle, pb.array.ctypes.data, po.array.ctypes.data, n, pi.array.ctypes.data,
                                        cap, poo.array.ctypes.data, pst.array.ctypes.data)
    assert rc == 0
    assert np.array_equal(poo.array, oo_s) and np.array_equal(pi.array[: int(oo_s[-1])], ids_s)
    # many small chunks: the three sets of chunk buffers go round six times
    monkeypatch.setenv("HUTK_PIPE_CHUNK_MB", "4")
    pi.array[:] = -7
    poo.array[:] = -7
    rc = _capi.load().hutk_encode_batch(ctx.handle, pb.array.ctypes.data, po.array.ctypes.data, n, pi.array.ctypes.data,
                                        cap, poo.array.ctypes.data, pst.array.ctypes.data)
    monkeypatch.delenv("HUTK_PIPE_CHUNK_MB")
    assert rc == 0
    assert np.array_equal(poo.array, oo_s) and np.array_equal(pi.array[: int(oo_s[-1])], ids_s) and np.array_equal(pst.array, st_s)
    # an over-long word in the middle of a chunked batch
    docs_mid = int(n // 2)
    cut_at = int(o[docs_mid])
    big = np.frombuffer(b"x" * 262200 + b" tail", dtype=np.uint8)
    d2 = np.concatenate([d[:cut_at], big, d[cut_at:]])
    o2 = np.concatenate([o[: docs_mid + 1], o[docs_mid:] + len(big)]).astype(np.int64)
    o2[docs_mid + 1] = cut_at + len(big)
    ids_c2, oo_c2, st_c2, rc = ctx.encode_packed(d2, o2)
    monkeypatch.setenv("HUTK_NO_PIPELINE", "1")
    ids_s2, oo_s2, st_s2, rc2 = ctx.encode_packed(d2, o2)
    assert rc == 0 and rc2 == 0
    assert st_c2[docs_mid] == 1 and oo_c2[docs_mid + 1] == oo_c2[docs_mid]
    assert np.array_equal(oo_s2, oo_c2) and np.array_equal(ids_s2, ids_c2) and np.array_equal(st_s2, st_c2)


def test_g8_reference_fixtures(tmp_path):
    """Loader quirk files, the reference parser's own test strings and texts with repeated pairs (tools/make_golden_g8.py:
    outcomes of the compiled reference) through the product's loader and the GPU path."""
    import sys
    sys.path.insert(0, os.path.join(H.ROOT, "tools"))
    import make_golden_g8 as G8  # only its seeded file builders
    from test_reference_fixtures_cpu import REFUSED_BY_DESIGN
    g = load("g8_reference_fixtures.json")
    files = G8.quirk_files()
    for case in g["loaders"]:
        vocab, special, probes = files[case["name"]]
        vp, sp = G8.write_case_files(str(tmp_path), case["name"], vocab, special)
        if case["name"] in REFUSED_BY_DESIGN:
            with pytest.raises(REFUSED_BY_DESIGN[case["name"]]):
                ctx_for(vp, sp, None, True)
            continue
        if "error" in case:
            with pytest.raises(Exception) as ei:
                ctx_for(vp, sp, None, True)
            assert [type(ei.value).__name__, str(ei.value)] == case["error"], case["name"]
            continue
        ctx = ctx_for(vp, sp, None, True)
        assert encode_texts(ctx, probes) == case["ids"], case["name"]
        ctx.close()
    t = vf.bytes_to_unicode()
    g1 = load("g1_handpicked.json")
    raw = [bytes([b]) for b in vf.byte_token_order()] + [bytes.fromhex(m) for m in g1["byte_vocab"]["merges_hex"]]
    vp, sp = H.write_vocab(tmp_path, "g8t", [(vf.encode_visible(tok, t), i) for i, tok in enumerate(raw)],
                           vf.gpt2_special_mapping())
    ctx = ctx_for(vp, sp, None, True)
    cases = g["ties"]  # the parser's 28 strings are among them: wrong word boundaries would show as wrong ids
    assert encode_texts(ctx, [c["text"] for c in cases]) == [c["ids"] for c in cases]
    for c in cases:
        assert ctx.encode_one(c["text"].encode("utf-8"))[0] == c["ids"]


def test_g11_cjk_dense_vocabulary():
    """VC: merges across neighbouring CJK characters, so the seam map cuts nothing and every paragraph is one word of
    several hundred bytes (the exception kernels); VG on the same text (seams cut).  Expected ids: the compiled reference's."""
    from hutoken_amd import data, synth
    for g in load("g11_cjk_dense.json"):
        vp, sp, kw = data.vocab_files(g["vocab"])
        ctx = ctx_for(vp, sp, kw["prefix"], kw["is_byte_encoder"])
        d, o = getattr(synth, g["generator"])(g["n_docs"])
        assert hashlib.sha256(d.tobytes()).hexdigest() == g["corpus_sha256"]
        ids, oo, st, rc = ctx.encode_packed(d, o)
        assert rc == 0
        res = [ids[oo[i]:oo[i + 1]].tolist() for i in range(g["n_docs"])]
        assert res[:len(g["first"])] == g["first"]
        assert int(oo[-1]) == g["n_ids"]
        assert sha_ids(res) == g["sha256"]


def test_second_level_of_the_seam_map(oracle_mod):
    """VC on batches large enough for k_ptiles, which asks the seam map's second level (whole characters, hutk_seam2.h) where
    the byte map says "may join": the vocabulary's own distribution (a cut at one boundary in five) and characters drawn at
    random (cut nearly everywhere).  Every id against the oracle -- with the level, without it (HUTK_NO_SEAM2=1), and
    through k_tiles, which does not ask it (HUTK_PTILES=0)."""
    import os
    from hutoken_amd import data, synth
    vp, sp, kw = data.vocab_files("VC")
    orc = oracle_mod.Oracle(vp, sp, kw["prefix"], kw["is_byte_encoder"])
    for gen in (lambda: synth.cjk_text(1800), lambda: synth.cjk_paragraphs(1500)):
        d, o = gen()
        assert (len(d) + 959) // 960 >= 2048
        ids_o, oo_o, _ = orc.encode_packed(d, o, 8)
        for env in ({}, {"HUTK_NO_SEAM2": "1"}, {"HUTK_PTILES": "0"}):
            old = {k: os.environ.pop(k, None) for k in ("HUTK_NO_SEAM2", "HUTK_PTILES")}
            os.environ.update(env)
            try:
                ctx = ctx_for(vp, sp, kw["prefix"], kw["is_byte_encoder"])
                ids, oo, st, rc = ctx.encode_packed(d, o)
            finally:
                for k in ("HUTK_NO_SEAM2", "HUTK_PTILES"):
                    os.environ.pop(k, None)
                    if old[k] is not None:
                        os.environ[k] = old[k]
            assert rc == 0
            assert np.array_equal(oo, oo_o) and np.array_equal(ids, ids_o), env


def test_batches_the_persistent_tile_kernel_is_chosen_for(oracle_mod):
    """A batch of a few thousand tiles dense in three-byte characters: the default mode enqueues both tile kernels and the
    sample k_pre takes of the batch makes k_ptiles the one that runs (hutk_api.cpp, Workspace::select); mixed text of the same
    size goes to k_tiles.  Every id of both against the oracle, and both again with each kernel forced."""
    import os
    from hutoken_amd import data, synth
    vp, sp, kw = data.vocab_files("VG")
    orc = oracle_mod.Oracle(vp, sp, kw["prefix"], kw["is_byte_encoder"])
    for gen in (lambda: synth.cjk_paragraphs(3000), lambda: synth.corpus("C3", 12000)):
        d, o = gen()
        assert (len(d) + 959) // 960 >= 2048
        ids_o, oo_o, _ = orc.encode_packed(d, o, 8)
        for mode in (None, "0", "1"):
            old = os.environ.pop("HUTK_PTILES", None)
            if mode is not None:
                os.environ["HUTK_PTILES"] = mode
            try:
                ctx = ctx_for(vp, sp, kw["prefix"], kw["is_byte_encoder"])
                ids, oo, st, rc = ctx.encode_packed(d, o)
            finally:
                os.environ.pop("HUTK_PTILES", None)
                if old is not None:
                    os.environ["HUTK_PTILES"] = old
            assert rc == 0
            assert np.array_equal(oo, oo_o) and np.array_equal(ids, ids_o), mode
