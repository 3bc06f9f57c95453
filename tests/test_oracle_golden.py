"""The CPU oracle against the committed golden vectors (outputs of the reference itself,
tools/make_golden.py).  No GPU, no /root/reference."""
import hashlib
import json
import os
import random

import numpy as np
import pytest

import helpers as H
from hutoken_amd import vocab_files as vf


def sha_ids(list_of_lists):
    h = hashlib.sha256()
    for ids in list_of_lists:
        h.update(np.asarray(ids, dtype="<i4").tobytes())
        h.update(b"|")
    return h.hexdigest()


def load(name):
    with open(os.path.join(H.GOLDEN_DIR, name)) as f:
        return json.load(f)


def g1_byte_vocab(tmp_path, merges_hex):
    t = vf.bytes_to_unicode()
    raw = [bytes([b]) for b in vf.byte_token_order()] + [bytes.fromhex(m) for m in merges_hex]
    entries = [(vf.encode_visible(tok, t), i) for i, tok in enumerate(raw)]
    return H.write_vocab(tmp_path, "g1", entries, vf.gpt2_special_mapping())


def test_g1_handpicked_byte_vocab(tmp_path, oracle_mod):
    g = load("g1_handpicked.json")["byte_vocab"]
    vp, sp = g1_byte_vocab(tmp_path, g["merges_hex"])
    orc = oracle_mod.Oracle(vp, sp, None, True)
    for c in g["cases"]:
        assert orc.encode(c["text"]) == c["ids"], repr(c["text"])
    texts = [c["text"] for c in g["cases"]]
    assert orc.batch_encode(texts, 3) == [c["ids"] for c in g["cases"]]


def test_g1_handpicked_char_vocab_with_prefix(tmp_path, oracle_mod):
    g = load("g1_handpicked.json")["char_vocab"]
    ents, sp = H.random_char_vocab(5, n_merges=300, drop_chars="qző漢")
    vp, spath = H.write_vocab(tmp_path, "g1c", ents, sp)
    orc = oracle_mod.Oracle(vp, spath, "▁", False)
    for c in g["cases"]:
        assert orc.encode(c["text"]) == c["ids"], repr(c["text"])
    assert any(-1 in c["ids"] for c in g["cases"])  # unknown characters are exercised


def test_g2_mid_vocabs(tmp_path, oracle_mod):
    for g in load("g2_mid_vocabs.json"):
        ents, sp = H.random_byte_vocab(g["seed"], n_merges=2000, proper=g["proper"], dup_ids=g["dup_ids"])
        vp, spath = H.write_vocab(tmp_path, "g2_%d" % g["seed"], ents, sp)
        orc = oracle_mod.Oracle(vp, spath, None, True)
        rng = random.Random(g["seed"] * 1000)
        texts = [H.random_text(rng, max_words=30) for _ in range(1500)]
        res = orc.batch_encode(texts, 4)
        assert res[:40] == g["first"]
        assert sum(len(x) for x in res) == g["n_ids"]
        assert sha_ids(res) == g["sha256"]


@pytest.mark.parametrize("fixture,vocab", [("g3_vg_corpora.json", "VG"), ("g4_vl_corpora.json", "VL")])
def test_corpora(fixture, vocab, oracle_mod):
    from hutoken_amd import data, synth
    vp, sp, kw = data.vocab_files(vocab)
    orc = oracle_mod.Oracle(vp, sp, kw["prefix"], kw["is_byte_encoder"])
    for g in load(fixture):
        if "text" in g:
            assert orc.encode(g["text"]) == g["ids"]
            continue
        d, o = synth.corpus(g["corpus"], g["n_docs"])
        assert hashlib.sha256(d.tobytes()).hexdigest() == g["corpus_sha256"], "corpus generator drifted"
        ids, oo, st = orc.encode_packed(d, o, 8)
        res = [ids[oo[i]:oo[i + 1]].tolist() for i in range(g["n_docs"])]
        assert res[:len(g["first"])] == g["first"]
        assert int(oo[-1]) == g["n_ids"]
        assert sha_ids(res) == g["sha256"]


def test_g5_merges_path(tmp_path, oracle_mod):
    """The id-keyed merge path (merges file) against the reference's outputs."""
    from hutoken_amd import data, synth
    for g in load("g5_merges_path.json"):
        if g.get("vocab") == "VG+merges":
            vp, sp, kw = data.vocab_files("VG")
            orc = oracle_mod.Oracle(vp, sp, kw["prefix"], kw["is_byte_encoder"], data.merges_file("VG"))
            d, o = synth.corpus(g["corpus"], g["n_docs"])
            assert hashlib.sha256(d.tobytes()).hexdigest() == g["corpus_sha256"]
            ids, oo, st = orc.encode_packed(d, o, 8)
            res = [ids[oo[i]:oo[i + 1]].tolist() for i in range(g["n_docs"])]
        else:
            ents, sp = H.random_byte_vocab(g["seed"], n_merges=2000, proper=g["proper"], dup_ids=g["dup_ids"])
            vp, spath = H.write_vocab(tmp_path, "g5_%d" % g["seed"], ents, sp)
            mp = H.write_merges(tmp_path, "g5_%d" % g["seed"], H.random_merges_text(ents, g["seed"] * 3, keep=0.8))
            orc = oracle_mod.Oracle(vp, spath, g["prefix"], True, mp)
            rng = random.Random(g["seed"] * 1000)
            res = orc.batch_encode([H.random_text(rng, max_words=30) for _ in range(1500)], 4)
        assert res[:len(g["first"])] == g["first"]
        assert sum(len(x) for x in res) == g["n_ids"]
        assert sha_ids(res) == g["sha256"]


def _g6_cases(g, tmp_path):
    import json as _json
    if g["mode"] == "byte":
        ents, sp = H.random_byte_vocab(g["seed"], n_merges=1500, proper=g["proper"])
        prefix, is_byte = None, True
    else:
        ents, sp = H.random_char_vocab(g["seed"], n_merges=1500)
        prefix, is_byte = "▁", False
    vp, spath = H.write_vocab(tmp_path, "g6_%d" % g["seed"], ents, sp)
    return vp, spath, prefix, is_byte, len(ents)


def test_g6_decode(tmp_path, oracle_mod):
    """Decode direction against the reference's outputs (texts, or the exception an invalid result raises)."""
    import json as _json
    for g in load("g6_decode.json"):
        vp, spath, prefix, is_byte, n = _g6_cases(g, tmp_path)
        orc = oracle_mod.Oracle(vp, spath, prefix, is_byte)
        rng = random.Random(g["seed"] * 1000)
        h = hashlib.sha256()
        for k in range(g["n"]):
            if k % 2 == 0:
                ids = [x for x in orc.encode(H.random_text(rng, max_words=20)) if x >= 0]
            else:
                ids = [rng.randrange(0, n) for _ in range(rng.randint(0, 24))]
            try:
                res = orc.decode(ids)
            except Exception as e:  # noqa: BLE001
                res = {"raises": type(e).__name__}
            if k < len(g["first"]):
                assert [ids, res] == g["first"][k]
            h.update(_json.dumps([ids, res], ensure_ascii=True).encode())
        assert h.hexdigest() == g["sha256"]


def test_g11_cjk_dense_vocabulary(oracle_mod):
    """A vocabulary with merges across neighbouring CJK characters (VC: no seam cuts a paragraph) and VG on the same text."""
    from hutoken_amd import data, synth
    for g in load("g11_cjk_dense.json"):
        vp, sp, kw = data.vocab_files(g["vocab"])
        orc = oracle_mod.Oracle(vp, sp, kw["prefix"], kw["is_byte_encoder"])
        d, o = getattr(synth, g["generator"])(g["n_docs"])
        assert hashlib.sha256(d.tobytes()).hexdigest() == g["corpus_sha256"], "text generator drifted"
        ids, oo, st = orc.encode_packed(d, o, 8)
        res = [ids[oo[i]:oo[i + 1]].tolist() for i in range(g["n_docs"])]
        assert res[:len(g["first"])] == g["first"]
        assert int(oo[-1]) == g["n_ids"]
        assert sha_ids(res) == g["sha256"]
