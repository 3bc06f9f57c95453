"""Decode direction on the GPU against the CPU oracle, which is pinned against the compiled
reference in tests/test_oracle_vs_reference.py::test_decode_direction.  Needs a real MI355X."""
import random

import numpy as np
import pytest

import helpers as H

pytestmark = pytest.mark.gpu


def _ctx(vp, sp, prefix, is_byte):
    from hutoken_amd import _capi
    return _capi.Context(vp, sp, prefix, is_byte)


def _compare(ctx, orc, id_lists, tag):
    offs = np.zeros(len(id_lists) + 1, dtype=np.int64)
    np.cumsum([len(x) for x in id_lists], out=offs[1:])
    flat = np.asarray([t for x in id_lists for t in x], dtype=np.int32)
    out, oo, st = ctx.decode_packed(flat, offs)
    assert (st == 0).all(), tag
    raw = out.tobytes()
    for d, ids in enumerate(id_lists):
        want, status = orc.decode_bytes(ids)
        assert status == 0, (tag, d, ids)
        got = raw[oo[d]:oo[d + 1]]
        assert got == want, f"{tag}: document {d} ids {ids[:12]}: {got[:40]!r} != {want[:40]!r}"


def test_round_trip_and_arbitrary_ids_byte_mode(tmp_path, oracle_mod):
    for seed in range(3):
        ents, sp = H.random_byte_vocab(seed, n_merges=500, proper=seed != 1)
        vp, spath = H.write_vocab(tmp_path, f"d{seed}", ents, sp)
        ctx, orc = _ctx(vp, spath, None, True), oracle_mod.Oracle(vp, spath, None, True)
        rng = random.Random(seed * 11)
        texts = [H.random_text(rng, max_words=40) for _ in range(2000)]
        lists = [orc.encode(t) for t in texts]
        _compare(ctx, orc, lists, f"roundtrip{seed}")
        out, oo, _ = ctx.decode_packed(np.asarray([t for x in lists for t in x], dtype=np.int32),
                                       np.concatenate([[0], np.cumsum([len(x) for x in lists])]).astype(np.int64))
        raw = out.tobytes()
        for d, t in enumerate(texts[:300]):
            assert raw[oo[d]:oo[d + 1]] == t.encode("utf-8")
        lists = [[rng.randrange(0, len(ents)) for _ in range(rng.randint(0, 40))] for _ in range(3000)]
        lists += [[], [], [5], [], list(range(256))]
        _compare(ctx, orc, lists, f"random{seed}")


def test_character_mode_with_prefix(tmp_path, oracle_mod):
    for seed in range(2):
        ents, sp = H.random_char_vocab(seed, n_merges=500)
        vp, spath = H.write_vocab(tmp_path, f"dc{seed}", ents, sp)
        ctx, orc = _ctx(vp, spath, "▁", False), oracle_mod.Oracle(vp, spath, "▁", False)
        rng = random.Random(seed * 13)
        lists = [[x for x in orc.encode(H.random_text(rng, max_words=30)) if x >= 0] for _ in range(2000)]
        lists += [[rng.randrange(0, len(ents)) for _ in range(rng.randint(0, 30))] for _ in range(2000)]
        _compare(ctx, orc, lists, f"char{seed}")


def test_vg_corpus_round_trip(vg_files, oracle_mod):
    from hutoken_amd import synth
    vp, sp, kw = vg_files
    ctx = _ctx(vp, sp, kw["prefix"], kw["is_byte_encoder"])
    d, o = synth.corpus("C3", 20000)
    ids, oo, st, rc = ctx.encode_packed(d, o)
    assert rc == 0
    out, boff, st = ctx.decode_packed(ids, oo)
    assert (st == 0).all()
    assert np.array_equal(boff, o) and np.array_equal(out, d)  # decode(encode(text)) == text, byte for byte


def test_look_back_that_helps_itself(vg_files, monkeypatch):
    """k_dec_tiles' look-back does not depend on the order in which workgroups start: a tile that has waited long enough
    for one in front of it adds that tile's bytes up itself.  HUTK_DEC_HELP_AFTER=0 makes every tile do so at its first
    unanswered poll -- the same text and offsets (character mode with a prefix too: the stripped first tokens)."""
    from hutoken_amd import synth
    monkeypatch.setenv("HUTK_DEC_HELP_AFTER", "0")
    vp, sp, kw = vg_files
    ctx = _ctx(vp, sp, kw["prefix"], kw["is_byte_encoder"])
    d, o = synth.corpus("C3", 60000)
    ids, oo, st, rc = ctx.encode_packed(d, o)
    assert rc == 0
    out, boff, st = ctx.decode_packed(ids, oo)
    assert (st == 0).all()
    assert np.array_equal(boff, o) and np.array_equal(out, d)
    from hutoken_amd import data
    vp, sp, kw = data.vocab_files("VL")
    ctx = _ctx(vp, sp, kw["prefix"], kw["is_byte_encoder"])
    d, o = synth.corpus("C5", 30000)
    ids, oo, st, rc = ctx.encode_packed(d, o)
    assert rc == 0 and (ids >= 0).all()
    out, boff, st = ctx.decode_packed(ids, oo)
    monkeypatch.delenv("HUTK_DEC_HELP_AFTER")
    out2, boff2, st2 = ctx.decode_packed(ids, oo)
    assert np.array_equal(boff, boff2) and np.array_equal(out, out2) and np.array_equal(st, st2)


def test_errors(tmp_path, oracle_mod):
    ents, sp = H.random_byte_vocab(7, n_merges=100)
    vp, spath = H.write_vocab(tmp_path, "e", ents, sp)
    ctx = _ctx(vp, spath, None, True)
    n = len(ents)
    with pytest.raises(ValueError, match="non-negative and less than vocab size"):
        ctx.decode_packed(np.array([1, n, 2], dtype=np.int32), np.array([0, 3], dtype=np.int64))
    with pytest.raises(ValueError, match="non-negative and less than vocab size"):
        ctx.decode_packed(np.array([-1], dtype=np.int32), np.array([0, 1], dtype=np.int64))
    # an id that two keys carry, an id that no key carries: undefined in the reference, refused here
    ents2, sp2 = H.random_byte_vocab(8, n_merges=100, dup_ids=True)
    vp2, spath2 = H.write_vocab(tmp_path, "e2", ents2, sp2)
    ctx2 = _ctx(vp2, spath2, None, True)
    from collections import Counter
    cnt = Counter(i for _k, i in ents2)
    dup = next(i for i, c in cnt.items() if c > 1)
    hole = next(i for i in range(len(ents2)) if i not in cnt)
    for bad in (dup, hole):
        with pytest.raises(ValueError, match="cannot be decoded on its own"):
            ctx2.decode_packed(np.array([bad], dtype=np.int32), np.array([0, 1], dtype=np.int64))


def test_python_surface(vg_files, oracle_mod):
    import hutoken_amd as hutoken
    vp, sp, kw = vg_files
    hutoken.initialize(vp, sp, **kw)
    s = "How can the net amount of entropy of the universe be massively decreased?"
    assert hutoken.decode(hutoken.encode(s)) == s
    texts = ["árvíztűrő tükörfúrógép", " 漢字 仮名", "", "a"]
    assert hutoken.batch_decode(hutoken.batch_encode(texts, 2), 2) == texts
    with pytest.raises(ValueError, match="hutoken: Error decoding tokens"):
        hutoken.decode([10 ** 7])
    with pytest.raises(RuntimeError, match="No tokens provided"):
        hutoken.batch_decode([])
    with pytest.raises(RuntimeError, match="Each item must be a list of integers"):
        hutoken.batch_decode([1, 2])


def test_g6_golden(tmp_path):
    """Against the committed outputs of the reference's own decode() (tests/golden/g6_decode.json)."""
    import hashlib
    import json
    import os
    import hutoken_amd as hutoken
    for g in json.load(open(os.path.join(H.GOLDEN_DIR, "g6_decode.json"))):
        if g["mode"] == "byte":
            ents, sp = H.random_byte_vocab(g["seed"], n_merges=1500, proper=g["proper"])
            prefix, is_byte = None, True
        else:
            ents, sp = H.random_char_vocab(g["seed"], n_merges=1500)
            prefix, is_byte = "▁", False
        vp, spath = H.write_vocab(tmp_path, "g6_%d" % g["seed"], ents, sp)
        hutoken.initialize(vp, spath, prefix=prefix, is_byte_encoder=is_byte)
        rng = random.Random(g["seed"] * 1000)
        h = hashlib.sha256()
        for k in range(g["n"]):
            if k % 2 == 0:
                ids = [x for x in hutoken.encode(H.random_text(rng, max_words=20)) if x >= 0]
            else:
                ids = [rng.randrange(0, len(ents)) for _ in range(rng.randint(0, 24))]
            try:
                res = hutoken._native_decode(ids)
            except Exception as e:  # noqa: BLE001
                res = {"raises": type(e).__name__}
            if k < len(g["first"]):
                assert [ids, res] == g["first"][k]
            h.update(json.dumps([ids, res], ensure_ascii=True).encode())
        assert h.hexdigest() == g["sha256"]
