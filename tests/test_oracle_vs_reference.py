"""The oracle against the reference itself on seeded random inputs.  Runs only where
oracle/_ref has been built (the build container; `make -C oracle ref`)."""
import random

import pytest

import helpers as H
from oracle import ref

pytestmark = pytest.mark.skipif(not ref.available(), reason="oracle/_ref not built")


def test_byte_mode_random_text_and_bytes(tmp_path, oracle_mod):
    for seed in range(4):
        ents, sp = H.random_byte_vocab(seed, n_merges=400, proper=seed % 2 == 0, dup_ids=seed == 3)
        vp, spath = H.write_vocab(tmp_path, f"b{seed}", ents, sp)
        orc = oracle_mod.Oracle(vp, spath, None, True)
        r = ref.RefTokenizer(vp, spath, None, True)
        rng = random.Random(seed * 77)
        for _ in range(600):
            t = H.random_text(rng)
            assert orc.encode(t) == r.encode(t), repr(t)
        for _ in range(600):  # not valid UTF-8: through the reference's internal C seam
            b = H.random_bytes_text(rng, rng.randint(0, 20))
            assert orc.encode_bytes(b)[0] == r.encode_bytes(b)[0], repr(b)


def test_char_mode_with_prefix(tmp_path, oracle_mod):
    for seed in range(3):
        ents, sp = H.random_char_vocab(seed, n_merges=400, drop_chars="qző漢" if seed % 2 else "")
        vp, spath = H.write_vocab(tmp_path, f"c{seed}", ents, sp)
        orc = oracle_mod.Oracle(vp, spath, "▁", False)
        r = ref.RefTokenizer(vp, spath, "▁", False)
        rng = random.Random(seed * 131)
        for _ in range(800):
            t = H.random_text(rng)
            assert orc.encode(t) == r.encode(t), repr(t)


def test_merges_path_byte_mode(tmp_path, oracle_mod):
    """The id-keyed merge path (merges file; core.c:211-337, 457-477; lib.c:573-663)."""
    for seed in range(5):
        ents, sp = H.random_byte_vocab(seed, n_merges=400, proper=seed % 2 == 0, dup_ids=seed == 3)
        vp, spath = H.write_vocab(tmp_path, f"mb{seed}", ents, sp)
        mp = H.write_merges(tmp_path, f"mb{seed}", H.random_merges_text(ents, seed, keep=0.6 if seed == 4 else 0.9))
        orc = oracle_mod.Oracle(vp, spath, None, True, mp)
        r = ref.RefTokenizer(vp, spath, None, True, mp)
        assert orc.has_merges
        rng = random.Random(seed * 91)
        for _ in range(600):
            t = H.random_text(rng)
            assert orc.encode(t) == r.encode(t), repr(t)
        for _ in range(400):
            b = H.random_bytes_text(rng, rng.randint(0, 20))
            assert orc.encode_bytes(b)[0] == r.encode_bytes(b)[0], repr(b)


def test_merges_path_char_mode_with_prefix(tmp_path, oracle_mod):
    """Non-byte mode: "<0xHH>" replacements are split per character on this path (core.c:460-474), the
    prefix encoded on its own still takes the string path (core.c:421-446)."""
    for seed in range(3):
        ents, sp = H.random_char_vocab(seed, n_merges=400, drop_chars="qző漢" if seed % 2 else "")
        vp, spath = H.write_vocab(tmp_path, f"mc{seed}", ents, sp)
        mp = H.write_merges(tmp_path, f"mc{seed}", H.random_merges_text(ents, seed + 10))
        orc = oracle_mod.Oracle(vp, spath, "▁", False, mp)
        r = ref.RefTokenizer(vp, spath, "▁", False, mp)
        rng = random.Random(seed * 57)
        for _ in range(800):
            t = H.random_text(rng)
            assert orc.encode(t) == r.encode(t), repr(t)


def test_merges_file_degenerate_forms(tmp_path, oracle_mod):
    """No countable line: the string path stays in force.  Countable lines but no valid rule: the id path
    with no merges at all (every character its own id)."""
    ents, sp = H.random_byte_vocab(9, n_merges=200)
    vp, spath = H.write_vocab(tmp_path, "md", ents, sp)
    texts = ["hello world", "árvíztűrő tükörfúrógép", " a  b", "", "x"]
    for name, body in [("empty", ""), ("hdr", "#version: 0.2\n"), ("nosp", "abc\ndef\n"),
                       ("junk", "zz yy\nqq ww\n"), ("half", "a \n")]:
        mp = H.write_merges(tmp_path, name, body)
        orc = oracle_mod.Oracle(vp, spath, None, True, mp)
        r = ref.RefTokenizer(vp, spath, None, True, mp)
        for t in texts:
            assert orc.encode(t) == r.encode(t), (name, t)
    with pytest.raises(FileNotFoundError):
        oracle_mod.Oracle(vp, spath, None, True, str(tmp_path / "missing.txt"))


def _same_decode(orc, r, ids):
    def run(f):
        try:
            return f(ids)
        except Exception as e:  # noqa: BLE001
            return ("raised", type(e).__name__)
    a, b = run(r.decode), run(orc.decode)
    assert a == b, (ids, a, b)


def test_decode_direction(tmp_path, oracle_mod):
    """decode (core.c:513-581, pretokenizer.c:197-296): byte mode and character mode with a prefix and
    byte-fallback literals; round trips, arbitrary id sequences (invalid UTF-8 raises in both), ids out of
    range.  Vocabularies with unique ids: with repeated ids the reference's table depends on its hash map."""
    for seed in range(3):
        ents, sp = H.random_byte_vocab(seed, n_merges=400, proper=seed != 1)
        vp, spath = H.write_vocab(tmp_path, f"d{seed}", ents, sp)
        orc = oracle_mod.Oracle(vp, spath, None, True)
        r = ref.RefTokenizer(vp, spath, None, True)
        rng = random.Random(seed * 31)
        for _ in range(400):
            t = H.random_text(rng)
            ids = r.encode(t)
            assert r.decode(ids) == t
            _same_decode(orc, r, ids)
        for _ in range(600):
            _same_decode(orc, r, [rng.randrange(0, len(ents)) for _ in range(rng.randint(0, 10))])
        for bad in ([-1], [len(ents)], [3, 10 ** 6]):
            _same_decode(orc, r, bad)
    for seed in range(2):
        ents, sp = H.random_char_vocab(seed, n_merges=400)
        vp, spath = H.write_vocab(tmp_path, f"dc{seed}", ents, sp)
        orc = oracle_mod.Oracle(vp, spath, "▁", False)
        r = ref.RefTokenizer(vp, spath, "▁", False)
        rng = random.Random(seed * 17)
        for _ in range(500):
            ids = [x for x in r.encode(H.random_text(rng)) if x >= 0]
            _same_decode(orc, r, ids)
        for _ in range(600):
            _same_decode(orc, r, [rng.randrange(0, len(ents)) for _ in range(rng.randint(0, 10))])


def test_batch_threads_and_word_too_large(tmp_path, oracle_mod):
    ents, sp = H.random_byte_vocab(9, n_merges=100)
    vp, spath = H.write_vocab(tmp_path, "t", ents, sp)
    orc = oracle_mod.Oracle(vp, spath, None, True)
    r = ref.RefTokenizer(vp, spath, None, True)
    texts = ["How can the net", " amount of entropy of", " the universe be massively decreased?"]
    for nt in (1, 3, 4, 8):
        assert orc.batch_encode(texts, nt) == r.batch_encode(texts, nt)
    assert r.batch_encode(texts, 0) == [[], [], []] == orc.batch_encode(texts, 0)
    big = "ab " + "x" * 262145 + " cd"
    # an over-long word silently ends the document: core.c:402-407 sets error_msg, core.c:503
    # clears it again, so neither encode (lib.c:692-697) nor batch_encode (lib.c:796-808) reports it
    assert orc.encode(big) == r.encode(big) == r.encode("ab")
    assert r.batch_encode([big, "ab"], 2) == orc.batch_encode([big, "ab"], 2)


def test_pattern_together_with_a_prefix(tmp_path, oracle_mod):
    """The regex pre-token path with a prefix (core.c:362-366, 420-451: the prefix goes with the document's first MATCH)
    and with replacements of several units, both shapes of vocabulary."""
    import locale
    if locale.setlocale(locale.LC_CTYPE, None) not in ("C.UTF-8", "C", "en_US.UTF-8"):
        pytest.skip("POSIX regex matching depends on LC_CTYPE")
    rng = random.Random(4242)
    texts = [H.random_text(rng, max_words=rng.choice([3, 12, 40])) for _ in range(500)]
    texts += ["", " ", "   x", "...abc", " ...abc def", "\n\nhello", " 123 abc"]
    for kind in ("char", "byte"):
        if kind == "char":
            ents, sp = H.random_char_vocab(5, n_merges=400)
            prefix, is_byte = "▁", False
        else:
            ents, sp = H.random_byte_vocab(15, n_merges=1200)
            sp = dict(sp)
            sp[ord("q")] = "qu"
            prefix, is_byte = "Ġ", True
        vp, spath = H.write_vocab(tmp_path, "rp" + kind, ents, sp)
        for pat in ["[a-z]+", "[ ]?[[:alpha:]]+|[ ]?[[:digit:]]+", ".+", "[^ ]+", "x"]:
            orc = oracle_mod.Oracle(vp, spath, prefix, is_byte, pattern=pat)
            r = ref.RefTokenizer(vp, spath, prefix, is_byte, pattern=pat)
            for t in texts:
                assert orc.encode(t) == r.encode(t), (kind, pat, repr(t))
