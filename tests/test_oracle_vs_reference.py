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
