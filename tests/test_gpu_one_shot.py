"""The one-launch path of small calls (k_tiles<..., ONE>, hutk_api.cpp encode_one_shot): a batch of at most four tiles is
encoded by ONE kernel that reads the caller's page-locked buffer and raises a flag the host polls -- hutk_encode() of a
sentence in 33 us instead of 63.  Every shape such a batch can have, against the oracle: empty documents, documents cut
by tile borders, words for the exception kernels (the flag's second value: the tail runs as a launch of its own), the
Llama-shaped vocabulary with its prefix, the merges path.  GPU only."""
import random

import numpy as np
import pytest

import helpers as H

pytestmark = pytest.mark.gpu


def _batches(rng, n):
    out = []
    for _ in range(n):
        docs = []
        budget = rng.choice([40, 200, 900, 1800, 3700])
        while budget > 0 and len(docs) < 60:
            kind = rng.random()
            if kind < 0.1:
                t = ""
            elif kind < 0.2:
                t = rng.choice(["x" * rng.randint(33, 120), "ab" * rng.randint(17, 60), "漢字" * rng.randint(1, 40),
                                " " * rng.randint(1, 20) + "a", "é" * rng.randint(20, 70)])
            else:
                t = H.random_text(rng, rng.randint(1, 30))
            b = t.encode("utf-8")
            if len(b) > budget:
                b = b[:budget].decode("utf-8", "ignore").encode("utf-8")
            docs.append(b)
            budget -= max(len(b), 1)
        out.append(docs)
    return out


@pytest.mark.parametrize("vocab,merges", [("VG", False), ("VL", False), ("VG", True)])
def test_small_batches_in_one_launch(vocab, merges, oracle_mod):
    from hutoken_amd import _capi, data
    vp, sp, kw = data.vocab_files(vocab)
    mp = data.merges_file(vocab) if merges else None
    ctx = _capi.Context(vp, sp, kw["prefix"], kw["is_byte_encoder"], device=0, merges_path=mp)
    orc = oracle_mod.Oracle(vp, sp, kw["prefix"], kw["is_byte_encoder"], mp)
    rng = random.Random(99 + len(vocab) + int(merges))
    for docs in _batches(rng, 150):
        d = np.frombuffer(b"".join(docs), dtype=np.uint8).copy() if sum(map(len, docs)) else np.zeros(0, dtype=np.uint8)
        o = np.zeros(len(docs) + 1, dtype=np.int64)
        np.cumsum([len(x) for x in docs], out=o[1:])
        ids, oo, st, rc = ctx.encode_packed(d, o)
        ids_o, oo_o, st_o = orc.encode_packed(d, o, 1)
        assert rc == 0
        assert np.array_equal(oo, oo_o), docs
        assert np.array_equal(ids, ids_o), docs
    for s in ["", "a", "hello world", "How can the net amount of entropy of the universe be massively decreased?", "x" * 300]:
        assert ctx.encode_one(s.encode())[0] == orc.encode(s), s
    ctx.close()
