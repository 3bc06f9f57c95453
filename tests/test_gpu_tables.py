"""GPU parity at the edges of the device tables: vocabularies beyond 65519 symbols (32-bit symbols in LDS and in the tile
runs, 12-byte keys in the whole-word table), and words whose length sits at the whole-word table's key limits (14 bytes
with 16-bit symbols, 12 with 32-bit ones, 28 for the companion table).  Bit-exact against the
oracle.  Needs a real MI355X."""
import random

import numpy as np
import pytest

import helpers as H

pytestmark = pytest.mark.gpu


def _ctx(vp, sp, prefix, is_byte):
    from hutoken_amd import _capi
    return _capi.Context(vp, sp, prefix, is_byte)


def _same(ctx, orc, docs, tag):
    from oracle import oracle as O
    data, offs = O.pack(docs)
    ids_o, oo_o, st_o = orc.encode_packed(data, offs, num_threads=8)
    ids_g, oo_g, st_g, rc = ctx.encode_packed(data, offs)
    assert rc == 0, tag
    assert np.array_equal(oo_o, oo_g), tag
    if not np.array_equal(ids_o, ids_g):
        k = int(np.nonzero(ids_o != ids_g)[0][0])
        d = int(np.searchsorted(oo_o, k, side="right") - 1)
        raise AssertionError(f"{tag}: ids differ in doc {d}={docs[d]!r}\n oracle={ids_o[oo_o[d]:oo_o[d + 1]].tolist()}\n"
                             f" gpu   ={ids_g[oo_g[d]:oo_g[d + 1]].tolist()}")
    assert (st_g == 0).all(), tag


def _raw_tokens(entries, is_byte):
    """The vocabulary keys as raw input bytes (byte mode: the visible form undone; other mode: the prefix as a space)."""
    from hutoken_amd import vocab_files as vf
    if not is_byte:
        return [k.decode("utf-8").replace("▁", " ").encode("utf-8") for k, _ in entries if not k.startswith(b"<0x")]
    back = {c: b for b, c in vf.bytes_to_unicode().items()}
    out = []
    for k, _ in entries:
        try:
            out.append(bytes(back[c] for c in k.decode("utf-8")))
        except (KeyError, UnicodeDecodeError):
            pass
    return out


def _docs_of_tokens(rng, toks, n_docs, by_len=None):
    """Documents of vocabulary tokens as words of their own (the whole-word table's hits), of tokens glued together (its
    misses: the merge loop) and of single letters."""
    docs = []
    for _ in range(n_docs):
        parts = []
        for _ in range(rng.randint(1, 14)):
            r = rng.random()
            if by_len and r < 0.5:
                parts.append(rng.choice(by_len[rng.choice(list(by_len))]))
            elif r < 0.8:
                parts.append(rng.choice(toks))
            elif r < 0.9:
                parts.append(rng.choice(toks) + rng.choice(toks))
            else:
                parts.append(bytes([rng.choice(b"etaoinshrdlu")]))
        docs.append(b" ".join(p.strip(b" \n\t") or b"x" for p in parts).replace(b"\0", b""))
    return docs


def test_vocabulary_beyond_16_bit_symbols(tmp_path, oracle_mod):
    """70 000 merges: symbols no longer fit 16 bits, so the 32-bit instantiations of every kernel run, the whole-word
    table holds 12-byte keys with the symbol in the fourth dword, and the pair table's 20-bit fields are all in use."""
    ents, sp = H.random_byte_vocab(5, n_merges=70000, max_len=16)
    vp, spath = H.write_vocab(tmp_path, "big", ents, sp)
    ctx = _ctx(vp, spath, None, True)
    st = ctx.table_stats()
    assert st["n_sym"] >= 65520 and st["n_word_entries"] > 200
    orc = oracle_mod.Oracle(vp, spath, None, True)
    rng = random.Random(11)
    toks = [t for t in _raw_tokens(ents, True) if b"\0" not in t]
    by_len = {n: [t for t in toks if len(t.strip(b" \n\t")) == n] for n in (11, 12, 13, 14)}
    by_len = {n: v for n, v in by_len.items() if v}
    assert 12 in by_len and 13 in by_len
    _same(ctx, orc, _docs_of_tokens(rng, toks, 4000, by_len), "32-bit symbols, token words")
    _same(ctx, orc, [H.random_text(rng, max_words=40).encode("utf-8") for _ in range(3000)], "32-bit symbols, random text")
    long_words = [bytes(rng.choice(b"etaoinshrdlu") for _ in range(rng.randint(20, 300))) for _ in range(300)]
    _same(ctx, orc, [b" ".join(rng.sample(long_words, 3)) for _ in range(200)], "32-bit symbols, exception words")
    ctx.close()


@pytest.mark.parametrize("is_byte", [True, False], ids=["byte-mode", "char-mode"])
def test_word_lengths_at_the_table_key_limits(tmp_path, oracle_mod, is_byte):
    """Tokens of 11..16 bytes as words of their own: 14 bytes is the last length the 16-byte slot holds (16-bit symbols);
    15 and 16 go to the companion table outside byte-encoder mode and through the merge loop in it."""
    if is_byte:
        ents, sp = H.random_byte_vocab(21, n_merges=6000, max_len=16)
        prefix = None
    else:
        ents, sp = H.random_char_vocab(21, n_merges=6000, max_len=16)
        prefix = "▁"
    vp, spath = H.write_vocab(tmp_path, "edge", ents, sp)
    ctx = _ctx(vp, spath, prefix, is_byte)
    assert ctx.table_stats()["n_word_entries"] > 50
    orc = oracle_mod.Oracle(vp, spath, prefix, is_byte)
    rng = random.Random(12)
    toks = [t for t in _raw_tokens(ents, is_byte) if b"\0" not in t and t.strip(b" \n\t")]
    by_len = {n: [t for t in toks if len(b" " + t.strip(b" \n\t")) == n] for n in range(11, 18)}
    by_len = {n: v for n, v in by_len.items() if v}
    assert {13, 14, 15, 16} <= set(by_len), sorted(by_len)
    _same(ctx, orc, _docs_of_tokens(rng, toks, 6000, by_len), "key limits")
    ctx.close()


def _with_chains(ents, is_byte, rng, n_chains=400, lo=10, hi=31):
    """The vocabulary plus chains of tokens that make long single-token WORDS: for a random word of capital consonants
    (letters no other key holds) every prefix " B", " BC", " BCD", ... is a key, so the merge loop can only walk the chain
    and the word ends as one token.  -> (entries, the words as raw bytes with their leading space)"""
    from hutoken_amd import vocab_files as vf
    t = vf.bytes_to_unicode()
    have = {k for k, _ in ents}
    nid = max(i for _, i in ents) + 1
    out, words = list(ents), []
    for _ in range(n_chains):
        w = " " + "".join(rng.choice("BCDFGHJKLMPQRSVWXZ") for _ in range(rng.randint(lo, hi) - 1))
        for k in range(2, len(w) + 1):
            key = vf.encode_visible(w[:k].encode(), t) if is_byte else w[:k].replace(" ", "▁").encode("utf-8")
            if key not in have:
                have.add(key)
                out.append((key, nid))
                nid += 1
        words.append(w.encode())
    return out, words


@pytest.mark.parametrize("is_byte", [True, False], ids=["byte-mode", "char-mode"])
def test_long_words_of_the_companion_table(tmp_path, oracle_mod, is_byte, monkeypatch):
    """Single-token words of 10..31 bytes: 15..28 bytes are the companion table's (two 16-byte slots behind the main
    table, loaded by the word's lane instead of the main table's two candidate slots), 29 and more go through the merge
    loop, and so does a table word with one letter changed.  With and without the companion the ids are the oracle's."""
    rng = random.Random(13)
    if is_byte:
        ents, sp = H.random_byte_vocab(31, n_merges=3000, max_len=16)
        prefix = None
    else:
        ents, sp = H.random_char_vocab(31, n_merges=3000, max_len=12)
        prefix = "▁"
    ents, words = _with_chains(ents, is_byte, rng)
    assert {15, 16, 17, 27, 28, 29, 31} <= {len(w) for w in words}
    vp, spath = H.write_vocab(tmp_path, "long", ents, sp)
    orc = oracle_mod.Oracle(vp, spath, prefix, is_byte)
    toks = [t for t in _raw_tokens(ents, is_byte) if b"\0" not in t and t.strip(b" \n\t")]
    docs = []
    for _ in range(6000):
        parts = []
        for _ in range(rng.randint(1, 12)):
            r = rng.random()
            w = rng.choice(words).strip()
            if r < 0.5:
                parts.append(w)
            elif r < 0.65:  # one letter off: same hash input length, no entry
                k = rng.randrange(len(w))
                parts.append(w[:k] + bytes([rng.choice(b"BCDFGHJKLMPQRSVWXZ")]) + w[k + 1:])
            elif r < 0.75:
                parts.append(w[:rng.randint(1, len(w))])
            else:
                parts.append(rng.choice(toks).strip(b" \n\t") or b"x")
        docs.append((b" " if rng.random() < 0.5 else b"") + b" ".join(parts))
    ctx = _ctx(vp, spath, prefix, is_byte)
    n_long = ctx.table_stats()["n_long_word_entries"]
    assert n_long > 1000, n_long  # (400 chains x the prefixes of 15..28 bytes, minus the one-choice table's collisions)
    _same(ctx, orc, docs, "companion table")
    ctx.close()
    monkeypatch.setenv("HUTK_NO_LONG_WORD_TABLE", "1")
    ctx = _ctx(vp, spath, prefix, is_byte)
    assert ctx.table_stats()["n_long_word_entries"] == 0
    _same(ctx, orc, docs, "without the companion table")
    ctx.close()
