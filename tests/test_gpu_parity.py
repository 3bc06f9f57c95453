"""GPU parity: the HIP path, called through the C ABI, against the CPU oracle on the
same seeded inputs.  Bit-exact ids and offsets are required (integer work, no
tolerance).  Every test here needs a real MI355X."""
import random

import numpy as np
import pytest

import helpers as H

pytestmark = pytest.mark.gpu


def _ctx(vp, sp, prefix, is_byte, merges=None):
    from hutoken_amd import _capi
    return _capi.Context(vp, sp, prefix, is_byte, merges_path=merges)


def _compare(ctx, orc, docs, tag=""):
    """docs: list[bytes] without 0x00."""
    from oracle import oracle as O
    data, offs = O.pack(docs)
    ids_o, oo_o, st_o = orc.encode_packed(data, offs, num_threads=4)
    ids_g, oo_g, st_g, rc = ctx.encode_packed(data, offs)
    assert rc == 0, f"{tag}: rc={rc}"
    if not np.array_equal(oo_o, oo_g):
        bad = int(np.nonzero(oo_o != oo_g)[0][0])
        d = max(bad - 1, 0)
        raise AssertionError(
            f"{tag}: out_offsets differ first at {bad}; doc {d}={docs[d]!r}\n"
            f" oracle={ids_o[oo_o[d]:oo_o[d + 1]].tolist()}\n gpu   ={ids_g[oo_g[d]:oo_g[d + 1]].tolist()}")
    if not np.array_equal(ids_o, ids_g):
        k = int(np.nonzero(ids_o != ids_g)[0][0])
        d = int(np.searchsorted(oo_o, k, side="right") - 1)
        raise AssertionError(
            f"{tag}: ids differ first at {k} (doc {d}={docs[d]!r})\n"
            f" oracle={ids_o[oo_o[d]:oo_o[d + 1]].tolist()}\n gpu   ={ids_g[oo_g[d]:oo_g[d + 1]].tolist()}")
    assert (st_g == 0).all()


@pytest.fixture(scope="module")
def small_byte(tmp_path_factory, oracle_mod):
    tmp = tmp_path_factory.mktemp("bv")
    out = []
    for seed, proper, dup in [(1, True, False), (2, False, False), (3, True, True)]:
        ents, sp = H.random_byte_vocab(seed, n_merges=500, proper=proper, dup_ids=dup)
        vp, spath = H.write_vocab(tmp, f"b{seed}", ents, sp)
        out.append((_ctx(vp, spath, None, True), oracle_mod.Oracle(vp, spath, None, True)))
    return out


@pytest.fixture(scope="module")
def small_char(tmp_path_factory, oracle_mod):
    tmp = tmp_path_factory.mktemp("cv")
    out = []
    for seed, drop in [(1, ""), (2, "qző漢")]:
        ents, sp = H.random_char_vocab(seed, n_merges=500, drop_chars=drop)
        vp, spath = H.write_vocab(tmp, f"c{seed}", ents, sp)
        out.append((_ctx(vp, spath, "▁", False), oracle_mod.Oracle(vp, spath, "▁", False)))
    return out


def test_hello_world(small_byte):
    for ctx, orc in small_byte:
        _compare(ctx, orc, [b"hello world"], "hello")


def test_edge_documents(small_byte):
    docs = [b"", b" ", b"  ", b"a", b" a", b"  a", b"a  b", b"a   b   c", b"aaaaa", b"\t\n\r\f\v", b"word ",
            b"", b"", "árvíztűrő tükörfúrógép".encode(), " First Second".encode(), "€".encode(),
            "😂".encode(), b"word123", b"123.!", b".!word", b"A  B   C", b"Hello world 123. End!", b"x \ta",
            "a b".encode(), "ab".encode(), b"", b"z"]
    for ctx, orc in small_byte:
        _compare(ctx, orc, docs, "edge")
        _compare(ctx, orc, [b""], "one-empty")
        _compare(ctx, orc, [b"", b"", b""], "all-empty")


def test_random_text_byte_mode(small_byte):
    for k, (ctx, orc) in enumerate(small_byte):
        rng = random.Random(100 + k)
        docs = [H.random_text(rng, max_words=40).encode("utf-8") for _ in range(3000)]
        _compare(ctx, orc, docs, f"text{k}")


def test_arbitrary_bytes_byte_mode(small_byte):
    """Truncated / invalid / overlong UTF-8: defined for the byte encoder."""
    for k, (ctx, orc) in enumerate(small_byte):
        rng = random.Random(200 + k)
        docs = [H.random_bytes_text(rng, rng.randint(0, 60)) for _ in range(3000)]
        _compare(ctx, orc, docs, f"bytes{k}")


def test_document_boundaries_inside_characters(small_byte):
    """Ragged packing: documents end in the middle of multi-byte sequences and
    tiles end in the middle of words."""
    ctx, orc = small_byte[0]
    rng = random.Random(7)
    blob = "".join(H.random_text(rng, max_words=30) for _ in range(400)).encode("utf-8").replace(b"\0", b"")
    docs, i = [], 0
    while i < len(blob):
        n = rng.choice([0, 1, 2, 3, 5, 17, 64, 300, 2047, 2048, 2049, 5000])
        docs.append(blob[i:i + n])
        i += n
    _compare(ctx, orc, docs, "ragged")


def test_long_words_exception_path(small_byte):
    """Words beyond one lane's capacity (48 units), beyond the staged window,
    and beyond the LDS capacity of the exception kernel (1024 units)."""
    rng = random.Random(11)
    docs = []
    for n in [47, 48, 49, 50, 62, 63, 64, 65, 100, 126, 127, 128, 129, 130, 191, 192, 193, 255, 256, 257, 300, 1000, 1023, 1024,
              1025, 1500, 2045, 2046, 2047, 2048, 2049, 3000, 9000]:
        docs.append(bytes(rng.choice(b"etaoinshr") for _ in range(n)))
        docs.append(b"pre " + bytes(rng.choice(b"etaoin") for _ in range(n)) + b" post")
        docs.append(("漢" * (n // 3 + 1)).encode("utf-8"))
    docs.append(b"a" * 5000 + b" " + b"b" * 2100)
    docs.append(b" " * 3000)
    docs.append(b"1" * 2500 + b"x" * 2500)
    for ctx, orc in small_byte:
        _compare(ctx, orc, docs, "long")


def test_giant_words(vg_files, oracle_mod):
    """Words of tens of thousands of units that really merge (the two-level minimum structure of
    bpe_wave_big), up to the reference's limit of 262144 bytes.  The oracle's quadratic specification loop
    checks 30000 units; the largest size is checked against the compiled reference when it is present."""
    vp, sp, kw = vg_files
    ctx = _ctx(vp, sp, kw["prefix"], kw["is_byte_encoder"])
    orc = oracle_mod.Oracle(vp, sp, kw["prefix"], kw["is_byte_encoder"])
    rng = random.Random(29)
    def blob(n):
        parts = []
        while sum(map(len, parts)) < n:
            parts.append(rng.choice([b"international", b"szolg", "árvíztűrő".encode(), b"xq", b"the", b"ation"]))
        return b"".join(parts)[:n]
    docs = [b"a " + blob(1100) + b" b", blob(5000), b"x " + blob(30000), blob(2047) + b" " + blob(1025)]
    _compare(ctx, orc, docs, "giant")
    from oracle import ref
    if ref.available():
        big = blob(262143)
        tok = ref.RefTokenizer(vp, sp, kw["prefix"], kw["is_byte_encoder"])
        want, _err = tok.encode_bytes(b"q " + big)
        from oracle import oracle as O
        data, offs = O.pack([b"q " + big])
        ids_g, oo_g, st_g, rc = ctx.encode_packed(data, offs)
        assert rc == 0 and st_g.tolist() == [0]
        assert ids_g.tolist() == list(want)


def test_words_of_65_to_1024_units(vg_files, vl_files, small_byte, oracle_mod):
    """d_exc_group_fast: two .. sixteen lanes per word, 32 .. 4 words per wavefront, block minima in registers.  Every
    length from 65 to 256 units and those around the lists' limits (128, 256, 512, 1024), enough words of each kind to
    fill wavefronts and to leave some half empty; words that merge into long tokens (live units far apart: the lane
    that owns two changed blocks searches all of its blocks); the prefix units of a character vocabulary in front."""
    from hutoken_amd import synth
    rng = random.Random(65256)
    lengths = (list(range(60, 262)) * 2 + list(range(505, 520)) + list(range(1015, 1030)) + list(range(2040, 2052))
               + [rng.randrange(257, 1025) for _ in range(60)] + [rng.randrange(1025, 2300) for _ in range(20)])
    letters = [bytes(rng.choice(b"etaoinshrdlucmfw") for _ in range(n)) for n in lengths]
    rng.shuffle(letters)
    # words of the corpus glued together: they merge back into their tokens, a dozen bytes and more each
    data, offs = synth.corpus("C3", 400)
    words = [w for w in data.tobytes().split(b" ") if 8 <= len(w) <= 40]
    words = [w for w in words if w.decode("utf-8").isalpha() and max(map(ord, w.decode("utf-8"))) < 0x250]  # (one splitter class)
    glued = []
    for _ in range(300):
        w, want = b"", rng.randrange(65, 257) if rng.random() < 0.6 else rng.randrange(257, 2200)
        while len(w) < want:
            w += rng.choice(words)
        glued.append(w[:want].decode("utf-8", "ignore").encode("utf-8"))
    same = [bytes([c]) * n for c in b"ae" for n in (65, 128, 129, 200, 256, 257, 512, 513, 1024, 1025, 2046, 2047, 2048)]
    docs = [b" ".join(letters[i:i + 7]) for i in range(0, len(letters), 7)]
    docs += [b" ".join(glued[i:i + 5]) for i in range(0, len(glued), 5)]
    docs += [b"x " + w + b" y" for w in same] + letters[:40] + glued[:40]
    vp, sp, kw = vg_files
    _compare(_ctx(vp, sp, kw["prefix"], kw["is_byte_encoder"]), oracle_mod.Oracle(vp, sp, kw["prefix"], kw["is_byte_encoder"]), docs, "65-256 VG")
    vp, sp, kw = vl_files
    _compare(_ctx(vp, sp, kw["prefix"], kw["is_byte_encoder"]), oracle_mod.Oracle(vp, sp, kw["prefix"], kw["is_byte_encoder"]), docs, "65-256 VL")
    for ctx, orc in small_byte:
        _compare(ctx, orc, docs[:120], "65-256 small")


def test_word_ends_in_later_tiles(vg_files, small_byte, oracle_mod):
    """d_exc_ends: a word whose end its tile cannot see ends at the first word start of the tiles behind it
    (Workspace::tile_first_start): words that end exactly on tile limits (multiples of 960 bytes), one to three tiles
    further on, at a document's end, at the end of the batch, with another long word or nothing behind."""
    rng = random.Random(960)
    def word(n):
        return bytes(rng.choice(b"etaoinshrdlu") for _ in range(n))
    docs = []
    for lead in (0, 1, 5, 63, 64, 100, 500, 896, 897, 959, 960, 961, 1000):
        for n in (64, 65, 100, 959 - lead % 960, 960, 961, 1024, 1025, 1919, 1920, 1921, 2880, 3000):
            if n < 64:
                continue
            pre = (word(lead - 1) + b" ") if lead else b""
            docs.append(pre + word(n))                  # the document ends with the word
            docs.append(pre + word(n) + b" x")          # a short word behind
            docs.append(pre + word(n) + b" " + word(n))  # a long one behind
    docs.append(word(70))      # the batch ends with a long word
    vp, sp, kw = vg_files
    _compare(_ctx(vp, sp, kw["prefix"], kw["is_byte_encoder"]), oracle_mod.Oracle(vp, sp, kw["prefix"], kw["is_byte_encoder"]), docs, "ends VG")
    for ctx, orc in small_byte[:2]:
        _compare(ctx, orc, docs[::3] + [word(5000)], "ends small")


def test_many_exception_words_in_one_tile(vl_files, small_char, oracle_mod):
    """k_finish's d_gather_exc: tiles of more than 256 exception words (documents of one to three bytes under a prefix
    vocabulary: every document's first word is one) take the second pass with the workgroup's whole LDS; mixed with
    tiles of a few."""
    rng = random.Random(256)
    tiny = [bytes(rng.choice(b"abcdeghi") for _ in range(rng.randrange(1, 4))) for _ in range(6000)]
    mixed = tiny[:1500] + [b"some longer document with a few words in it " * 3] * 40 + tiny[1500:3000]
    vp, sp, kw = vl_files
    ctx, orc = _ctx(vp, sp, kw["prefix"], kw["is_byte_encoder"]), oracle_mod.Oracle(vp, sp, kw["prefix"], kw["is_byte_encoder"])
    _compare(ctx, orc, tiny, "tiny VL")
    _compare(ctx, orc, mixed, "mixed VL")
    for c, o in small_char:
        _compare(c, o, mixed, "mixed small")


def test_dense_word_tiles(small_byte):
    """Tiles packed with the shortest possible words: every byte a word (newlines, stray bytes), and
    two-byte words back to back (the most multi-unit words a tile can start)."""
    docs = [b"\n" * 5000, b"\xff\x80" * 3000, b"a " * 4000, b" a" * 4000, b"ab" + b"\tab" * 3000,
            b"a1" * 3000, b".a" * 3000 + b"!" * 2000, bytes(range(1, 256)) * 20]
    for ctx, orc in small_byte:
        _compare(ctx, orc, docs, "dense")
        _compare(ctx, orc, [bytes([b]) for b in range(1, 256)] * 8, "one-byte-docs")


def test_merge_pool_overflow(small_byte):
    """More merge-loop words in a workgroup's four tiles than its pool holds (256, of which 96 long ones) or than
    fit its merge array (1600 units): the rest waits in the tiles and goes through further epochs.  Words are
    random letter strings, so nearly none is a vocabulary key and all of them need the merge loop."""
    rng = random.Random(23)
    def words(n_words, lo, hi):
        return b" ".join(bytes(rng.choice(b"qxzjkvwy") for _ in range(rng.randint(lo, hi))) for _ in range(n_words))
    docs = [words(3000, 2, 3),        # ~270 short merge words per tile
            words(2000, 10, 14),      # ~75 long merge words per tile
            words(1500, 2, 30),       # mixed
            words(1500, 24, 32),      # at the lane path's unit limit: ~50 per epoch fit the merge array
            b"\n".join(words(40, 2, 20) for _ in range(60))]
    for ctx, orc in small_byte:
        _compare(ctx, orc, docs, "pool")
        _compare(ctx, orc, [words(rng.randint(0, 400), 2, 16) for _ in range(200)], "pool-many-docs")


def test_random_text_char_mode_with_prefix(small_char):
    for k, (ctx, orc) in enumerate(small_char):
        rng = random.Random(300 + k)
        docs = [H.random_text(rng, max_words=40).encode("utf-8") for _ in range(3000)]
        docs += [b"", b" ", b"a", b" a", "漢".encode(), b"\n", b"  x"]
        _compare(ctx, orc, docs, f"char{k}")


def test_prefix_on_document_first_words(tmp_path, oracle_mod):
    """Prefix handling of a document's first word (core.c:364-366, 421-451) in the tile kernel's side
    arena and, past its capacity, the exception path: documents packed denser than the arena holds, first
    words at the unit limit, multi-unit prefixes, the leading-space form, byte mode with a prefix."""
    rng = random.Random(17)
    short = [rng.choice(["a", " a", "ab", "é", " é", "", " ", "\n", "漢", "word", " word", "Szia!"]).encode()
             for _ in range(4000)]
    limit = []
    for n in range(26, 40):
        limit.append(("e" * n).encode())
        limit.append((" " + "t" * n).encode())
        limit.append(("ő" * n + " tail").encode())
    text = [H.random_text(rng, max_words=12).encode("utf-8") for _ in range(3000)]
    spaced = [b" " + H.random_text(rng, max_words=5).encode("utf-8") for _ in range(1000)]
    for seed, prefix, is_byte in [(1, "▁", False), (2, "▁▁", False), (3, "ab", False), (4, "et a", False),
                                  (5, "Ġ", True), (6, "xy", True)]:
        if is_byte:
            ents, sp = H.random_byte_vocab(seed, n_merges=400)
        else:
            ents, sp = H.random_char_vocab(seed, n_merges=400)
        vp, spath = H.write_vocab(tmp_path, f"p{seed}", ents, sp)
        ctx = _ctx(vp, spath, prefix, is_byte)
        orc = oracle_mod.Oracle(vp, spath, prefix, is_byte)
        for name, docs in [("short", short), ("limit", limit), ("text", text), ("spaced", spaced)]:
            _compare(ctx, orc, docs, f"prefix[{prefix!r},{is_byte}]/{name}")


def test_long_words_char_mode(small_char):
    rng = random.Random(13)
    docs = []
    for n in [47, 48, 49, 100, 300, 1023, 1024, 1025, 3000]:
        docs.append("".join(rng.choice("etaoinőű") for _ in range(n)).encode("utf-8"))
        docs.append((" x " + "".join(rng.choice("漢字aé") for _ in range(n)) + " y").encode("utf-8"))
    for ctx, orc in small_char:
        _compare(ctx, orc, docs, "charlong")


def test_vg_vocab_on_corpora(vg_files, oracle_mod):
    from hutoken_amd import synth
    vp, sp, kw = vg_files
    ctx = _ctx(vp, sp, kw["prefix"], kw["is_byte_encoder"])
    orc = oracle_mod.Oracle(vp, sp, kw["prefix"], kw["is_byte_encoder"])
    for name, n in [("C2", 3000), ("C3", 3000), ("C5", 1000)]:
        data, offs = synth.corpus(name, n)
        ids_o, oo_o, _ = orc.encode_packed(data, offs, num_threads=8)
        ids_g, oo_g, st, rc = ctx.encode_packed(data, offs)
        assert rc == 0
        assert np.array_equal(oo_o, oo_g), name
        assert np.array_equal(ids_o, ids_g), name


def test_vl_vocab_on_corpora(vl_files, oracle_mod):
    from hutoken_amd import synth
    vp, sp, kw = vl_files
    ctx = _ctx(vp, sp, kw["prefix"], kw["is_byte_encoder"])
    orc = oracle_mod.Oracle(vp, sp, kw["prefix"], kw["is_byte_encoder"])
    for name, n in [("C5", 3000), ("C3", 2000)]:
        data, offs = synth.corpus(name, n)
        ids_o, oo_o, _ = orc.encode_packed(data, offs, num_threads=8)
        ids_g, oo_g, st, rc = ctx.encode_packed(data, offs)
        assert rc == 0
        assert np.array_equal(oo_o, oo_g), name
        assert np.array_equal(ids_o, ids_g), name


def test_merges_path_random_vocabularies(tmp_path, oracle_mod):
    """The id-keyed merge path (a merges file; reference core.c:211-337, 457-477, lib.c:573-663): rule order
    unrelated to id order, skipped rules, repeated pairs, shuffled and duplicate ids, with and without a
    prefix; the same kernels on different tables."""
    for seed, proper, dup, prefix in [(1, True, False, None), (2, False, False, None), (3, True, True, None),
                                      (4, True, False, "Ġ"), (5, False, False, "ab")]:
        ents, sp = H.random_byte_vocab(seed, n_merges=500, proper=proper, dup_ids=dup)
        vp, spath = H.write_vocab(tmp_path, f"m{seed}", ents, sp)
        mp = H.write_merges(tmp_path, f"m{seed}", H.random_merges_text(ents, seed * 3, keep=0.8))
        ctx = _ctx(vp, spath, prefix, True, mp)
        orc = oracle_mod.Oracle(vp, spath, prefix, True, mp)
        assert ctx.uses_merges and orc.has_merges
        rng = random.Random(400 + seed)
        docs = [H.random_text(rng, max_words=40).encode("utf-8") for _ in range(2500)]
        docs += [H.random_bytes_text(rng, rng.randint(0, 60)) for _ in range(1500)]
        docs += [b"", b" ", b"a", b" a", bytes(rng.choice(b"etaoin") for _ in range(700)), b"x" * 3000]
        _compare(ctx, orc, docs, f"merges{seed}")
    # a file without a countable line leaves the string path in force (lib.c:592)
    ents, sp = H.random_byte_vocab(7, n_merges=300)
    vp, spath = H.write_vocab(tmp_path, "m7", ents, sp)
    mp = H.write_merges(tmp_path, "m7", "#version: 0.2\n")
    ctx = _ctx(vp, spath, None, True, mp)
    assert not ctx.uses_merges
    _compare(ctx, oracle_mod.Oracle(vp, spath, None, True, mp), [b"hello world", b"  x"], "merges-empty")


def test_merges_path_char_mode(tmp_path, oracle_mod):
    """Non-byte mode on the id-keyed path: one-character replacements, and the byte-fallback literals of a Llama-style
    special file -- several units per input byte there (core.c:460-474 splits "<0x0A>" per character): the words with a
    tab or a line feed go through the exception kernels, which expand them."""
    ents, _sp = H.random_char_vocab(1, n_merges=400)
    vp, spath = H.write_vocab(tmp_path, "mc1", ents, {32: "▁"})
    mp = H.write_merges(tmp_path, "mc1", H.random_merges_text(ents, 5))
    ctx = _ctx(vp, spath, "▁", False, mp)
    orc = oracle_mod.Oracle(vp, spath, "▁", False, mp)
    rng = random.Random(501)
    docs = [H.random_text(rng, max_words=30).replace("\t", " ").replace("\n", " ").replace("\r", " ").encode("utf-8")
            for _ in range(3000)]
    _compare(ctx, orc, docs + [b"", b" ", b"a", b" a"], "merges-char")
    ents2, sp2 = H.random_char_vocab(2, n_merges=100)
    vp2, spath2 = H.write_vocab(tmp_path, "mc2", ents2, sp2)
    mp2 = H.write_merges(tmp_path, "mc2", H.random_merges_text(ents2, 6, noise=False))
    ctx2 = _ctx(vp2, spath2, "▁", False, mp2)
    orc2 = oracle_mod.Oracle(vp2, spath2, "▁", False, mp2)
    rng = random.Random(502)
    docs2 = [H.random_text(rng, max_words=30).encode("utf-8") for _ in range(2000)]  # tabs and line feeds included
    _compare(ctx2, orc2, docs2 + [b"\n", b"a\tb", b"\r\n\r\n", b" \n"], "merges-char-literals")


def test_merges_path_vg_on_corpora(vg_files, oracle_mod):
    from hutoken_amd import data, synth
    vp, sp, kw = vg_files
    mp = data.merges_file("VG")
    ctx = _ctx(vp, sp, kw["prefix"], kw["is_byte_encoder"], mp)
    orc = oracle_mod.Oracle(vp, sp, kw["prefix"], kw["is_byte_encoder"], mp)
    for name, n in [("C3", 3000), ("C2", 2000)]:
        d, o = synth.corpus(name, n)
        ids_o, oo_o, _ = orc.encode_packed(d, o, num_threads=8)
        ids_g, oo_g, st, rc = ctx.encode_packed(d, o)
        assert rc == 0
        assert np.array_equal(oo_o, oo_g), name
        assert np.array_equal(ids_o, ids_g), name


def test_python_surface(vg_files, oracle_mod):
    import hutoken_amd as hutoken
    vp, sp, kw = vg_files
    hutoken.initialize(vp, sp, **kw)
    orc = oracle_mod.Oracle(vp, sp, kw["prefix"], kw["is_byte_encoder"])
    assert hutoken.encode("hello world") == orc.encode("hello world")
    assert hutoken.encode("") == []
    texts = ["How can the net", " amount of entropy of", " the universe be massively decreased?", ""]
    for nt in (1, 3, 8):
        assert hutoken.batch_encode(texts, nt) == orc.batch_encode(texts, nt)
    assert hutoken.batch_encode(texts, 0) == [[], [], [], []]
    with pytest.raises(RuntimeError, match="hutoken: Error encoding texts"):
        hutoken.batch_encode("not a list")
    with pytest.raises(RuntimeError, match="embedded null character"):
        hutoken.encode("a\0b")
    assert hutoken.batch_encode(["ab\0cd"]) == [orc.encode("ab")]


def test_initialize_while_another_thread_encodes(vg_files, oracle_mod):
    """The shim releases the GIL around the C ABI, so initialize() on one thread can replace the module's context while
    batch_encode() runs on another: the old context lives until the last call on it has returned (no freed streams or
    device buffers under running kernels), and every call returns the ids of the vocabulary it started with."""
    import threading
    import hutoken_amd as hutoken
    from hutoken_amd import synth
    vp, sp, kw = vg_files
    orc = oracle_mod.Oracle(vp, sp, kw["prefix"], kw["is_byte_encoder"])
    d, o = synth.corpus("C3", 3000, first_doc=123_000)
    docs = synth.docs_as_str(d, o)
    want = orc.batch_encode(docs, 8)
    hutoken.initialize(vp, sp, **kw)
    stop, bad = threading.Event(), []

    def encoder():
        while not stop.is_set():
            got = hutoken.batch_encode(docs, 4)
            if got != want:
                bad.append("ids differ")
            if hutoken.encode(docs[5]) != want[5]:
                bad.append("encode differs")

    ts = [threading.Thread(target=encoder) for _ in range(2)]
    for t in ts:
        t.start()
    for _ in range(6):
        hutoken.initialize(vp, sp, **kw)
    stop.set()
    for t in ts:
        t.join()
    assert not bad
    assert hutoken.batch_encode(docs[:10], 1) == want[:10]


def test_python_surface_with_several_devices(vg_files, oracle_mod, monkeypatch):
    """initialize(..., devices=[...]) and HUTOKEN_DEVICES: batch_encode spreads the list over the device contexts."""
    import hutoken_amd as hutoken
    from hutoken_amd import synth
    vp, sp, kw = vg_files
    monkeypatch.setenv("HUTK_MULTI_MIN_BYTES", "0")
    orc = oracle_mod.Oracle(vp, sp, kw["prefix"], kw["is_byte_encoder"])
    d, o = synth.corpus("C3", 1500, first_doc=7000)
    docs = synth.docs_as_str(d, o)
    want = orc.batch_encode(docs, 8)
    hutoken.initialize(vp, sp, devices=[0, 0, 0], **kw)
    assert hutoken._ctx.device_count == 3
    assert hutoken.batch_encode(docs, 4) == want
    assert hutoken.encode(docs[3]) == want[3]
    monkeypatch.setenv("HUTOKEN_DEVICES", "0,0")
    hutoken.initialize(vp, sp, **kw)
    assert hutoken._ctx.device_count == 2
    assert hutoken.batch_encode(docs, 4) == want
    monkeypatch.delenv("HUTOKEN_DEVICES")
    hutoken.initialize(vp, sp, **kw)
    assert hutoken._ctx.device_count == 1


def test_two_streams_and_two_threads_on_one_context(vg_files, oracle_mod):
    """Calls on one context are serialised (a mutex on the host, an event on the device): launches on two streams without
    any synchronisation in between, and host calls from two threads, give the same ids as one call after the other."""
    import threading
    import torch
    from hutoken_amd import _capi, synth
    vp, sp, kw = vg_files
    ctx = _capi.Context(vp, sp, kw["prefix"], kw["is_byte_encoder"])
    orc = oracle_mod.Oracle(vp, sp, kw["prefix"], kw["is_byte_encoder"])
    dev = torch.device("cuda", 0)
    batches = []
    for k, n in enumerate((6000, 900)):  # different sizes: the second call shrinks/grows what the first one uses
        d, o = synth.corpus("C3", n, first_doc=50_000 * (k + 1))
        want = orc.encode_packed(d, o, 8)
        db, do = torch.from_numpy(d).to(dev), torch.from_numpy(o).to(dev)
        cap = ctx.ids_capacity(len(d), n)
        bufs = [(torch.empty(cap, dtype=torch.int32, device=dev), torch.empty(n + 1, dtype=torch.int64, device=dev),
                 torch.zeros(1, dtype=torch.int32, device=dev)) for _ in range(4)]
        batches.append((d, o, n, db, do, cap, bufs, want, torch.cuda.Stream(dev)))
    torch.cuda.synchronize(dev)
    for rep in range(4):  # A on stream 1, B on stream 2, A, B, ... with nothing in between
        for (d, o, n, db, do, cap, bufs, want, st) in batches:
            ids, oo, err = bufs[rep]
            ctx.encode_device(db.data_ptr(), do.data_ptr(), n, len(d), ids.data_ptr(), cap, oo.data_ptr(), 0, err.data_ptr(),
                              st.cuda_stream)
    torch.cuda.synchronize(dev)
    for (d, o, n, db, do, cap, bufs, want, st) in batches:
        for ids, oo, err in bufs:
            assert int(err.item()) == 0
            got_oo = oo.cpu().numpy()
            assert np.array_equal(got_oo, want[1])
            assert np.array_equal(ids[: int(got_oo[-1])].cpu().numpy(), want[0])
    # host entry point from two threads at once
    results = {}

    def work(k):
        d, o = batches[k][0], batches[k][1]
        for _ in range(3):
            results[k] = ctx.encode_packed(d, o)

    th = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for k in range(2):
        ids, oo, st, rc = results[k]
        assert rc == 0 and np.array_equal(oo, batches[k][7][1]) and np.array_equal(ids, batches[k][7][0])


@pytest.mark.parametrize("n_dev", [2, 3])
def test_one_batch_over_several_device_contexts(vg_files, oracle_mod, monkeypatch, n_dev):
    """hutk_ctx_add_device: the batch is cut into byte-balanced runs of whole documents, one per device context, encoded
    side by side and put together in document order.  The box has one GPU, so the same ordinal is added again: every
    step of the dispatch runs (cuts, threads, per-device tables and staging, placement), only the speed-up does not."""
    from hutoken_amd import _capi, synth
    vp, sp, kw = vg_files
    monkeypatch.setenv("HUTK_MULTI_MIN_BYTES", "0")
    ctx = _capi.Context(vp, sp, kw["prefix"], kw["is_byte_encoder"], devices=[0] * n_dev)
    assert ctx.device_count == n_dev
    orc = oracle_mod.Oracle(vp, sp, kw["prefix"], kw["is_byte_encoder"])
    d, o = synth.corpus("C3", 5000, first_doc=123_000)
    # ragged: empty documents at the front, in the middle and at the end, and one document of a third of the bytes
    cuts = o.tolist()
    big = len(d) // 3
    keep = [c for c in cuts if not (cuts[700] < c < cuts[700] + big)]
    o2 = np.array([0, 0] + keep[1:1000] + [keep[1000]] * 3 + keep[1001:] + [keep[-1]] * 2, dtype=np.int64)
    for dd, oo_in in ((d, o), (d, o2)):
        want_ids, want_oo, want_st = orc.encode_packed(dd, oo_in, 8)
        ids, oo, st, rc = ctx.encode_packed(dd, oo_in)
        assert rc == 0
        assert np.array_equal(oo, want_oo) and np.array_equal(ids, want_ids) and np.array_equal(st, want_st)
    # fewer documents than devices, and an empty batch: the context's own device takes them
    d1, o1 = synth.corpus("C3", 1)
    ids, oo, st, rc = ctx.encode_packed(d1, o1)
    assert rc == 0 and np.array_equal(ids, orc.encode_packed(d1, o1, 1)[0])
    # an error in one run is the call's error (a NUL byte in the last third)
    bad = d.copy()
    bad[len(bad) - 1000] = 0
    with pytest.raises(Exception):
        r = ctx.encode_packed(bad, o)
        if r[3] != 0:
            raise RuntimeError("rc %d" % r[3])
    # the regex pre-token path goes through the same dispatch
    ctx.set_pattern(r"[A-Za-z]+|[0-9]+|[^A-Za-z0-9 ]+| +")
    orx = oracle_mod.Oracle(vp, sp, kw["prefix"], kw["is_byte_encoder"], pattern=r"[A-Za-z]+|[0-9]+|[^A-Za-z0-9 ]+| +")
    d3, o3 = synth.corpus("C2", 800)
    want_ids, want_oo, _ = orx.encode_packed(d3, o3, 8)
    ids, oo, st, rc = ctx.encode_packed(d3, o3)
    assert rc == 0 and np.array_equal(oo, want_oo) and np.array_equal(ids, want_ids)
    ctx.close()
