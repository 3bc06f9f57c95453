#!/usr/bin/env python3
"""Generates tests/golden/*.json by running THE REFERENCE ITSELF (compiled from
/root/reference by `make -C oracle ref`) on seeded inputs built by this repo's own
generators.  Only data is committed: inputs (or the recipe that regenerates them
bit-for-bit) and the reference's outputs.  Run in the build container only.

  G1  tiny GPT-2-shaped byte vocab (256 byte tokens + hand merges) x hand-picked strings
      covering every splitter / merge / prefix quirk           -> ids in full
  G2  mid vocabularies (seeded, proper and shuffled ids)  x 1500 seeded texts each
                                                               -> ids of the first 40 texts + sha256 of all
  G3  VG (data/vg50257_*) x first 5000 docs of C2 and of C3   -> ids of the first 48 docs + count + sha256
  G4  VL (data/vl32000_*, prefix U+2581, is_byte_encoder=False) x first 3000 docs of C5 and
      2000 docs of C3 (unknown characters -> -1)              -> same
  G5  the ID-KEYED merge path (a merges file): mid vocabularies with shuffled / duplicate ids and a
      merges file whose rule order is unrelated to the ids (helpers.random_merges_text: comments,
      skipped rules, repeated pairs, CRLF), with and without a prefix, x 1500 seeded texts; VG with
      its merges file x 2000 docs of C3                       -> same
  G7  FULL-SIZE configurations, every document: VG x all 1,000,000 docs of C3 (BASELINE config 3/4),
      VG x all 100,000 docs of C2 (config 2), VL x all 1,000,000 docs of C5 (config 5), VG + merges x C3
      -> per 100k-document block: id count, sha256 of the ids, sha256 of the block-relative offsets
         (helpers.block_hashes); whole batch: id count and sha256
  G6  the DECODE direction: mid vocabularies (byte mode proper / shuffled ids, character mode with prefix)
      x 1200 id sequences each (round trips and random ids)   -> text or exception of the first 60 + sha256
"""
import hashlib
import json
import os
import random
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402

import helpers as H  # noqa: E402
from hutoken_amd import data, synth, vocab_files as vf  # noqa: E402
from oracle import ref  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")

G1_STRINGS = [
    "", " ", "  ", "   ", "a", " a", "  a", "a ", "a  b", "a   b   c", "A  B   C", "aaaaa", "aaaa", "aaa", "abab",
    "hello world", "Hello", " Hello", "Hello world 123. End!", "word123", "word.!_", "123word", "123.!", ".!word",
    ".!123", "12345", " 123", "!!!@#$", " !!!", "!@#$%^&*()_+", "\t\n\r\f\v", "x \ta", "a\tb", "a\n\nb", "a \n b",
    "word ", " First Second", "árvíztűrő", " árvíztűrő", "ÁrvíztűrőTükör", "árvíztűrő tükörfúrógép.",
    "€", "😂", "😂😂", " 😂", "a😂b", "漢字", " 漢字 仮名", "中文測試中文測試", "a b", "ab", " ", "naïve café",
    "ä ö ü ß", "Ő Ű ő ű", "x▁y", "<0x41>", "<|endoftext|>", "a<b", "1<2", "the the the", "tthe", "thethe",
    "This is a test 123. With some special chars: !@# and spaces. árvíztűrő tükörfúrógép!",
    "How can the net amount of entropy of the universe be massively decreased?",
    "What I cannot create, I do not understand.", "e" * 40, "ab" * 30, " " * 9 + "x", "x" + " " * 9,
    chr(1) + chr(2), chr(127), "a" + chr(127) + "b", "Ab Cd eF", "I'm", "don't", "U.S.A.", "3.14159", "1,000,000",
    "a--b", "--", "a_b",
]


def sha_ids(list_of_lists):
    h = hashlib.sha256()
    for ids in list_of_lists:
        h.update(np.asarray(ids, dtype="<i4").tobytes())
        h.update(b"|")
    return h.hexdigest()


def g1(tmp):
    t = vf.bytes_to_unicode()
    merges = [b"th", b"he", b"the", b" t", b" th", b" the", b"in", b"er", b"an", b" a", b"aa", b"ab", b"abab",
              b"ll", b"lo", b"llo", b"hello", b" w", b"or", b"ld", b"wor", b"world", b" world", b"  ", b"   ",
              b"12", b"123", b" 1", b"!!", b"..", "é".encode(), "á".encode(), "ő".encode(), "ű".encode(),
              "ár".encode(), "víz".encode(), "漢".encode(), "字".encode(), "漢字".encode(), "😂".encode(),
              b"\xe2\x82", "€".encode(), b"\n\n", b"tt", b"tthe"]
    raw = [bytes([b]) for b in vf.byte_token_order()] + merges
    entries = [(vf.encode_visible(tok, t), i) for i, tok in enumerate(raw)]
    vp, sp = H.write_vocab(tmp, "g1", entries, vf.gpt2_special_mapping())
    r = ref.RefTokenizer(vp, sp, None, True)
    cases = [{"text": s, "ids": r.encode(s)} for s in G1_STRINGS]
    batch = r.batch_encode(G1_STRINGS, 3)
    assert batch == [c["ids"] for c in cases]
    # Llama-shaped twin: prefix, <0xHH> literals, -1 for unknown characters
    centries, cspecial = H.random_char_vocab(5, n_merges=300, drop_chars="qző漢")
    vp2, sp2 = H.write_vocab(tmp, "g1c", centries, cspecial)
    r2 = ref.RefTokenizer(vp2, sp2, "▁", False)
    ccases = [{"text": s, "ids": r2.encode(s)} for s in G1_STRINGS]
    return {"byte_vocab": {"merges_hex": [m.hex() for m in merges], "cases": cases},
            "char_vocab": {"recipe": "helpers.random_char_vocab(5, n_merges=300, drop_chars='qző漢'), prefix U+2581",
                           "cases": ccases}}


def g2(tmp):
    out = []
    for seed, proper, dup in [(11, True, False), (12, False, False), (13, True, True)]:
        ents, sp = H.random_byte_vocab(seed, n_merges=2000, proper=proper, dup_ids=dup)
        vp, spath = H.write_vocab(tmp, f"g2_{seed}", ents, sp)
        r = ref.RefTokenizer(vp, spath, None, True)
        rng = random.Random(seed * 1000)
        texts = [H.random_text(rng, max_words=30) for _ in range(1500)]
        res = r.batch_encode(texts, 4)
        out.append({"recipe": f"helpers.random_byte_vocab({seed}, n_merges=2000, proper={proper}, dup_ids={dup}); "
                              f"texts: random.Random({seed * 1000}), helpers.random_text(rng, max_words=30) x 1500",
                    "seed": seed, "proper": proper, "dup_ids": dup,
                    "first": res[:40], "n_ids": sum(len(x) for x in res), "sha256": sha_ids(res)})
    return out


def g5(tmp):
    out = []
    for seed, proper, dup, prefix in [(21, False, False, None), (22, True, True, None), (23, True, False, "Ġ")]:
        ents, sp = H.random_byte_vocab(seed, n_merges=2000, proper=proper, dup_ids=dup)
        vp, spath = H.write_vocab(tmp, f"g5_{seed}", ents, sp)
        mp = H.write_merges(tmp, f"g5_{seed}", H.random_merges_text(ents, seed * 3, keep=0.8))
        r = ref.RefTokenizer(vp, spath, prefix, True, mp)
        rng = random.Random(seed * 1000)
        texts = [H.random_text(rng, max_words=30) for _ in range(1500)]
        res = r.batch_encode(texts, 4)
        out.append({"recipe": f"helpers.random_byte_vocab({seed}, n_merges=2000, proper={proper}, dup_ids={dup}); "
                              f"merges: helpers.random_merges_text(entries, {seed * 3}, keep=0.8); prefix {prefix!r}; "
                              f"texts: random.Random({seed * 1000}), helpers.random_text(rng, max_words=30) x 1500",
                    "seed": seed, "proper": proper, "dup_ids": dup, "prefix": prefix,
                    "first": res[:40], "n_ids": sum(len(x) for x in res), "sha256": sha_ids(res)})
    vp, sp, kw = data.vocab_files("VG")
    r = ref.RefTokenizer(vp, sp, kw["prefix"], kw["is_byte_encoder"], data.merges_file("VG"))
    out.append(dict(corpus_case(r, "C3", 2000), vocab="VG+merges"))
    return out


def g6(tmp):
    """Decode direction: the reference's own decode() on seeded id sequences.  Unique-id vocabularies (with
    repeated ids the reference's table depends on its hash map).  Results that are not valid UTF-8 are recorded
    as the exception they raise."""
    out = []
    for seed, proper, mode in [(31, True, "byte"), (32, False, "byte"), (33, True, "char")]:
        if mode == "byte":
            ents, sp = H.random_byte_vocab(seed, n_merges=1500, proper=proper)
            prefix, is_byte = None, True
        else:
            ents, sp = H.random_char_vocab(seed, n_merges=1500)
            prefix, is_byte = "▁", False
        vp, spath = H.write_vocab(tmp, f"g6_{seed}", ents, sp)
        r = ref.RefTokenizer(vp, spath, prefix, is_byte)
        rng = random.Random(seed * 1000)
        cases = []
        for k in range(1200):
            if k % 2 == 0:
                ids = [x for x in r.encode(H.random_text(rng, max_words=20)) if x >= 0]
            else:
                ids = [rng.randrange(0, len(ents)) for _ in range(rng.randint(0, 24))]
            try:
                res = r.decode(ids)
            except Exception as e:  # noqa: BLE001
                res = {"raises": type(e).__name__}
            cases.append((ids, res))
        h = hashlib.sha256()
        for ids, res in cases:
            h.update(json.dumps([ids, res], ensure_ascii=True).encode())
        out.append({"seed": seed, "proper": proper, "mode": mode, "first": cases[:60], "n": len(cases),
                    "sha256": h.hexdigest(),
                    "recipe": "even k: ids of helpers.random_text(rng, max_words=20) without -1; odd k: random ids"})
    return out


def g7():
    """The reference over every document of the full-size configurations, 100k documents per
    batch_encode call (8 threads); only counts and hashes are kept."""
    out = []
    for vocab, corpus, merges in [("VG", "C3", False), ("VG", "C2", False), ("VL", "C5", False), ("VG", "C3", True)]:
        vp, sp, kw = data.vocab_files(vocab)
        r = ref.RefTokenizer(vp, sp, kw["prefix"], kw["is_byte_encoder"], data.merges_file(vocab) if merges else None)
        n_docs = synth.KINDS[corpus][2]
        blocks, whole, n_ids, n_bytes = [], hashlib.sha256(), 0, 0
        csha = hashlib.sha256()
        for first in range(0, n_docs, H.BLOCK_DOCS):
            cnt = min(H.BLOCK_DOCS, n_docs - first)
            d, o = synth.corpus(corpus, cnt, first_doc=first)
            csha.update(d.tobytes())
            n_bytes += int(o[-1])
            res = r.batch_encode(synth.docs_as_str(d, o), 8)
            oo = np.zeros(cnt + 1, dtype=np.int64)
            np.cumsum(np.fromiter(map(len, res), dtype=np.int64, count=cnt), out=oo[1:])
            ids = np.fromiter((t for x in res for t in x), dtype=np.int32, count=int(oo[-1]))
            blocks += H.block_hashes(ids, oo)
            whole.update(ids.astype("<i4").tobytes())
            n_ids += len(ids)
            print(vocab, corpus, merges, first, n_ids, flush=True)
        out.append({"vocab": vocab, "corpus": corpus, "merges": merges, "n_docs": n_docs, "n_bytes": n_bytes,
                    "corpus_sha256": csha.hexdigest(), "n_ids": n_ids, "sha256": whole.hexdigest(),
                    "block_docs": H.BLOCK_DOCS, "blocks": blocks})
    return out


def corpus_case(tok, name, n_docs, nfirst=48):
    d, o = synth.corpus(name, n_docs)
    docs = synth.docs_as_str(d, o)
    res = tok.batch_encode(docs, 8)
    return {"corpus": name, "n_docs": n_docs, "corpus_sha256": hashlib.sha256(d.tobytes()).hexdigest(),
            "first": res[:nfirst], "n_ids": sum(len(x) for x in res), "sha256": sha_ids(res)}


def main():
    assert ref.available(), "build the reference first: make -C oracle ref"
    os.makedirs(OUT, exist_ok=True)
    tmp = tempfile.mkdtemp()
    if "--only-g5" in sys.argv:
        json.dump(g5(tmp), open(os.path.join(OUT, "g5_merges_path.json"), "w"))
        return
    if "--only-g7" in sys.argv:
        json.dump(g7(), open(os.path.join(OUT, "g7_full.json"), "w"), indent=0)
        return
    if "--only-g6" in sys.argv:
        json.dump(g6(tmp), open(os.path.join(OUT, "g6_decode.json"), "w"), ensure_ascii=True)
        return
    json.dump(g1(tmp), open(os.path.join(OUT, "g1_handpicked.json"), "w"), ensure_ascii=True, indent=0)
    json.dump(g2(tmp), open(os.path.join(OUT, "g2_mid_vocabs.json"), "w"))
    vp, sp, kw = data.vocab_files("VG")
    r = ref.RefTokenizer(vp, sp, kw["prefix"], kw["is_byte_encoder"])
    g3 = [corpus_case(r, "C2", 5000), corpus_case(r, "C3", 5000), corpus_case(r, "C5", 1000)]
    g3.append({"text": "hello world", "ids": r.encode("hello world")})
    json.dump(g3, open(os.path.join(OUT, "g3_vg_corpora.json"), "w"))
    vp, sp, kw = data.vocab_files("VL")
    r = ref.RefTokenizer(vp, sp, kw["prefix"], kw["is_byte_encoder"])
    g4 = [corpus_case(r, "C5", 3000), corpus_case(r, "C3", 2000)]
    json.dump(g4, open(os.path.join(OUT, "g4_vl_corpora.json"), "w"))
    json.dump(g5(tmp), open(os.path.join(OUT, "g5_merges_path.json"), "w"))
    json.dump(g6(tmp), open(os.path.join(OUT, "g6_decode.json"), "w"), ensure_ascii=True)
    json.dump(g7(), open(os.path.join(OUT, "g7_full.json"), "w"), indent=0)
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
