#!/usr/bin/env python3
"""G8 of tests/golden: what the reference's own small tests and loaders pin, as DATA produced by the compiled
reference (oracle/_ref, built from /root/reference by oracle/Makefile).  Run in the build container only.

  loaders  quirk files for the vocabulary and special-character loaders (src/lib.c:243-388, 460-571): no trailing
           newline, repeated keys, a 0x00 first byte, an over-long token, empty files, missing separators, long special
           values, ... -> the ids of probe texts, or the exception class + message of the reference's initialize()
  parser   the 28 strings of tests/test_parser.c (and a few more) -> the word boundaries the reference's own
           parser_next_token returns (src/parser.c:24-88)
  queue    push / pop sequences of the reference's MinPQ (src/queue.c:152-199; tests/test_queue.c:81-121, 146-178:
           equal ranks come out in left_idx order)
  ties     texts with repeated pairs x G1's byte vocabulary -> ids (the leftmost of equal ranks merges first)
  --g9     g9_regex_path.json: the regex pre-token path (initialize(pattern=...)): POSIX EREs x seeded texts -> ids

Files that make the reference write out of bounds (special index 256, ids >= the line count) are NOT run through it;
they are deliberate deviations.
"""
import ctypes as C
import hashlib
import json
import os
import random
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import helpers as H  # noqa: E402
from hutoken_amd import vocab_files as vf  # noqa: E402
from oracle import ref  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")

# tests/test_parser.c:77-108 (the reference's golden master: parser_next_token against a POSIX ERE)
PARSER_STRINGS = [
    "", "Hello", " Hello", "árvíztűrő", " árvíztűrő",
    "ÁrvíztűrőTükör", "12345", " 123", "!!!@#$", " !!!", "!@#$%^&*()_+", "€",
    "\U0001F602", " ", "   ", "word123", "word.!_", "123word", "123.!", ".!word", ".!123", "\t\n\r\f\v", "A  B   C",
    "word ", " First Second", "Hello world 123. End!",
    "árvíztűrő tükörfúrógép.",
    "This is a test 123. With some special chars: !@# and spaces. "
    "árvíztűrő tükörfúrógép!"]
MORE_SPLITS = ["a  b", "x \ta", "a b", "abcd", "漢字 仮名", "a\U0001F602b", "ab", " \t ", "1 2  3   4"]
TIE_TEXTS = ["aaaa", "aaaaa", "aaaaaa", "aaaaaaa", "abab", "ababab", "abababab", "aabaab", "thethethe", "tthetthe",
             "lololo", "hellohello", " the the", "ththth", "abaabaaba", "aaabaaab"]


def quirk_files():
    """name -> (vocab file bytes, special file bytes or None for the GPT-2 one, probe texts)"""
    t = vf.bytes_to_unicode()

    def line(tok, i):
        return vf.hex_line(vf.encode_visible(tok, t), i).encode()

    base = b"".join(line(bytes([b]), i) for i, b in enumerate(vf.byte_token_order()))
    more = line(b"th", 256) + line(b"he", 257) + line(b"the", 258) + line(b" t", 259) + line(b"aa", 260)
    probes = ["the then", "aaaa the", " the", "hathe"]
    long_tok = b"0x61" * 2048 + b" == 300\n"  # 2048 token bytes: over the limit (lib.c:341-343, helper.c:101-111)
    ok_long = b"0x61" * 2047 + b" == 300\n"
    sp31 = "".join("%d == %s\n" % (b, t[b]) for b in vf.SPECIAL_BYTES).encode()
    g_dot = "Ġ".encode()  # the GPT-2 replacement of the space
    return {
        "well_formed": (base + more, None, probes),
        "last_line_without_newline": (base + more[:-1], None, probes),            # the final line is dropped (lib.c:264-289)
        "repeated_key_last_id_wins": (base + more + line(b"th", 999), None, probes),  # hashmap.c:208-214
        "repeated_key_in_the_middle": (base + line(b"th", 300) + more, None, probes),
        "first_byte_zero": (base + b"0x00 == 300\n", None, probes),               # ValueError (lib.c:345-355)
        "zero_inside_a_key": (base + more + b"0x740x000x68 == 301\n", None, probes),  # the key ends at the 0x00
        "token_of_2048_bytes": (base + long_tok, None, probes),
        "token_of_2047_bytes": (base + more + ok_long, None, probes + ["a" * 2047, "a" * 30]),
        "empty_vocab": (b"", None, probes),
        "only_line_without_newline": (b"0x61 == 0", None, probes),
        "missing_separator": (base + b"0x610x62 5\n", None, probes),
        "separator_without_value": (base + b"0x610x62 == \n", None, probes),
        "non_numeric_value": (base + b"0x610x62 == x7\n", None, probes),
        "value_out_of_int_range": (base + b"0x610x62 == 99999999999999999999\n", None, probes),
        "negative_id": (base + more + line(b"an", -5), None, probes + ["an and"]),
        "upper_case_marker": (base + more + b"0X610X6E == 400\n", None, probes + ["an"]),   # "0X" is no hex marker
        "junk_between_bytes": (base + more + b"0x61zz0x6E == 401\n", None, probes + ["an"]),
        "blank_line": (base + b"\n" + more, None, probes),
        "crlf_lines": ((base + more).replace(b"\n", b"\r\n"), None, probes),
        "special_last_line_without_newline": (base + more, sp31[:-1], probes + [" a b", "\t"]),  # last character dropped (lib.c:527-531)
        "special_empty": (base + more, b"", probes),
        "special_missing_separator": (base + more, b"32 " + g_dot + b"\n", probes),
        "special_long_value": (base + more, b"32 == " + g_dot * 14 + b"\n", probes),   # beyond the 31-character chunk (lib.c:483)
        "special_value_of_one_chunk": (base + more, b"32 == " + b"x" * 20 + b"\n", probes),
        "special_negative_index": (base + more, b"-1 == x\n", probes),
        "special_index_not_a_number": (base + more, b"x == y\n", probes),
        "special_repeated_index": (base + more, sp31 + b"32 == Q\n", probes + [" a b"]),
    }


def write_case_files(tmp, name, vocab, special):
    """-> (vocab path, special path); special None = the GPT-2 special file"""
    vp = os.path.join(tmp, "g8_%s_vocab.txt" % name)
    with open(vp, "wb") as f:
        f.write(vocab)
    sp = os.path.join(tmp, "g8_%s_special.txt" % name)
    if special is None:
        vf.write_special_file(sp, vf.gpt2_special_mapping())
    else:
        with open(sp, "wb") as f:
            f.write(special)
    return vp, sp


def g8(tmp):
    out = {"loaders": [], "parser": [], "queue": [], "ties": []}
    for name, (vocab, special, probes) in quirk_files().items():
        vp, sp = write_case_files(tmp, name, vocab, special)
        case = {"name": name, "vocab_sha256": hashlib.sha256(vocab).hexdigest(),
                "special_sha256": None if special is None else hashlib.sha256(special).hexdigest(), "probes": probes}
        try:
            r = ref.RefTokenizer(vp, sp, None, True)
            case["ids"] = [r.encode(x) for x in probes]
        except Exception as e:  # noqa: BLE001
            case["error"] = [type(e).__name__, str(e)]
        out["loaders"].append(case)

    L = C.CDLL(ref.so_path())

    class TokenSlice(C.Structure):      # parser.h:7-10
        _fields_ = [("start", C.c_void_p), ("length", C.c_size_t)]

    class ParserState(C.Structure):     # parser.h:12-14
        _fields_ = [("current_pos", C.c_void_p)]

    L.parser_init.restype = ParserState
    L.parser_init.argtypes = [C.c_void_p]
    L.parser_next_token.restype = C.c_bool
    L.parser_next_token.argtypes = [C.POINTER(ParserState), C.POINTER(TokenSlice)]
    for text in PARSER_STRINGS + MORE_SPLITS + TIE_TEXTS:
        buf = C.create_string_buffer(text.encode("utf-8"))
        base_addr = C.addressof(buf)
        st = L.parser_init(base_addr)
        tok = TokenSlice()
        starts = []
        while L.parser_next_token(C.byref(st), C.byref(tok)):
            starts.append(tok.start - base_addr)
            assert len(starts) < 1000
        out["parser"].append({"text": text, "word_starts": starts})

    class MergeCandidate(C.Structure):  # queue.h:9-13
        _fields_ = [("rank", C.c_int), ("left_idx", C.c_size_t), ("right_idx", C.c_size_t)]

    class MinPQ(C.Structure):           # queue.h:15-19
        _fields_ = [("data", C.c_void_p), ("size", C.c_size_t), ("capacity", C.c_size_t)]

    L.min_pq_init.argtypes = [C.POINTER(MinPQ), C.c_size_t]
    L.min_pq_push.argtypes = [C.POINTER(MinPQ), MergeCandidate]
    L.min_pq_pop.argtypes = [C.POINTER(MinPQ), C.POINTER(MergeCandidate)]
    L.min_pq_release.argtypes = [C.POINTER(MinPQ)]
    rng = random.Random(8)
    seqs = [[(10, 1), (5, 2), (10, 3), (5, 4)],                      # tests/test_queue.c:146-178
            [(20, 1), (5, 2), (15, 3), (10, 4)]]                     # tests/test_queue.c:81-121
    for _ in range(40):
        pos = list(range(30))
        rng.shuffle(pos)
        seqs.append([(rng.randrange(4), pos[k]) for k in range(rng.randint(1, 24))])  # distinct positions, few ranks
    for pushes in seqs:
        pq = MinPQ()
        L.min_pq_init(C.byref(pq), 4)
        for r, li in pushes:
            L.min_pq_push(C.byref(pq), MergeCandidate(r, li, li + 1))
        pops = []
        c = MergeCandidate()
        while L.min_pq_pop(C.byref(pq), C.byref(c)) == 0:
            pops.append([c.rank, c.left_idx])
        L.min_pq_release(C.byref(pq))
        out["queue"].append({"pushes": [list(x) for x in pushes], "pops": pops})

    t = vf.bytes_to_unicode()
    with open(os.path.join(OUT, "g1_handpicked.json")) as f:
        g1j = json.load(f)
    raw = [bytes([b]) for b in vf.byte_token_order()] + [bytes.fromhex(m) for m in g1j["byte_vocab"]["merges_hex"]]
    vp, sp = H.write_vocab(tmp, "g8t", [(vf.encode_visible(tok, t), i) for i, tok in enumerate(raw)],
                           vf.gpt2_special_mapping())
    r = ref.RefTokenizer(vp, sp, None, True)
    out["ties"] = [{"text": x, "ids": r.encode(x)} for x in TIE_TEXTS + PARSER_STRINGS + MORE_SPLITS]
    return out


# G9: the regex pre-token path (initialize(pattern=...), core.c:350-360).  POSIX EREs x seeded texts -> ids.
GPT2_POSIX = ("[ ]?[A-Za-z\u00e1\u00e9\u00ed\u00f3\u00fa\u0151\u0171\u00fc\u00f6\u00c1\u00c9\u00cd\u00d3\u00da\u0150\u00dc\u0170\u00d6]+|[ ]?[0-9]+|"
              "[ ]?[^[:space:][:alpha:][:digit:]]+|[ ]+")  # tests/test_parser.c:10-12
G9_PATTERNS = [GPT2_POSIX, "[a-z]+", " ?[[:alpha:]]+| ?[[:digit:]]+| ?[^[:space:][:alpha:][:digit:]]+|[[:space:]]+",
               ".+", "x*", "[^ ]+", "(th|he)+", "^[a-z]+", "[[:alpha:]]+[[:space:]]?", "a|"]


def g9_texts(seed, n):
    rng = random.Random(seed)
    return [H.random_text(rng, max_words=25) for _ in range(n)]


def g9(tmp):
    """needs LC_CTYPE = a UTF-8 locale (CPython's default C.UTF-8): recorded with the fixture"""
    import locale
    from hutoken_amd import data, synth
    out = {"lc_ctype": locale.setlocale(locale.LC_CTYPE, None), "mid": [], "vg": None}
    ents, sp = H.random_byte_vocab(11, n_merges=2000)
    vp, spath = H.write_vocab(tmp, "g9", ents, sp)
    texts = g9_texts(9000, 600)
    for pat in G9_PATTERNS:
        r = ref.RefTokenizer(vp, spath, None, True, pattern=pat)
        res = r.batch_encode(texts, 4)
        h = hashlib.sha256()
        for ids in res:
            h.update(json.dumps(ids).encode())
        out["mid"].append({"pattern": pat, "first": res[:25], "n_ids": sum(len(x) for x in res), "sha256": h.hexdigest()})
    vp, sp, kw = data.vocab_files("VG")
    r = ref.RefTokenizer(vp, sp, kw["prefix"], kw["is_byte_encoder"], pattern=GPT2_POSIX)
    d, o = synth.corpus("C3", 2000)
    res = r.batch_encode(synth.docs_as_str(d, o), 8)
    h = hashlib.sha256()
    for ids in res:
        h.update(json.dumps(ids).encode())
    out["vg"] = {"pattern": GPT2_POSIX, "corpus": "C3", "n_docs": 2000, "first": res[:10],
                 "n_ids": sum(len(x) for x in res), "sha256": h.hexdigest()}
    return out


def main():
    assert ref.available(), "build the reference first: make -C oracle ref"
    if "--g9" in sys.argv:
        res = g9(tempfile.mkdtemp())
        with open(os.path.join(OUT, "g9_regex_path.json"), "w") as f:
            json.dump(res, f, ensure_ascii=True, indent=0)
        print(res["lc_ctype"], [(m["pattern"][:20], m["n_ids"]) for m in res["mid"]], res["vg"]["n_ids"])
        return
    res = g8(tempfile.mkdtemp())
    with open(os.path.join(OUT, "g8_reference_fixtures.json"), "w") as f:
        json.dump(res, f, ensure_ascii=True, indent=0)
    for c in res["loaders"]:
        print(c["name"], c.get("error") or [len(x) for x in c["ids"]][:6])
    print(len(res["parser"]), "parser cases,", len(res["queue"]), "queue cases,", len(res["ties"]), "tie texts")


if __name__ == "__main__":
    main()
