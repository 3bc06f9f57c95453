#!/usr/bin/env python3
"""G10 of tests/golden: special-character replacements of SEVERAL units, as DATA produced by the compiled reference
(oracle/_ref, built from /root/reference by oracle/Makefile).  Run in the build container only.

  pretokenizer  the inputs of the reference's tests/test_pretokenizer.c:23-251 (text, replacement table, prefix,
                byte-encoder flag) -> the string its pretokenizer_encode returns (src/pretokenizer.c:102-168, called
                through ctypes), and -- through initialize() on a vocabulary whose keys are the units of those strings
                plus a few of their pairs -- the ids its encode() returns for the same text
  files         seeded vocabularies of both shapes with replacement values of two to six units (and Llama-style
                "<0xHH>" files together with a merges file: the id-keyed path splits them per character,
                src/core.c:460-474) x seeded texts -> ids (first documents in full, all of them as a hash)
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

OUT = os.path.join(ROOT, "tests", "golden", "g10_pretokenizer.json")

# tests/test_pretokenizer.c:23-251: (name, text, {byte: replacement}, prefix, is_byte_encoder)
CASES = [
    ("no_replacements_needed", "hello world", {}, None, False),
    ("single_replacement_at_start", "apple", {ord("a"): "Alpha"}, None, False),
    ("single_replacement_at_end", "apple", {ord("e"): "End"}, None, False),
    ("single_replacement_in_middle", "apple", {ord("p"): "P"}, None, False),
    ("multiple_different_replacements", "apple", {ord("a"): "ab", ord("e"): "ef"}, None, False),
    ("multiple_occurrences_of_same_char", "banana", {ord("a"): "o"}, None, False),
    ("replacement_with_empty_string", "hello", {ord("l"): ""}, None, False),
    ("empty_input_string", "", {ord("a"): "b"}, None, False),
    ("all_chars_are_replaced", "abc", {ord("a"): "1", ord("b"): "22", ord("c"): "333"}, None, False),
    ("replacement_with_single_char", "test", {ord("t"): "T"}, None, False),
    ("with_prefix_and_replacements", "apple", {ord("a"): "A", ord("e"): "E"}, "Juicy ", False),
    ("with_prefix_and_empty_string_replacement", "-abc-", {ord("-"): ""}, "Prefix:", False),
    ("with_prefix_and_empty_input_string", "", {}, "Start:", False),
    ("with_empty_string_as_prefix", "test", {}, "", False),
    ("with_multibyte_char", "midőn", {145: "ĳ"}, None, True),
]


def reference_pretokenizer(text, repl, prefix, is_byte):
    L = C.CDLL(ref.so_path())
    L.pretokenizer_encode.restype = C.c_void_p
    L.pretokenizer_encode.argtypes = [C.c_char_p, C.POINTER(C.c_char_p), C.c_char_p, C.c_bool]
    table = (C.c_char_p * 256)()
    for b, s in repl.items():
        table[b] = s.encode("utf-8")
    p = L.pretokenizer_encode(text.encode("utf-8"), table, None if prefix is None else prefix.encode("utf-8"), is_byte)
    out = C.string_at(p)
    C.CDLL(None).free(C.c_void_p(p))
    return out


def units_of(enc):
    """The reference's unit rule on an encoded word (core.c:35-55): a "<0x..>" literal or one UTF-8 character."""
    out, i = [], 0
    while i < len(enc):
        n = 1 if enc[i] < 0x80 else 2 if enc[i] < 0xE0 else 3 if enc[i] < 0xF0 else 4
        out.append(enc[i:i + n])
        i += n
    return out


def main():
    random.seed(10)
    tmp = tempfile.mkdtemp(prefix="g10_")
    cases = []
    for name, text, repl, prefix, is_byte in CASES:
        enc = reference_pretokenizer(text, repl, prefix, is_byte)
        case = {"name": name, "text": text, "replacements": {str(k): v for k, v in repl.items()}, "prefix": prefix,
                "is_byte_encoder": is_byte, "encoded_hex": enc.hex()}
        # the whole path: a vocabulary of the units that can occur (every byte's own form, the replacements' units, the
        # prefix's) plus the first two adjacent pairs of the encoded string; files as initialize() reads them.  The
        # special-file loader refuses an empty value (lib.c:533-543), so those two cases stop at the pretokenizer.
        if all(v for v in repl.values()):
            toks = []
            for b in range(1, 256):
                if is_byte:
                    toks.append(vf.encode_visible(bytes([b])) if b not in repl else None)
                elif b < 0x80:
                    toks.append(bytes([b]))
            for v in repl.values():
                toks += units_of(v.encode("utf-8"))
            if prefix:
                toks += units_of(prefix.encode("utf-8"))
            toks += [c.encode("utf-8") for c in "őĳÅ"]
            us = units_of(enc)
            for k in (0, 2):
                if k + 1 < len(us):
                    toks.append(us[k] + us[k + 1])
            seen, entries = set(), []
            for t in toks:
                if t and t not in seen:
                    seen.add(t)
                    entries.append((t, len(entries)))
            vp, sp = H.write_vocab(tmp, name, entries, repl)
            r = ref.RefTokenizer(vp, sp, prefix if prefix else None, is_byte)
            case["vocab"] = [[t.hex(), i] for t, i in entries]
            case["ids"] = r.encode(text)
            case["ids_in_sentence"] = r.encode("x " + text + " y " + text)
        cases.append(case)

    files = []
    for seed in range(4):
        rng = random.Random(100 + seed)
        if seed < 2:   # byte-encoder shape: a few ASCII bytes become strings of visible characters
            entries, special = H.random_byte_vocab(40 + seed, n_merges=300, proper=seed == 0)
            special = dict(special)
            special[ord("q")] = "qu"
            special[ord("z")] = "zzz"
            special[ord("!")] = "!?!"
            special[10] = "ĊĊ"    # the line feed's visible form, twice
            prefix, is_byte, merges = None, True, None
        else:          # character shape, Llama-style "<0xHH>" values; seed 3 with a merges file (id-keyed path)
            entries, special = H.random_char_vocab(40 + seed, n_merges=250)
            special = dict(special)
            special[ord("q")] = "qu"
            prefix, is_byte = "▁", False
            merges = H.random_merges_text(entries, seed, noise=False) if seed == 3 else None
        vp, sp = H.write_vocab(tmp, "f%d" % seed, entries, special)
        mp = H.write_merges(tmp, "f%d" % seed, merges) if merges else None
        r = ref.RefTokenizer(vp, sp, prefix, is_byte, mp)
        texts = [H.random_text(rng, 12) + rng.choice(["", " quiz!", "\nq z\n", "\tzq!", " q"]) for _ in range(400)]
        ids = [r.encode(t) for t in texts]
        flat = [i for d in ids for i in d]
        files.append({"seed": seed, "kind": "byte" if is_byte else "char", "merges": merges is not None,
                      "special": {str(k): v for k, v in special.items()}, "prefix": prefix, "n_texts": len(texts),
                      "first": ids[:12], "n_ids": len(flat),
                      "sha256": hashlib.sha256(b"".join(int(i).to_bytes(4, "little", signed=True) for i in flat)).hexdigest()})
    with open(OUT, "w") as f:
        json.dump({"pretokenizer": cases, "files": files}, f, ensure_ascii=False, indent=0)
    print("wrote", OUT, len(cases), "cases,", len(files), "files")


if __name__ == "__main__":
    main()
