"""Shared builders for small vocabularies and adversarial texts (own code, seeded)."""
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from hutoken_amd import vocab_files as vf  # noqa: E402

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")

BLOCK_DOCS = 100_000


def block_hashes(ids, oo, block=BLOCK_DOCS):
    """Per block of `block` documents: id count, sha256 of the ids (int32 LE), sha256 of the block-relative
    offsets (int64 LE).  The unit of tests/golden/g7_full.json."""
    import hashlib

    import numpy as np
    ids = np.asarray(ids)
    oo = np.asarray(oo, dtype=np.int64)
    out = []
    n = len(oo) - 1
    for a in range(0, n, block):
        b = min(a + block, n)
        seg = ids[int(oo[a]):int(oo[b])].astype("<i4", copy=False)
        rel = (oo[a:b + 1] - oo[a]).astype("<i8", copy=False)
        out.append({"n_ids": int(oo[b] - oo[a]), "ids_sha256": hashlib.sha256(seg.tobytes()).hexdigest(),
                    "offsets_sha256": hashlib.sha256(rel.tobytes()).hexdigest()})
    return out


HU = "áéíóöőúüűÁÉÍÓÖŐÚÜŰ"
CJK = "漢字仮名交じり文中文測試"
EMOJI = "😂🙂🚀"
ODD = " €—…«»"


def random_byte_vocab(seed, n_merges=300, proper=True, dup_ids=False, neg_ids=False, max_len=12):
    """GPT-2-shaped byte-level vocabulary: 256 byte tokens then `n_merges`
    tokens that are concatenations of two earlier tokens (proper=True) or
    arbitrary short byte strings with shuffled ids (proper=False).
    Returns (entries [(visible-bytes, id)], special mapping)."""
    rng = random.Random(seed)
    t = vf.bytes_to_unicode()
    order = vf.byte_token_order()
    raw_tokens = [bytes([b]) for b in order]
    alphabet = (b"etaoinshrdlucmfwypvbgkqjxz" * 3 + b" " * 6 + b"ETAOIN0123456789.,!?\n\t"
                + "áéőű漢😂".encode("utf-8"))
    seen = set(raw_tokens)
    while len(raw_tokens) < 256 + n_merges:
        if proper:
            a = raw_tokens[rng.randrange(len(raw_tokens))]
            b = rng.choice(raw_tokens)
            if rng.random() < 0.7:
                a = bytes([rng.choice(alphabet)]) if rng.random() < 0.5 else a
            tok = a + b
        else:
            tok = bytes(rng.choice(alphabet) for _ in range(rng.randint(2, 5)))
        if tok in seen or len(tok) > max_len:
            continue
        seen.add(tok)
        raw_tokens.append(tok)
    ids = list(range(len(raw_tokens)))
    if not proper:
        tail = ids[256:]
        rng.shuffle(tail)
        ids[256:] = tail
    if dup_ids:
        for _ in range(n_merges // 10):
            i = rng.randrange(256, len(ids))
            ids[i] = ids[rng.randrange(256, len(ids))]
    if neg_ids:
        for _ in range(n_merges // 20):
            ids[rng.randrange(256, len(ids))] = rng.choice([-1, -2, -7])
    entries = [(vf.encode_visible(tok, t), i) for tok, i in zip(raw_tokens, ids)]
    return entries, vf.gpt2_special_mapping()


def random_char_vocab(seed, n_merges=300, drop_chars="", max_len=10):
    """SentencePiece/Llama-shaped vocabulary (is_byte_encoder=False,
    prefix U+2581): single characters, byte-fallback literals <0xHH>, and
    merges of earlier tokens.  Characters in `drop_chars` are left out so that
    they encode to -1 yet can still appear inside longer tokens."""
    rng = random.Random(seed)
    chars = list("▁etaoinshrdlucmfwypvbgkqjxzETAOIN0123456789.,!?-") + list(HU) + list(CJK[:6])
    toks = ["<0x%02X>" % b for b in range(256)]
    toks += [c for c in chars if c not in drop_chars]
    base = list(chars)
    seen = set(toks)
    while len(toks) < 256 + len(chars) + n_merges:
        a = rng.choice(base if rng.random() < 0.5 else toks[256:])
        b = rng.choice(base if rng.random() < 0.5 else toks[256:])
        tok = a + b
        if tok in seen or len(tok) > max_len:
            continue
        seen.add(tok)
        toks.append(tok)
    entries = [(tk.encode("utf-8"), i) for i, tk in enumerate(toks)]
    return entries, vf.llama_special_mapping()


def random_merges_text(entries, seed, keep=0.85, noise=True):
    """A merges.txt for the id-keyed merge path (reference lib.c:573-663): every split of a vocabulary key
    into two vocabulary keys is a possible rule; a random subset in random order (so rank != id order), plus
    the lines the loader has to cope with: comments, lines without a space, runs of spaces, rules with
    unknown tokens (skipped, take no rank), a repeated pair (the later line and rank win), CRLF endings."""
    rng = random.Random(seed)
    keys = {}
    for k, i in entries:
        keys[k] = i
    rules = []
    for k in keys:
        try:
            txt = k.decode("utf-8")
        except UnicodeDecodeError:
            continue
        for cut in range(1, len(txt)):
            a, b = txt[:cut].encode("utf-8"), txt[cut:].encode("utf-8")
            if a in keys and b in keys and b" " not in k:
                rules.append((a, b))
    rng.shuffle(rules)
    rules = rules[: int(len(rules) * keep)]
    lines = ["#version: 0.2"]
    for a, b in rules:
        sep = " " if not noise or rng.random() < 0.9 else "   "
        end = "" if not noise or rng.random() < 0.9 else "\r"
        lines.append(a.decode("utf-8") + sep + b.decode("utf-8") + end)
        if noise:
            r = rng.random()
            if r < 0.03:
                lines.append("# a comment with a space")
            elif r < 0.06:
                lines.append("nospacehere")
            elif r < 0.09:
                lines.append("zzqx notinvocabq")
            elif r < 0.12 and len(lines) > 5:
                lines.append(rng.choice(lines[1:]))  # an earlier line again
            elif r < 0.13:
                lines.append(a.decode("utf-8") + " ")  # right half missing
    return "\n".join(lines) + "\n"


def write_merges(tmpdir, name, text):
    mp = os.path.join(str(tmpdir), name + "_merges.txt")
    with open(mp, "w", encoding="utf-8", newline="") as f:
        f.write(text)
    return mp


def write_vocab(tmpdir, name, entries, special):
    vp = os.path.join(str(tmpdir), name + "_vocab.txt")
    sp = os.path.join(str(tmpdir), name + "_special.txt")
    vf.write_vocab_file(vp, entries)
    vf.write_special_file(sp, special)
    return vp, sp


def random_text(rng, max_words=12, exotic=0.3):
    """Valid-UTF-8 text that exercises every splitter rule."""
    parts = []
    for _ in range(rng.randint(0, max_words)):
        r = rng.random()
        if r < 0.45:
            w = "".join(rng.choice("etaoinshrdlucmfwypvbgkqjxz") for _ in range(rng.randint(1, 9)))
            if rng.random() < 0.15:
                w = w.capitalize()
        elif r < 0.55:
            w = "".join(rng.choice("0123456789") for _ in range(rng.randint(1, 5)))
        elif r < 0.65:
            w = "".join(rng.choice(".,!?-()\"'") for _ in range(rng.randint(1, 3)))
        elif r < 0.65 + exotic * 0.4:
            w = "".join(rng.choice("aeiou" + HU) for _ in range(rng.randint(1, 7)))
        elif r < 0.65 + exotic * 0.7:
            w = "".join(rng.choice(CJK) for _ in range(rng.randint(1, 5)))
        elif r < 0.65 + exotic * 0.85:
            w = rng.choice(EMOJI) * rng.randint(1, 2)
        else:
            w = rng.choice(ODD)
        sep = rng.choice([" "] * 12 + ["  ", "   ", "\n", "\t", "\r\n", "", "", ", ", ". "])
        parts.append(w + sep)
    s = "".join(parts)
    if rng.random() < 0.2:
        s = " " + s
    return s


def random_bytes_text(rng, n):
    """Arbitrary bytes without 0x00: truncated and invalid UTF-8 included."""
    pool = [b"a", b"b", b" ", b"  ", b"1", b".", b"\t", b"\xc3\xa9", b"\xc5\x91", b"\xe6\xbc\xa2",
            b"\xf0\x9f\x98\x82", b"\xc3", b"\xe6\xbc", b"\xf0\x9f", b"\x80", b"\xbf", b"\xff",
            b"\xc0\xa0", b"\xc1\xa1", b"\xc0\x80", b"\xc2\xa0", b"\xc2\x85", b"\xe0\x80\x80", b"\n"]
    out = b"".join(rng.choice(pool) for _ in range(n))
    return out
