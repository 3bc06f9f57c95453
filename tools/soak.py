#!/usr/bin/env python3
"""Randomised parity soak on the GPU: many small random vocabularies (byte-level and character-level, proper and shuffled
ids, with and without a merges file) x random texts, long words, token concatenations; every id against the oracle.
usage: soak.py [SECONDS=120] [FIRST_SEED=1000]"""
import os, random, sys, tempfile, time
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np
import helpers as H
from hutoken_amd import _capi
from oracle import oracle as O

O.build()
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
t_end = time.time() + budget
t_note = time.time() + 60
tmp = tempfile.mkdtemp()
n_vocab = n_docs = n_ids = 0
while time.time() < t_end:
    rng = random.Random(seed)
    is_byte = rng.random() < 0.6
    nm = rng.choice([40, 300, 1500, 4000, 9000])
    if is_byte:
        ents, sp = H.random_byte_vocab(seed, n_merges=nm, proper=rng.random() < 0.7, dup_ids=rng.random() < 0.2,
                                       max_len=rng.choice([8, 12, 16, 24, 31]))
        prefix = None
    else:
        ents, sp = H.random_char_vocab(seed, n_merges=nm, drop_chars=rng.choice(["", "qz", "ő漢"]), max_len=rng.choice([8, 12, 16, 24]))
        prefix = "▁"
    vp, spath = H.write_vocab(tmp, "s%d" % seed, ents, sp)
    mp = None
    if is_byte and rng.random() < 0.3:
        mp = os.path.join(tmp, "m%d.txt" % seed)
        open(mp, "w", encoding="utf-8").write(H.random_merges_text(ents, seed))
    try:
        ctx = _capi.Context(vp, spath, prefix, is_byte, merges_path=mp)
        orc = O.Oracle(vp, spath, prefix, is_byte, mp)
    except Exception as e:  # both must refuse the same files; the loaders' own tests cover that
        seed += 1
        continue
    docs = []
    toks = []  # the vocabulary's keys as raw input (words of their own: the whole-word tables, up to 28 bytes)
    if is_byte:
        from hutoken_amd import vocab_files as vf
        back = {c: b for b, c in vf.bytes_to_unicode().items()}
        for k, _ in ents:
            try:
                t = bytes(back[c] for c in k.decode("utf-8")).strip(b" \n\t\0")
            except (KeyError, UnicodeDecodeError):
                continue
            if t and b"\0" not in t:
                toks.append(t)
    else:
        toks = [k.decode("utf-8").replace("\u2581", " ").strip().encode("utf-8") for k, _ in ents if not k.startswith(b"<0x")]
        toks = [t for t in toks if t]
    for _ in range(rng.randint(200, 1500)):
        r = rng.random()
        if r < 0.15 and toks:
            docs.append(b" ".join(rng.choice(toks) for _ in range(rng.randint(1, 30))))
        elif r < 0.6:
            docs.append(H.random_text(rng, max_words=rng.choice([3, 12, 60])).encode("utf-8"))
        elif r < 0.8:
            w = bytes(rng.choice(b"etaoinshrdlu ") for _ in range(rng.randint(1, 400)))
            docs.append(w)
        elif r < 0.9:
            docs.append(b"")
        else:
            docs.append(" ".join("".join(rng.choice("aeiouáéő漢xyz") for _ in range(rng.randint(1, 90))) for _ in range(rng.randint(1, 6))).encode("utf-8"))
    docs = [d.replace(b"\0", b"") for d in docs]
    if not is_byte:  # the character mode refuses invalid UTF-8: keep to valid text
        docs = [d.decode("utf-8", "ignore").encode("utf-8") for d in docs]
    data, offs = O.pack(docs)
    ids_o, oo_o, st_o = orc.encode_packed(data, offs, 8)
    ids_g, oo_g, st_g, rc = ctx.encode_packed(data, offs)
    ok = rc == 0 and np.array_equal(oo_o, oo_g) and np.array_equal(ids_o, ids_g)
    if not ok:
        k = int(np.nonzero(oo_o != oo_g)[0][0]) if not np.array_equal(oo_o, oo_g) else -1
        print(f"MISMATCH seed {seed} byte={is_byte} merges={mp is not None} rc={rc} first bad offset index {k}", flush=True)
        sys.exit(1)
    n_vocab += 1; n_docs += len(docs); n_ids += len(ids_o)
    if time.time() > t_note:  # (a line a minute: a silent run is taken for a hung one)
        print(f"... {n_vocab} vocabularies, {n_ids} ids, seed {seed}", flush=True)
        t_note = time.time() + 60
    ctx.close(); orc.close()
    seed += 1
print(f"soak OK: {n_vocab} vocabularies, {n_docs} documents, {n_ids} ids, seeds up to {seed - 1}")
