"""G10: special-character replacements of several units (tests/golden/g10_pretokenizer.json, produced by the compiled
reference with tools/make_golden_g10.py: its pretokenizer_encode on the inputs of its own tests/test_pretokenizer.c:23-251,
and its encode() on vocabularies with such replacement files).  Here: the oracle against those vectors, and the product's
loader accepts every one of the files (the GPU side: tests/test_gpu_golden.py::test_g10_replacements_of_several_units)."""
import hashlib
import json
import os
import random

import pytest

import helpers as H
from hutoken_amd import _capi


@pytest.fixture(scope="module")
def g10():
    with open(os.path.join(H.GOLDEN_DIR, "g10_pretokenizer.json")) as f:
        return json.load(f)


def case_files(tmp_path, case):
    entries = [(bytes.fromhex(t), i) for t, i in case["vocab"]]
    return H.write_vocab(tmp_path, case["name"], entries, {int(k): v for k, v in case["replacements"].items()})


def file_case(tmp_path, g):
    """The vocabulary, files and texts of one entry of g10["files"] (same seeded builders as the generator)."""
    seed = g["seed"]
    rng = random.Random(100 + seed)
    if g["kind"] == "byte":
        entries, _ = H.random_byte_vocab(40 + seed, n_merges=300, proper=seed == 0)
    else:
        entries, _ = H.random_char_vocab(40 + seed, n_merges=250)
    special = {int(k): v for k, v in g["special"].items()}
    vp, sp = H.write_vocab(tmp_path, "f%d" % seed, entries, special)
    mp = H.write_merges(tmp_path, "f%d" % seed, H.random_merges_text(entries, seed, noise=False)) if g["merges"] else None
    texts = [H.random_text(rng, 12) + rng.choice(["", " quiz!", "\nq z\n", "\tzq!", " q"]) for _ in range(g["n_texts"])]
    return vp, sp, mp, texts


def sha_ids(lists):
    return hashlib.sha256(b"".join(int(i).to_bytes(4, "little", signed=True) for d in lists for i in d)).hexdigest()


def test_oracle_equals_the_reference(g10, tmp_path, oracle_mod):
    n = 0
    for case in g10["pretokenizer"]:
        if "ids" not in case:
            continue  # (an empty replacement value: the reference's file loader refuses it, lib.c:533-543)
        vp, sp = case_files(tmp_path, case)
        orc = oracle_mod.Oracle(vp, sp, case["prefix"] or None, case["is_byte_encoder"])
        assert orc.encode(case["text"]) == case["ids"], case["name"]
        assert orc.encode("x " + case["text"] + " y " + case["text"]) == case["ids_in_sentence"], case["name"]
        n += 1
    assert n >= 12
    for g in g10["files"]:
        vp, sp, mp, texts = file_case(tmp_path, g)
        orc = oracle_mod.Oracle(vp, sp, g["prefix"], g["kind"] == "byte", merges_path=mp)
        ids = [orc.encode(t) for t in texts]
        assert ids[:len(g["first"])] == g["first"], g["seed"]
        assert sum(len(d) for d in ids) == g["n_ids"] and sha_ids(ids) == g["sha256"], g["seed"]


def test_loader_accepts_them(g10, tmp_path):
    for case in g10["pretokenizer"]:
        if "ids" not in case:
            continue
        vp, sp = case_files(tmp_path, case)
        ctx = _capi.Context(vp, sp, case["prefix"] or None, case["is_byte_encoder"], device=-2)
        most = max([len(v) for v in case["replacements"].values()] + [1])
        assert ctx.ids_capacity(10, 0) >= 10 * min(most, 2) if most > 1 else True
        ctx.close()
    for g in g10["files"]:
        vp, sp, mp, _ = file_case(tmp_path, g)
        ctx = _capi.Context(vp, sp, g["prefix"], g["kind"] == "byte", device=-2, merges_path=mp)
        assert ctx.ids_capacity(100, 0) >= 200  # a replacement of several units: more ids than bytes are possible
        ctx.close()
