"""tests/golden/g8_reference_fixtures.json -- data produced by the compiled reference (tools/make_golden_g8.py) --
against the oracle, the product's loader (host-only context) and the host form of the device splitter.  No GPU."""
import json
import os
import subprocess
import sys

import pytest

import helpers as H
from hutoken_amd import _capi, vocab_files as vf

sys.path.insert(0, os.path.join(H.ROOT, "tools"))
import make_golden_g8 as G8  # noqa: E402  (only its seeded file builders: the reference is not touched here)

# file shapes the device tables refuse at load although the reference accepts them
REFUSED_BY_DESIGN = {
    "special_last_line_without_newline": ValueError,  # the dropped last byte leaves half a character as byte 173's replacement
}


@pytest.fixture(scope="module")
def g8():
    with open(os.path.join(H.GOLDEN_DIR, "g8_reference_fixtures.json")) as f:
        return json.load(f)


def outcome(make, probes=None, encode=None):
    try:
        obj = make()
    except Exception as e:  # noqa: BLE001
        return {"error": [type(e).__name__, str(e)]}
    return {"ids": [encode(obj, x) for x in probes]} if encode else {"ok": True}


def test_loader_quirks_oracle_and_product_equal_the_reference(g8, tmp_path, oracle_mod):
    files = G8.quirk_files()
    assert set(files) == {c["name"] for c in g8["loaders"]}
    for case in g8["loaders"]:
        vocab, special, probes = files[case["name"]]
        vp, sp = G8.write_case_files(str(tmp_path), case["name"], vocab, special)
        want = {"error": case["error"]} if "error" in case else {"ids": case["ids"]}
        got_o = outcome(lambda: oracle_mod.Oracle(vp, sp, None, True), probes, lambda o, x: o.encode(x))
        assert got_o == want, case["name"]
        got_p = outcome(lambda: _capi.Context(vp, sp, None, True, device=-2))
        if case["name"] in REFUSED_BY_DESIGN:
            assert got_p["error"][0] == REFUSED_BY_DESIGN[case["name"]].__name__, case["name"]
        elif "error" in want:
            assert got_p == want, case["name"]
        else:
            assert got_p == {"ok": True}, case["name"]


def test_word_boundaries_of_the_reference_parser(g8, tmp_path, oracle_mod):
    """The 28 strings of the reference's tests/test_parser.c (and more): the reference's own parser_next_token against
    the oracle's splitter and the three host forms of the device splitter (hutk_classify.h)."""
    exe = os.path.join(str(tmp_path), "classify_check")
    obj = os.path.join(str(tmp_path), "oracle.o")
    subprocess.check_call(["gcc", "-O2", "-c", os.path.join(H.ROOT, "oracle", "hutk_oracle.c"), "-o", obj])
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I" + os.path.join(H.ROOT, "hutoken_amd", "csrc"),
                           "-o", exe, os.path.join(H.ROOT, "tests", "cpu", "classify_check.cpp"), obj, "-lpthread"])
    cases = g8["parser"]
    assert len(cases) >= 28
    for c in cases:
        assert oracle_mod.split_words(c["text"].encode("utf-8")) == c["word_starts"], repr(c["text"])
    feed = "".join(c["text"].encode("utf-8").hex() + "\n" for c in cases)
    out = subprocess.run([exe, "--texts"], input=feed, capture_output=True, text=True, check=True).stdout.splitlines()
    assert len(out) == len(cases)
    for c, line in zip(cases, out):
        for form in line.split(";"):
            got = [int(x) for x in form.split(",")] if form else []
            assert got == c["word_starts"], repr(c["text"])


def test_pop_order_of_the_reference_queue(g8):
    """queue.c:152-199: candidates leave the reference's heap in (rank, left_idx) order -- the rule the merge loops
    here restate as "smallest key rank << 5 | position" (tests/test_queue.c:146-178 is the first case)."""
    for c in g8["queue"]:
        assert c["pops"] == [list(x) for x in sorted(tuple(p) for p in c["pushes"])]


def test_repeated_pairs_oracle(g8, tmp_path, oracle_mod):
    with open(os.path.join(H.GOLDEN_DIR, "g1_handpicked.json")) as f:
        g1 = json.load(f)
    t = vf.bytes_to_unicode()
    raw = [bytes([b]) for b in vf.byte_token_order()] + [bytes.fromhex(m) for m in g1["byte_vocab"]["merges_hex"]]
    vp, sp = H.write_vocab(tmp_path, "g8t", [(vf.encode_visible(tok, t), i) for i, tok in enumerate(raw)],
                           vf.gpt2_special_mapping())
    orc = oracle_mod.Oracle(vp, sp, None, True)
    for c in g8["ties"]:
        assert orc.encode(c["text"]) == c["ids"], repr(c["text"])
