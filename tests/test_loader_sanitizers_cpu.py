"""The host half of the product (hutk_loader.cpp) under AddressSanitizer and UBSan on the CPU: the shipped vocabularies,
random ones of both shapes, merges files with noise, and the reference's loader quirk files (truncated lines, repeated
keys, NUL bytes, out-of-range indices, ...).  GPU sanitizers are not available on the pool; the device code is covered by
the parity tests."""
import os
import subprocess
import sys

import pytest

import helpers as H
from hutoken_amd import data

sys.path.insert(0, os.path.join(H.ROOT, "tools"))
import make_golden_g8 as G8  # noqa: E402  (only its seeded file builders)


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    out = os.path.join(str(tmp_path_factory.mktemp("san")), "loader_sanitize")
    csrc = os.path.join(H.ROOT, "hutoken_amd", "csrc")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                           "-I" + csrc, "-I" + os.path.join(H.ROOT, "include"), "-o", out,
                           os.path.join(H.ROOT, "tests", "cpu", "loader_sanitize.cpp"), os.path.join(csrc, "hutk_loader.cpp")])
    return out


def test_loader_is_clean_under_asan_and_ubsan(exe, tmp_path):
    cases = []
    for name in ("VG", "VL"):
        vp, sp, kw = data.vocab_files(name)
        cases.append((vp, sp, kw["prefix"] or "-", "1" if kw["is_byte_encoder"] else "0", "-"))
    vp, sp, kw = data.vocab_files("VG")
    cases.append((vp, sp, "-", "1", data.merges_file("VG")))
    for seed in range(6):
        ents, spm = H.random_byte_vocab(seed, n_merges=[30, 400, 3000][seed % 3], proper=seed % 2 == 0, dup_ids=seed == 3,
                                        neg_ids=seed == 4, max_len=16)
        v, s = H.write_vocab(tmp_path, "b%d" % seed, ents, spm)
        mp = "-"
        if seed % 2:
            mp = os.path.join(str(tmp_path), "m%d.txt" % seed)
            with open(mp, "w", encoding="utf-8") as f:
                f.write(H.random_merges_text(ents, seed))
        cases.append((v, s, "-", "1", mp))
        ents, spm = H.random_char_vocab(seed, n_merges=500, drop_chars="qző" if seed % 2 else "")
        v, s = H.write_vocab(tmp_path, "c%d" % seed, ents, spm)
        cases.append((v, s, "▁", "0", "-"))
    quirks = G8.quirk_files()
    for name, (vocab, special, _probes) in quirks.items():
        v, s = G8.write_case_files(str(tmp_path), name, vocab, special)
        cases.append((v, s, "-", "1", "-"))
    cases.append((os.path.join(str(tmp_path), "absent_vocab.txt"), cases[0][1], "-", "1", "-"))
    feed = "".join("\t".join(c) + "\n" for c in cases)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([exe], input=feed, capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "Sanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-3000:]
    lines = r.stdout.splitlines()
    assert len(lines) == len(cases)
    assert lines[0].startswith("ok 50257 symbols") and all(x.startswith("ok") for x in lines[:3])
    assert lines[-1].startswith("error")
    assert sum(x.startswith("error") for x in lines) >= 5  # the quirk files the loader refuses
