#!/usr/bin/env python3
"""Regenerates the frozen synthetic vocabularies under data/ (run once; outputs
are committed).  Test/bench data infrastructure, not the product path.

  VG  GPT-2 shape, 50257 lines: 256 byte tokens in GPT-2 id order, 50000 merges
      learned by tools/train_vocab.cpp on 125000 documents (64 MB) of corpus C3
      (generator seed 0x564f4347), then <|endoftext|>.  ids = merge order.
      is_byte_encoder=True, no prefix.  Special file = the 68 remapped bytes.
      vg50257_merges.txt: the 50000 rules in merges.txt form (id-keyed merge path).
  VL  SentencePiece/Llama shape, 32000 lines: <unk>, <s>, </s>, 256 byte-fallback
      literals <0xHH>, the base characters, then merges learned in "chars" mode
      on 60000 documents of corpus C5 (seed 0x564f434c).
      is_byte_encoder=False, prefix U+2581.
"""
import gzip
import hashlib
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hutoken_amd import vocab_files as vf  # noqa: E402

DATA = os.path.join(ROOT, "data")


def build_trainer(tmp):
    exe = os.path.join(tmp, "train_vocab")
    subprocess.check_call(["gcc", "-O2", "-c", os.path.join(ROOT, "oracle/hutk_oracle.c"),
                           "-o", os.path.join(tmp, "o.o")])
    subprocess.check_call(["gcc", "-O2", "-c", os.path.join(ROOT, "hutoken_amd/csrc/hutk_synth.c"),
                           "-o", os.path.join(tmp, "s.o")])
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-o", exe,
                           os.path.join(ROOT, "tools/train_vocab.cpp"),
                           os.path.join(tmp, "o.o"), os.path.join(tmp, "s.o"), "-lpthread"])
    return exe


def write_gz(path, text):
    with open(path, "wb") as raw:
        with gzip.GzipFile(fileobj=raw, mode="wb", mtime=0, filename="") as f:
            f.write(text.encode("utf-8"))
    return hashlib.sha256(text.encode("utf-8")).hexdigest()


def main():
    os.makedirs(DATA, exist_ok=True)
    tmp = tempfile.mkdtemp()
    exe = build_trainer(tmp)
    sums = {}

    # ---- VG ----
    mg = os.path.join(tmp, "vg.txt")
    mg_pairs = os.path.join(tmp, "vg_pairs.txt")
    subprocess.check_call([exe, "3", "0x564f4347", "125000", "50000", mg, "bytes", mg_pairs])
    t = vf.bytes_to_unicode()
    lines = []
    idx = 0
    for b in vf.byte_token_order():
        lines.append(vf.hex_line(vf.encode_visible(bytes([b]), t), idx))
        idx += 1
    for ln in open(mg):
        raw = bytes.fromhex(ln.strip())
        lines.append(vf.hex_line(vf.encode_visible(raw, t), idx))
        idx += 1
    lines.append(vf.hex_line(b"<|endoftext|>", idx))
    assert idx + 1 == 50257
    sums["vg50257_vocab.txt"] = write_gz(os.path.join(DATA, "vg50257_vocab.txt.gz"), "".join(lines))
    sp = "".join("%d == %s\n" % (b, s) for b, s in sorted(vf.gpt2_special_mapping().items()))
    with open(os.path.join(DATA, "vg50257_special.txt"), "w", encoding="utf-8") as f:
        f.write(sp)
    sums["vg50257_special.txt"] = hashlib.sha256(sp.encode("utf-8")).hexdigest()
    # the merge rules in GPT-2's merges.txt format (visible characters, one "left right" per line, rank =
    # line order): input of the id-keyed merge path (reference core.c:211-337, lib.c:573-663)
    mtext = "#version: 0.2\n"
    for ln in open(mg_pairs):
        l, r = ln.split()
        mtext += (vf.encode_visible(bytes.fromhex(l), t).decode("utf-8") + " " +
                  vf.encode_visible(bytes.fromhex(r), t).decode("utf-8") + "\n")
    sums["vg50257_merges.txt"] = write_gz(os.path.join(DATA, "vg50257_merges.txt.gz"), mtext)

    # ---- VL ----
    ml = os.path.join(tmp, "vl.txt")
    # first pass to learn the base character count
    subprocess.check_call([exe, "5", "0x564f434c", "60000", "1", ml, "chars"])
    nbase = open(ml).read().split("--\n")[0].count("\n")
    n_merges = 32000 - 3 - 256 - nbase
    subprocess.check_call([exe, "5", "0x564f434c", "60000", str(n_merges), ml, "chars"])
    base, merges = open(ml).read().split("--\n")
    toks = [b"<unk>", b"<s>", b"</s>"] + [b"<0x%02X>" % b for b in range(256)]
    toks += [bytes.fromhex(x) for x in base.split()] + [bytes.fromhex(x) for x in merges.split()]
    assert len(toks) == 32000 and len(set(toks)) == 32000
    text = "".join(vf.hex_line(tk, i) for i, tk in enumerate(toks))
    sums["vl32000_vocab.txt"] = write_gz(os.path.join(DATA, "vl32000_vocab.txt.gz"), text)
    sp = "".join("%d == %s\n" % (b, s) for b, s in sorted(vf.llama_special_mapping().items()))
    with open(os.path.join(DATA, "vl32000_special.txt"), "w", encoding="utf-8") as f:
        f.write(sp)
    sums["vl32000_special.txt"] = hashlib.sha256(sp.encode("utf-8")).hexdigest()

    with open(os.path.join(DATA, "SHA256SUMS"), "w") as f:
        for k in sorted(sums):
            f.write("%s  %s\n" % (sums[k], k))
    print(open(os.path.join(DATA, "SHA256SUMS")).read())


if __name__ == "__main__":
    main()
