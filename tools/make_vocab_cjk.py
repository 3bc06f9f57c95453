#!/usr/bin/env python3
"""Regenerates data/vc12257_vocab.txt.gz: a GPT-2-SHAPED byte-level vocabulary trained on CJK-dense text
(synth.cjk_text: words of one to three characters from a 3000-character alphabet, no spaces), VERDICT r03 item 6.

  VC  256 byte tokens in GPT-2 id order, 12000 merges learned by tools/train_vocab.cpp on 8000 documents of
      synth.cjk_text(seed 0x56435452) -- under the reference's splitter every paragraph is one word, so the merges run
      across character boundaries wherever a pair of neighbouring characters is frequent -- then <|endoftext|>: 12257 lines,
      ids = merge order, is_byte_encoder=True, no prefix, special file = VG's (the 68 remapped bytes).
What it is for: the seam map of hutk_loader.cpp cuts a paragraph of CJK characters only where no merge can join the byte
in front to the lead byte behind; this vocabulary has such merges for (nearly) every pair, so nothing is cut and each
paragraph stays the 300..1200-byte word the reference makes of it.  Test / bench data infrastructure, not the product path."""
import gzip, hashlib, os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from hutoken_amd import synth, vocab_files as vf  # noqa: E402
import make_vocab as MV  # noqa: E402

DATA = os.path.join(ROOT, "data")
N_MERGES = 12000


def main():
    tmp = tempfile.mkdtemp()
    exe = MV.build_trainer(tmp)
    d, o = synth.cjk_text(8000, seed=0x56435452)
    txt = os.path.join(tmp, "cjk.txt")
    with open(txt, "wb") as f:
        raw = d.tobytes()
        for i in range(len(o) - 1):
            for par in raw[o[i]:o[i + 1]].split(b"\n"):
                f.write(par + b"\n")
    out = os.path.join(tmp, "vc.txt")
    subprocess.check_call([exe, "0", txt, "0", str(N_MERGES), out, "bytes"])
    t = vf.bytes_to_unicode()
    lines, idx = [], 0
    for b in vf.byte_token_order():
        lines.append(vf.hex_line(vf.encode_visible(bytes([b]), t), idx)); idx += 1
    for ln in open(out):
        lines.append(vf.hex_line(vf.encode_visible(bytes.fromhex(ln.strip()), t), idx)); idx += 1
    lines.append(vf.hex_line(b"<|endoftext|>", idx))
    assert idx + 1 == 256 + N_MERGES + 1
    name = "vc%d_vocab.txt" % (idx + 1)
    h = MV.write_gz(os.path.join(DATA, name + ".gz"), "".join(lines))
    sums = {}
    for ln in open(os.path.join(DATA, "SHA256SUMS")):
        hh, nm = ln.split()
        sums[nm] = hh
    sums[name] = h
    with open(os.path.join(DATA, "SHA256SUMS"), "w") as f:
        for k in sorted(sums):
            f.write("%s  %s\n" % (sums[k], k))
    print(name, h)


if __name__ == "__main__":
    main()
