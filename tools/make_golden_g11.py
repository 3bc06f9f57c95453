#!/usr/bin/env python3
"""tests/golden/g11_cjk_dense.json, by THE REFERENCE ITSELF (oracle/_ref): the CJK-dense vocabulary VC (data/vc12257_*,
tools/make_vocab_cjk.py: merges across neighbouring characters, a saturated seam map) x
  - 300 documents of synth.cjk_text (the structure the vocabulary was trained on; other seed), and
  - 200 documents of synth.cjk_paragraphs (uniformly random characters),
and VG x the same cjk_text documents (seams that DO cut).  -> ids of the first documents, id count, sha256 of all ids.
Data only; run in the build container."""
import hashlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
from hutoken_amd import data, synth  # noqa: E402
from oracle import ref  # noqa: E402


def sha_ids(list_of_lists):
    h = hashlib.sha256()
    for ids in list_of_lists:
        h.update(np.asarray(ids, dtype="<i4").tobytes())
        h.update(b"|")
    return h.hexdigest()


def main():
    ref.build()
    out = []
    for vocab, gen, n_docs in [("VC", "cjk_text", 300), ("VC", "cjk_paragraphs", 200), ("VG", "cjk_text", 300)]:
        vp, sp, kw = data.vocab_files(vocab)
        R = ref.RefTokenizer(vp, sp, kw["prefix"], kw["is_byte_encoder"])
        d, o = getattr(synth, gen)(n_docs)
        texts = [bytes(d[o[i]:o[i + 1]]).decode("utf-8") for i in range(n_docs)]
        res = [list(map(int, x)) for x in R.batch_encode(texts, 8)]
        out.append(dict(vocab=vocab, generator=gen, n_docs=n_docs, corpus_sha256=hashlib.sha256(d.tobytes()).hexdigest(),
                        first=res[:6], n_ids=sum(len(x) for x in res), sha256=sha_ids(res)))
        print(vocab, gen, n_docs, "docs", len(d), "bytes ->", out[-1]["n_ids"], "ids")
    with open(os.path.join(ROOT, "tests", "golden", "g11_cjk_dense.json"), "w") as f:
        json.dump(out, f)


if __name__ == "__main__":
    main()
