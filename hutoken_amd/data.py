"""Frozen synthetic vocabularies shipped under data/ (see tools/make_vocab.py).

  VG  GPT-2 shape, 50257 entries, is_byte_encoder=True, no prefix
  VL  SentencePiece/Llama shape, 32000 entries, is_byte_encoder=False, prefix U+2581
  VC  GPT-2 shape, 12257 entries, trained on CJK-dense text (tools/make_vocab_cjk.py): merges across every frequent pair
      of neighbouring characters, i.e. a saturated seam map; is_byte_encoder=True, no prefix
"""
import gzip
import hashlib
import os
import tempfile

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_DATA = os.path.join(_ROOT, "data")

VOCABS = {
    "VG": dict(vocab="vg50257_vocab.txt", special="vg50257_special.txt", merges="vg50257_merges.txt",
               kwargs=dict(prefix=None, is_byte_encoder=True)),
    "VC": dict(vocab="vc12257_vocab.txt", special="vg50257_special.txt",
               kwargs=dict(prefix=None, is_byte_encoder=True)),
    "VL": dict(vocab="vl32000_vocab.txt", special="vl32000_special.txt",
               kwargs=dict(prefix="▁", is_byte_encoder=False)),
}


def _sums():
    out = {}
    with open(os.path.join(_DATA, "SHA256SUMS")) as f:
        for line in f:
            h, name = line.split()
            out[name] = h
    return out


def _unpacked(fname, cache_dir=None):
    """data/<fname>.gz unpacked into a per-user cache directory, checked against data/SHA256SUMS."""
    cache_dir = cache_dir or os.path.join(tempfile.gettempdir(), "hutoken_amd_data_%d" % os.getuid())
    os.makedirs(cache_dir, exist_ok=True)
    vpath = os.path.join(cache_dir, fname)
    want = _sums()[fname]
    ok = False
    if os.path.exists(vpath):
        with open(vpath, "rb") as f:
            ok = hashlib.sha256(f.read()).hexdigest() == want
    if not ok:
        with gzip.open(os.path.join(_DATA, fname + ".gz"), "rb") as f:
            raw = f.read()
        if hashlib.sha256(raw).hexdigest() != want:
            raise RuntimeError("data/%s.gz does not match data/SHA256SUMS" % fname)
        tmp = vpath + ".tmp%d" % os.getpid()
        with open(tmp, "wb") as f:
            f.write(raw)
        os.replace(tmp, vpath)
    return vpath


def vocab_files(name, cache_dir=None):
    """Unpack vocabulary `name` -> (vocab_path, special_path, initialize kwargs)."""
    spec = VOCABS[name]
    return _unpacked(spec["vocab"], cache_dir), os.path.join(_DATA, spec["special"]), dict(spec["kwargs"])


def merges_file(name, cache_dir=None):
    """Unpack the merges.txt of vocabulary `name` (id-keyed merge path) -> path."""
    return _unpacked(VOCABS[name]["merges"], cache_dir)
