"""Frozen synthetic corpora (csrc/hutk_synth.c) as numpy arrays."""
import ctypes as C
import os

import numpy as np

from . import build as _build

# seeds frozen by SURVEY.md section 8 d
SEED_C2 = 0x48554732
SEED_C3 = 0x48554733
SEED_C5 = 0x48554735
KINDS = {"C2": (2, SEED_C2, 100_000), "C3": (3, SEED_C3, 1_000_000), "C5": (5, SEED_C5, 1_000_000)}

_lib = None


def _load():
    global _lib
    if _lib is None:
        path = _build.build_synth()
        L = C.CDLL(path)
        L.hutk_synth_corpus.restype = C.c_int64
        L.hutk_synth_corpus.argtypes = [C.c_int, C.c_uint64, C.c_int64, C.c_int64, C.c_int,
                                        C.POINTER(C.POINTER(C.c_uint8)), C.c_void_p]
        L.hutk_synth_free.argtypes = [C.c_void_p]
        L.hutk_synth_lengths.restype = C.c_int
        L.hutk_synth_lengths.argtypes = [C.c_int, C.c_uint64, C.c_int64, C.c_int64, C.c_int, C.c_void_p]
        _lib = L
    return _lib


def corpus(name, n_docs=None, first_doc=0, seed=None, threads=None):
    """Documents [first_doc, first_doc+n_docs) of corpus "C2" | "C3" | "C5".
    -> (uint8 array of packed UTF-8, int64 offsets[n_docs+1])"""
    kind, dseed, dn = KINDS[name]
    n_docs = dn if n_docs is None else int(n_docs)
    seed = dseed if seed is None else seed
    threads = threads or min(16, os.cpu_count() or 1)
    L = _load()
    offs = np.zeros(n_docs + 1, dtype=np.int64)
    p = C.POINTER(C.c_uint8)()
    total = L.hutk_synth_corpus(kind, seed, first_doc, n_docs, threads, C.byref(p),
                                offs.ctypes.data)
    if total < 0:
        raise MemoryError("synthetic corpus generation failed")
    data = np.ctypeslib.as_array(p, shape=(max(total, 1),))[:total].copy()
    L.hutk_synth_free(p)
    return data, offs


def lengths(name, n_docs=None, first_doc=0, seed=None, threads=None):
    """Byte lengths (int64[n_docs]) of documents [first_doc, first_doc+n_docs) without keeping their bytes."""
    kind, dseed, dn = KINDS[name]
    n_docs = dn if n_docs is None else int(n_docs)
    out = np.zeros(max(n_docs, 1), dtype=np.int64)
    rc = _load().hutk_synth_lengths(kind, dseed if seed is None else seed, first_doc, n_docs,
                                    threads or min(16, os.cpu_count() or 1), out.ctypes.data)
    if rc:
        raise ValueError("unknown corpus")
    return out[:n_docs]


def docs_as_str(data, offs):
    b = data.tobytes()
    return [b[offs[i]:offs[i + 1]].decode("utf-8") for i in range(len(offs) - 1)]


def random_words(lo, hi, n_docs, words_per_doc=20, seed=0x52574f52, lexicon=20000, alphabet=b"etaoinshrdlucmfw"):
    """Texts OFF the vocabulary's distribution: documents of `words_per_doc` words drawn uniformly from a lexicon of
    random-letter words of lo..hi letters, each followed by one space (hardly any is a vocabulary key, so every
    word goes through the merge loop; beyond 32 units a word leaves the tile kernel for the exception kernels).
    -> (uint8 array, int64 offsets[n_docs+1]); numpy's PCG64 with a fixed seed, identical on every host."""
    rng = np.random.default_rng(seed + 1000 * lo + hi)
    alpha = np.frombuffer(alphabet, dtype=np.uint8)
    lens = rng.integers(lo, hi + 1, lexicon)
    W = alpha[rng.integers(0, len(alpha), (lexicon, hi + 1))]
    W[np.arange(lexicon), lens] = 0x20  # the separator right after the word
    idx = rng.integers(0, lexicon, (n_docs, words_per_doc))
    wl = (lens + 1)[idx]
    mask = np.arange(hi + 1)[None, None, :] < wl[:, :, None]
    data = W[idx][mask]
    offs = np.zeros(n_docs + 1, dtype=np.int64)
    np.cumsum(wl.sum(axis=1), out=offs[1:])
    return np.ascontiguousarray(data), offs


def cjk_paragraphs(n_docs, seed=0x434a4b50, lo=100, hi=400, alphabet=3000):
    """Texts that the reference's splitter takes as FEW, LONG words: paragraphs of lo..hi CJK characters (uniform over
    `alphabet` code points from U+4E00, a full stop or a comma now and then -- all of one splitter class, so a whole
    paragraph is one word of 300..1200 bytes, parser.c:102-138), one to four per document with a line feed between.
    -> (uint8 array, int64 offsets[n_docs+1]); numpy's PCG64 with a fixed seed, identical on every host."""
    rng = np.random.default_rng(seed)
    n_par = rng.integers(1, 5, n_docs)
    plen = rng.integers(lo, hi + 1, int(n_par.sum()))
    total_chars = int(plen.sum())
    cp = 0x4E00 + rng.integers(0, alphabet, total_chars)
    punct = rng.random(total_chars)
    cp = np.where(punct < 0.03, 0x3002, np.where(punct < 0.08, 0xFF0C, cp)).astype(np.uint32)
    b = np.empty((total_chars, 3), dtype=np.uint8)
    b[:, 0] = 0xE0 | (cp >> 12)
    b[:, 1] = 0x80 | ((cp >> 6) & 0x3F)
    b[:, 2] = 0x80 | (cp & 0x3F)
    # a line feed behind every paragraph but a document's last
    par_end = np.cumsum(plen)                      # in characters
    doc_last_par = np.cumsum(n_par) - 1
    is_last = np.zeros(len(plen), dtype=bool)
    is_last[doc_last_par] = True
    par_bytes = plen * 3 + (~is_last)
    out = np.empty(int(par_bytes.sum()), dtype=np.uint8)
    pos = np.concatenate(([0], np.cumsum(par_bytes)))
    flat = b.reshape(-1)
    cstart = np.concatenate(([0], par_end[:-1]))
    for k in range(len(plen)):  # (a few thousand paragraphs per 1000 documents: fine)
        out[pos[k]:pos[k] + 3 * plen[k]] = flat[3 * cstart[k]:3 * par_end[k]]
        if not is_last[k]:
            out[pos[k] + 3 * plen[k]] = 0x0A
    doc_bytes = np.add.reduceat(par_bytes, np.concatenate(([0], np.cumsum(n_par)[:-1])))
    offs = np.zeros(n_docs + 1, dtype=np.int64)
    np.cumsum(doc_bytes, out=offs[1:])
    return out, offs
