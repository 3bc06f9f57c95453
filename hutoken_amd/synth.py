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


def cjk_text(n_docs, seed=0x434a4b54, lexicon=6000, alphabet=3000, lo=100, hi=400):
    """CJK text WITH the structure BPE feeds on: paragraphs of lo..hi characters made of words of one to three characters
    drawn (Zipf, s = 1.0) from a lexicon of `lexicon` words over `alphabet` code points from U+4E00 (the characters
    themselves Zipf-distributed), written without spaces, a full stop or a comma behind one word in twelve; one to four
    paragraphs per document with a line feed between.  Under the reference's splitter a paragraph is ONE word
    (parser.c:102-138); a vocabulary trained on such text (data/vc12257_*, tools/make_vocab_cjk.py) has merges across every
    frequent pair of neighbouring characters, so its seam map is saturated: nothing cuts the paragraphs.
    -> (uint8 array, int64 offsets[n_docs+1]); numpy's PCG64 with fixed seeds, identical on every host."""
    lex_rng = np.random.default_rng(0x434a4b4c)  # the lexicon is the same for every seed
    pc = 1.0 / np.arange(1, alphabet + 1)
    pc /= pc.sum()
    wlen = lex_rng.choice([1, 2, 3], size=lexicon, p=[0.25, 0.55, 0.20])
    wchars = lex_rng.choice(alphabet, size=int(wlen.sum()), p=pc).astype(np.uint32) + 0x4E00
    wstart = np.concatenate(([0], np.cumsum(wlen)))
    pw = 1.0 / np.arange(1, lexicon + 1)
    pw /= pw.sum()
    rng = np.random.default_rng(seed)
    n_par = rng.integers(1, 5, n_docs)
    plen = rng.integers(lo, hi + 1, int(n_par.sum()))  # characters per paragraph (the last word may overshoot by two)
    out = []
    offsets = np.zeros(n_docs + 1, dtype=np.int64)
    par = 0
    total = 0
    # words for all paragraphs at once, cut where the running length reaches each paragraph's target
    need = int(plen.sum())
    words = rng.choice(lexicon, size=need, p=pw)  # (more than enough: every word has at least one character)
    punct = rng.random(need)
    wi = 0
    for d in range(n_docs):
        parts = []
        for k in range(int(n_par[d])):
            target = int(plen[par]); par += 1
            got = 0
            cps = []
            while got < target:
                w = int(words[wi])
                cps.append(wchars[wstart[w]:wstart[w + 1]])
                got += int(wlen[w])
                if punct[wi] < 1.0 / 12.0:
                    cps.append(np.array([0x3002 if punct[wi] < 1.0 / 30.0 else 0xFF0C], dtype=np.uint32))
                    got += 1
                wi += 1
            cp = np.concatenate(cps)
            b = np.empty((len(cp), 3), dtype=np.uint8)
            b[:, 0] = 0xE0 | (cp >> 12)
            b[:, 1] = 0x80 | ((cp >> 6) & 0x3F)
            b[:, 2] = 0x80 | (cp & 0x3F)
            parts.append(b.reshape(-1))
            if k + 1 < int(n_par[d]):
                parts.append(np.array([0x0A], dtype=np.uint8))
        doc = np.concatenate(parts)
        out.append(doc)
        total += len(doc)
        offsets[d + 1] = total
    return np.concatenate(out), offsets


def big_document(n_bytes, name="C3"):
    """ONE document of at least n_bytes: the first documents of corpus `name` back to back (the reference's own benchmark
    times one file of 1 MB .. 1 GB, scripts/benchmark.py:51-104).  -> (uint8 array, offsets [0, len])"""
    lens = lengths(name, max(1, int(n_bytes // 100)))  # (documents are at least 16 bytes, ~250..500 on average: more than enough)
    k = int(np.searchsorted(np.cumsum(lens), n_bytes)) + 1
    d, o = corpus(name, min(k, len(lens)))
    return d, np.array([0, len(d)], dtype=np.int64)


def whitespace_chunks(data, n_chunks):
    """Offsets that cut `data` into about n_chunks pieces at whitespace, as the reference's benchmark does before
    batch_encode (scripts/benchmark.py:26-48: a cut moves right to the next ' ', '\n' or '\t', which begins the next
    piece) -- with one more condition, so that the pieces' ids, concatenated, ARE the whole document's ids: the byte in front
    of the cut is no whitespace (a cut inside a run of spaces changes which word the last space goes with)."""
    n = len(data)
    cuts = [0]
    size = max(1, n // n_chunks)
    ws = (data == 0x20) | (data == 0x0A) | (data == 0x09)
    for k in range(1, n_chunks):
        e = max(k * size, cuts[-1] + 1)
        while e < n and not (ws[e] and not ws[e - 1]):
            e += 1
        if e >= n:
            break
        cuts.append(e)
    cuts.append(n)
    return np.array(cuts, dtype=np.int64)
