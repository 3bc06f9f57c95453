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


def docs_as_str(data, offs):
    b = data.tobytes()
    return [b[offs[i]:offs[i + 1]].decode("utf-8") for i in range(len(offs) - 1)]
