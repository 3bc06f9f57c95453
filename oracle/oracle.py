"""ctypes front end of the CPU oracle (oracle/hutk_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The product package never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libhutk_oracle.so")

DOC_OK, DOC_WORD_TOO_LARGE, DOC_INVALID_UTF8, DOC_NOMEM = 0, 1, 2, 3
_ERR_KIND = {1: FileNotFoundError, 2: ValueError, 3: MemoryError}

WORD_TOO_LARGE_MSG = "A single word in the input text is too large to be processed."


def build(force=False):
    """Compile the oracle with gcc (seconds)."""
    src = os.path.join(_HERE, "hutk_oracle.c")
    if (not force and os.path.exists(_LIB_PATH)
            and os.path.getmtime(_LIB_PATH) >= os.path.getmtime(src)):
        return _LIB_PATH
    subprocess.check_call(["make", "-s", "-C", _HERE, "_build/libhutk_oracle.so"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        L.hto_create.restype = C.c_void_p
        L.hto_create.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int,
                                 C.POINTER(C.c_int), C.c_char_p, C.c_size_t]
        L.hto_destroy.argtypes = [C.c_void_p]
        L.hto_load_merges.restype = C.c_int
        L.hto_load_merges.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_size_t]
        L.hto_set_pattern.restype = C.c_int
        L.hto_set_pattern.argtypes = [C.c_void_p, C.c_char_p]
        L.hto_has_merges.restype = C.c_int
        L.hto_has_merges.argtypes = [C.c_void_p]
        L.hto_rule_count.restype = C.c_uint64
        L.hto_rule_count.argtypes = [C.c_void_p]
        L.hto_vocab_count.restype = C.c_uint64
        L.hto_vocab_count.argtypes = [C.c_void_p]
        L.hto_vocab_lookup.restype = C.c_int
        L.hto_vocab_lookup.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t,
                                       C.POINTER(C.c_int32)]
        L.hto_special.restype = C.c_char_p
        L.hto_special.argtypes = [C.c_void_p, C.c_int]
        L.hto_split_words.restype = C.c_size_t
        L.hto_split_words.argtypes = [C.c_char_p, C.c_size_t, C.c_void_p, C.c_size_t]
        L.hto_encode.restype = C.c_int
        L.hto_encode.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t,
                                 C.POINTER(C.POINTER(C.c_int32)), C.POINTER(C.c_size_t)]
        L.hto_free.argtypes = [C.c_void_p]
        L.hto_decode.restype = C.c_int
        L.hto_decode.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.POINTER(C.c_uint8)),
                                 C.POINTER(C.c_size_t)]
        L.hto_encode_batch.restype = C.c_int
        L.hto_encode_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                       C.c_int, C.POINTER(C.POINTER(C.c_int32)),
                                       C.c_void_p, C.c_void_p]
        _lib = L
    return _lib


def split_words(data: bytes):
    """Word start offsets of one document (parser.c:24-88 restated)."""
    L = lib()
    n = L.hto_split_words(data, len(data), None, 0)
    starts = np.zeros(max(n, 1), dtype=np.uint32)
    L.hto_split_words(data, len(data), starts.ctypes.data, n)
    return starts[:n].tolist()


def pack(texts):
    """list[str|bytes] -> (uint8 array, int64 offsets).  A str is encoded to
    UTF-8 and cut at its first NUL, as strdup() does in lib.c:770-772."""
    chunks = []
    offs = np.zeros(len(texts) + 1, dtype=np.int64)
    for i, t in enumerate(texts):
        b = t.encode("utf-8") if isinstance(t, str) else bytes(t)
        z = b.find(b"\0")
        if z >= 0:
            b = b[:z]
        chunks.append(b)
        offs[i + 1] = offs[i] + len(b)
    data = np.frombuffer(b"".join(chunks), dtype=np.uint8)
    return data, offs


class Oracle:
    def __init__(self, vocab_path, special_path, prefix=None, is_byte_encoder=False, merges_path=None, pattern=None):
        L = lib()
        kind = C.c_int(0)
        err = C.create_string_buffer(256)
        self._h = L.hto_create(os.fsencode(vocab_path), os.fsencode(special_path),
                               None if prefix is None else prefix.encode("utf-8"),
                               1 if is_byte_encoder else 0, C.byref(kind), err, 256)
        if not self._h:
            raise _ERR_KIND.get(kind.value, RuntimeError)(err.value.decode())
        if merges_path is not None:  # the id-keyed merge path (lib.c:573-663)
            rc = L.hto_load_merges(self._h, os.fsencode(merges_path), err, 256)
            if rc:
                raise _ERR_KIND.get(rc, RuntimeError)(err.value.decode())
        if pattern is not None:  # the regex pre-token path (core.c:350-360)
            rc = L.hto_set_pattern(self._h, pattern.encode("utf-8"))
            if rc:
                raise _ERR_KIND.get(rc, RuntimeError)("Regex could not be compiled.")

    @property
    def has_merges(self):
        return bool(lib().hto_has_merges(self._h))

    @property
    def rule_count(self):
        return lib().hto_rule_count(self._h)

    def close(self):
        if getattr(self, "_h", None):
            try:
                lib().hto_destroy(self._h)
            except TypeError:  # interpreter shutdown: the module's globals are gone already
                pass
            self._h = None

    __del__ = close

    @property
    def vocab_count(self):
        return lib().hto_vocab_count(self._h)

    def lookup(self, key: bytes):
        v = C.c_int32(0)
        return v.value if lib().hto_vocab_lookup(self._h, key, len(key), C.byref(v)) else None

    def special(self, idx):
        return lib().hto_special(self._h, idx)

    def encode_bytes(self, data: bytes):
        """-> (ids list, status)"""
        L = lib()
        p = C.POINTER(C.c_int32)()
        n = C.c_size_t(0)
        st = L.hto_encode(self._h, data, len(data), C.byref(p), C.byref(n))
        ids = [p[i] for i in range(n.value)]
        L.hto_free(p)
        return ids, st

    def encode(self, text):
        """hutoken.encode(): an over-long word is not reported (lib.c:692-697)."""
        data = text.encode("utf-8") if isinstance(text, str) else bytes(text)
        return self.encode_bytes(data)[0]

    def decode_bytes(self, ids):
        """Decode direction (core.c:513-581): -> (bytes, status); status DEC_RANGE/HOLE/AMBIGUOUS as in the header."""
        L = lib()
        arr = np.ascontiguousarray(ids, dtype=np.int32)
        p = C.POINTER(C.c_uint8)()
        n = C.c_size_t(0)
        st = L.hto_decode(self._h, arr.ctypes.data, len(arr), C.byref(p), C.byref(n))
        out = bytes(bytearray(p[i] for i in range(n.value))) if st == 0 else b""
        if st == 0:
            L.hto_free(p)
        return out, st

    def decode(self, ids):
        """hutoken.decode(): ValueError for ids out of range, the C string ends at its first NUL, then UTF-8."""
        out, st = self.decode_bytes(ids)
        if st == 1:
            raise ValueError("Element must be non-negative and less than vocab size.")
        if st:
            raise RuntimeError(f"oracle decode status {st}")
        z = out.find(b"\0")
        return (out if z < 0 else out[:z]).decode("utf-8")

    def encode_packed(self, data, offsets, num_threads=1):
        """-> (ids int32 array, out_offsets int64 array, status int32 array)"""
        L = lib()
        data = np.ascontiguousarray(data, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        n = len(offsets) - 1
        out_offs = np.zeros(n + 1, dtype=np.int64)
        status = np.zeros(max(n, 1), dtype=np.int32)
        p = C.POINTER(C.c_int32)()
        rc = L.hto_encode_batch(self._h, data.ctypes.data, offsets.ctypes.data, n,
                                num_threads, C.byref(p), out_offs.ctypes.data,
                                status.ctypes.data)
        if rc != 0:
            raise MemoryError("oracle batch failed")
        total = int(out_offs[n])
        ids = np.ctypeslib.as_array(p, shape=(max(total, 1),))[:total].copy()
        L.hto_free(p)
        return ids, out_offs, status[:n]

    def batch_encode(self, texts, num_threads=1):
        """hutoken.batch_encode().  An over-long word is NOT reported: core.c:503 resets
        task->error_msg after the word loop, so lib.c:796-808 never fires and the
        document simply ends before that word, exactly as in encode()."""
        data, offs = pack(texts)
        ids, oo, status = self.encode_packed(data, offs, num_threads)
        for s in status:
            if s not in (DOC_OK, DOC_WORD_TOO_LARGE):
                raise RuntimeError(f"oracle document status {s}")
        return [ids[oo[i]:oo[i + 1]].tolist() for i in range(len(texts))]
