"""ctypes binding of include/hutoken_amd.h (the C-ABI shared library).

The library is built in-tree (hutoken_amd/lib/libhutoken_amd.so).  Nothing here
computes token ids: if the library or a GPU is missing the calls raise.
"""
import ctypes as C
import importlib.util
import os
import sys

from . import build as _build

OK = 0
E_FILE_NOT_FOUND, E_VALUE, E_MEMORY, E_ARG, E_DEVICE, E_UNSUPPORTED = 1, 2, 3, 4, 5, 6
E_CAPACITY, E_NUL_BYTE, E_WORD_TOO_LARGE, E_INVALID_UTF8 = 7, 8, 9, 10
DOC_OK, DOC_WORD_TOO_LARGE, DOC_INVALID_UTF8 = 0, 1, 2

# every symbol include/hutoken_amd.h declares
EXPORTS = [
    "hutk_ctx_create", "hutk_ctx_create_merges", "hutk_ctx_set_pattern", "hutk_ctx_add_device", "hutk_ctx_device_count", "hutk_uses_merges", "hutk_ctx_destroy", "hutk_last_error", "hutk_ids_capacity",
    "hutk_encode_batch", "hutk_encode_batch_device", "hutk_encode", "hutk_vocab_size", "hutk_host_alloc",
    "hutk_host_free", "hutk_decode_batch", "hutk_decode_batch_device",
    "hutk_pair_table_entries", "hutk_device_ordinal", "hutk_table_stats", "hutk_last_timing",
    "hutk_set_timing", "hutk_debug_pairs_second", "hutk_debug_long_words", "hutk_debug_profile", "hutk_debug_profile_read", "hutk_debug_profile_raw", "hutk_debug_tile_bytes",
    "hutk_debug_seam", "hutk_debug_seam2_cut",
]

_lib = None


def library_path():
    return _build.LIB_HIP


def _share_torch_hip_runtime():
    """PyTorch-ROCm wheels carry their own copy of the HIP and HSA runtimes (torch/lib), and a process can
    open the GPU through ONE runtime only: with /opt/rocm's loaded first by this library, a later `import
    torch` finds "No HIP GPUs".  So when torch is installed but not yet imported, its copy is loaded here
    (by path, without importing torch); libhutoken_amd.so then binds to it by soname, and torch to the
    same file later.  HUTOKEN_AMD_SYSTEM_HIP=1 keeps /opt/rocm's."""
    if "torch" in sys.modules or os.environ.get("HUTOKEN_AMD_SYSTEM_HIP"):
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if not spec or not spec.submodule_search_locations:
        return
    p = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(p):
        try:
            C.CDLL(p, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def load(build_if_missing=True):
    """Load libhutoken_amd.so (building it with hipcc when it is absent)."""
    global _lib
    if _lib is not None:
        return _lib
    _share_torch_hip_runtime()
    # HUTOKEN_AMD_LIB: another build of the same library (tools/ab.py compares two on one GPU box)
    path = os.environ.get("HUTOKEN_AMD_LIB") or _build.LIB_HIP
    if path == _build.LIB_HIP and build_if_missing and (not os.path.exists(path) or os.environ.get("HUTOKEN_AMD_REBUILD")):
        _build.build_hip()
    if not os.path.exists(path):
        raise RuntimeError("hutoken_amd: native library %s is missing (run `python -m hutoken_amd.build`)" % path)
    L = C.CDLL(path)
    vp, i64, i32 = C.c_void_p, C.c_int64, C.c_int
    L.hutk_ctx_create.restype = i32
    L.hutk_ctx_create.argtypes = [C.POINTER(vp), C.c_char_p, C.c_char_p, C.c_char_p, i32, i32]
    L.hutk_ctx_create_merges.restype = i32
    L.hutk_ctx_create_merges.argtypes = [C.POINTER(vp), C.c_char_p, C.c_char_p, C.c_char_p, i32, C.c_char_p, i32]
    if hasattr(L, "hutk_ctx_set_pattern"):  # (older builds under tools/ab.py lack it)
        L.hutk_ctx_set_pattern.restype = i32
        L.hutk_ctx_set_pattern.argtypes = [vp, C.c_char_p]
    L.hutk_uses_merges.restype = i32
    L.hutk_uses_merges.argtypes = [vp]
    L.hutk_decode_batch.restype = i32
    L.hutk_decode_batch.argtypes = [vp, vp, vp, i64, vp, i64, vp, vp]
    L.hutk_decode_batch_device.restype = i32
    L.hutk_decode_batch_device.argtypes = [vp, vp, vp, i64, i64, vp, i64, vp, vp, vp, vp]
    L.hutk_host_alloc.restype = vp
    L.hutk_host_alloc.argtypes = [C.c_size_t]
    L.hutk_host_free.restype = None
    L.hutk_host_free.argtypes = [vp]
    L.hutk_ctx_destroy.restype = None
    L.hutk_ctx_destroy.argtypes = [vp]
    L.hutk_last_error.restype = C.c_char_p
    L.hutk_last_error.argtypes = []
    L.hutk_ids_capacity.restype = i64
    L.hutk_ids_capacity.argtypes = [vp, i64, i64]
    L.hutk_encode_batch.restype = i32
    L.hutk_encode_batch.argtypes = [vp, vp, vp, i64, vp, i64, vp, vp]
    L.hutk_encode_batch_device.restype = i32
    L.hutk_encode_batch_device.argtypes = [vp, vp, vp, i64, i64, vp, i64, vp, vp, vp, vp]
    L.hutk_encode.restype = i32
    L.hutk_encode.argtypes = [vp, vp, i64, vp, i64, C.POINTER(i64), C.POINTER(C.c_int32)]
    L.hutk_vocab_size.restype = i64
    L.hutk_vocab_size.argtypes = [vp]
    L.hutk_pair_table_entries.restype = i64
    L.hutk_pair_table_entries.argtypes = [vp]
    L.hutk_device_ordinal.restype = i32
    L.hutk_device_ordinal.argtypes = [vp]
    L.hutk_last_timing.restype = i32
    L.hutk_last_timing.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.hutk_set_timing.restype = None
    L.hutk_set_timing.argtypes = [vp, i32]
    L.hutk_table_stats.restype = i32
    L.hutk_table_stats.argtypes = [vp, vp]
    if hasattr(L, "hutk_debug_pairs_second"):  # (older builds under tools/ab.py lack it)
        L.hutk_debug_pairs_second.restype = i64
        L.hutk_debug_pairs_second.argtypes = [vp]
    if hasattr(L, "hutk_debug_long_words"):
        L.hutk_debug_long_words.restype = i64
        L.hutk_debug_long_words.argtypes = [vp]
    if hasattr(L, "hutk_ctx_add_device"):
        L.hutk_ctx_add_device.restype = i32
        L.hutk_ctx_add_device.argtypes = [vp, i32]
        L.hutk_ctx_device_count.restype = i32
        L.hutk_ctx_device_count.argtypes = [vp]
    if hasattr(L, "hutk_debug_seam"):
        L.hutk_debug_seam.restype = i32
        L.hutk_debug_seam.argtypes = [vp, vp]
    if hasattr(L, "hutk_debug_seam2_cut"):
        L.hutk_debug_seam2_cut.restype = i32
        L.hutk_debug_seam2_cut.argtypes = [vp, C.c_uint32, C.c_uint32]
    L.hutk_debug_profile.restype = i32
    L.hutk_debug_profile.argtypes = [vp, i32]
    L.hutk_debug_tile_bytes.restype = i32
    L.hutk_debug_tile_bytes.argtypes = []
    L.hutk_debug_profile_read.restype = i32
    L.hutk_debug_profile_read.argtypes = [vp, i64, vp]
    if hasattr(L, "hutk_debug_profile_raw"):
        L.hutk_debug_profile_raw.restype = i32
        L.hutk_debug_profile_raw.argtypes = [vp, i64, vp]
    _lib = L
    return L


_shim = False


def shim():
    """The compiled CPython shim (csrc/pyshim/_hutoken_amd.c: the reference's _hutoken method table on this library), or
    None when it is not built or HUTOKEN_AMD_NO_SHIM is set.  Loaded after the library itself (see load())."""
    global _shim
    if _shim is False:
        _shim = None
        path = _build.LIB_PYSHIM
        if not os.environ.get("HUTOKEN_AMD_NO_SHIM") and not os.environ.get("HUTOKEN_AMD_LIB"):
            try:
                try:
                    _build.build_pyshim(force=bool(os.environ.get("HUTOKEN_AMD_REBUILD")))  # (checks for staleness itself)
                except Exception:
                    if not os.path.exists(path):  # no compiler and nothing built earlier
                        raise
                load()
                spec = importlib.util.spec_from_file_location("_hutoken_amd", path)
                mod = importlib.util.module_from_spec(spec)
                spec.loader.exec_module(mod)
                _shim = mod
            except Exception as e:  # no compiler, no Python.h: the ctypes path does the same work, 4x slower on lists
                import warnings
                warnings.warn("hutoken_amd: the compiled CPython shim is not available (%s); using ctypes" % (e,),
                              RuntimeWarning, stacklevel=2)
                _shim = None
    return _shim


def last_error():
    return load().hutk_last_error().decode("utf-8", "replace")


_EXC = {E_FILE_NOT_FOUND: FileNotFoundError, E_VALUE: ValueError, E_MEMORY: MemoryError,
        E_ARG: TypeError, E_DEVICE: RuntimeError, E_UNSUPPORTED: ValueError,
        E_CAPACITY: RuntimeError, E_NUL_BYTE: ValueError, E_WORD_TOO_LARGE: RuntimeError,
        E_INVALID_UTF8: ValueError}


def raise_for(code):
    if code != OK:
        raise _EXC.get(code, RuntimeError)(last_error())


class PinnedArray:
    """A numpy array on page-locked host memory from hutk_host_alloc (kept alive by this object)."""

    def __init__(self, n, dtype):
        import numpy as np
        self.dtype = np.dtype(dtype)
        self.nbytes = max(int(n) * self.dtype.itemsize, 1)
        self._p = load().hutk_host_alloc(self.nbytes)
        if not self._p:
            raise MemoryError("hutk_host_alloc failed")
        buf = (C.c_uint8 * self.nbytes).from_address(self._p)
        self.array = np.frombuffer(buf, dtype=self.dtype, count=int(n))

    def close(self):
        if getattr(self, "_p", None):
            self.array = None
            load().hutk_host_free(self._p)
            self._p = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Context:
    """Owns one hutk_ctx."""

    def __init__(self, vocab_path, special_path, prefix=None, is_byte_encoder=False, device=-1, merges_path=None,
                 devices=None):
        """devices: several ordinals of this process; encode_packed() / hutk_encode_batch then spreads a large batch
        over them (hutk_ctx_add_device).  The first one is the context's own device."""
        L = load()
        if devices:
            device = int(devices[0])
        h = C.c_void_p()
        rc = L.hutk_ctx_create_merges(C.byref(h), os.fsencode(vocab_path), os.fsencode(special_path),
                                      None if prefix is None else prefix.encode("utf-8"),
                                      1 if is_byte_encoder else 0,
                                      None if merges_path is None else os.fsencode(merges_path), device)
        raise_for(rc)
        self._h = h
        self._owned = True
        for d in (devices or [])[1:]:
            try:
                self.add_device(int(d))
            except Exception:
                self.close()
                raise

    def add_device(self, device):
        raise_for(load().hutk_ctx_add_device(self._h, device))

    @property
    def device_count(self):
        return load().hutk_ctx_device_count(self._h)

    @classmethod
    def from_handle(cls, address):
        """A view of a hutk_ctx that somebody else owns (the CPython shim's module-global context)."""
        self = cls.__new__(cls)
        self._h = C.c_void_p(address)
        self._owned = False
        return self

    def set_pattern(self, pattern):
        """The regex pre-token path (initialize's `pattern`, a POSIX ERE); None: the hand-written splitter."""
        raise_for(load().hutk_ctx_set_pattern(self._h, None if pattern is None else pattern.encode("utf-8")))

    @property
    def uses_merges(self):
        """True when the id-keyed merge path (merges file) is in force."""
        return bool(load().hutk_uses_merges(self._h))

    def close(self):
        if getattr(self, "_h", None):
            if getattr(self, "_owned", True):
                load().hutk_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def handle(self):
        return self._h

    def ids_capacity(self, n_bytes, n_docs):
        return load().hutk_ids_capacity(self._h, n_bytes, n_docs)

    def table_stats(self):
        import numpy as np
        out = np.zeros(8, dtype=np.int64)
        raise_for(load().hutk_table_stats(self._h, out.ctypes.data))
        keys = ["n_keys", "n_vocab_sym", "n_sym", "n_pairs", "pair_slots", "rank_is_sym", "ident_ids", "n_word_entries"]
        d = dict(zip(keys, out.tolist()))
        if hasattr(load(), "hutk_debug_long_words"):
            d["n_long_word_entries"] = int(load().hutk_debug_long_words(self._h))
        return d

    def seam_map(self):
        """-> (uint32[256], in use): bit y - 0xE0 of entry x set = some merge can join input bytes x | y."""
        import numpy as np
        out = np.zeros(256, dtype=np.uint32)
        on = load().hutk_debug_seam(self._h, out.ctypes.data)
        return out, bool(on)

    def seam2_cut(self, a3, b3):
        """The seam map's second level: True when no token can span the three-byte characters a3 | b3 (bytes objects)."""
        return bool(load().hutk_debug_seam2_cut(self._h, int.from_bytes(a3, "little"), int.from_bytes(b3, "little")))

    def encode_packed(self, data, offsets, want_status=True):
        """Host numpy buffers in, host numpy buffers out.
        -> (ids int32, out_offsets int64, status int32, return code)"""
        import numpy as np
        L = load()
        data = np.ascontiguousarray(data, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        n = len(offsets) - 1
        nbytes = int(offsets[n]) if n >= 0 else 0
        cap = self.ids_capacity(nbytes, n)
        ids = np.empty(max(cap, 1), dtype=np.int32)
        oo = np.zeros(n + 1, dtype=np.int64)
        st = np.zeros(max(n, 1), dtype=np.int32)
        rc = L.hutk_encode_batch(self._h, data.ctypes.data if nbytes else None, offsets.ctypes.data, n,
                                 ids.ctypes.data, cap, oo.ctypes.data, st.ctypes.data)
        if rc not in (OK, E_WORD_TOO_LARGE):
            raise_for(rc)
        return ids[: int(oo[n])], oo, st[:n], rc

    def decode_packed(self, ids, id_offsets):
        """Decode direction, host numpy buffers: ids int32 + id_offsets int64[n+1] ->
        (bytes uint8, out_offsets int64[n+1], status int32[n]).  Two calls: sizes, then the text."""
        import numpy as np
        L = load()
        ids = np.ascontiguousarray(ids, dtype=np.int32)
        id_offsets = np.ascontiguousarray(id_offsets, dtype=np.int64)
        n = len(id_offsets) - 1
        oo = np.zeros(n + 1, dtype=np.int64)
        st = np.zeros(max(n, 1), dtype=np.int32)
        pid = ids.ctypes.data if len(ids) else None
        raise_for(L.hutk_decode_batch(self._h, pid, id_offsets.ctypes.data, n, None, 0, oo.ctypes.data, st.ctypes.data))
        total = int(oo[n])
        out = np.empty(max(total, 1), dtype=np.uint8)
        raise_for(L.hutk_decode_batch(self._h, pid, id_offsets.ctypes.data, n, out.ctypes.data, total, oo.ctypes.data,
                                      st.ctypes.data))
        return out[:total], oo, st[:n]

    def decode_device(self, d_ids, d_id_offsets, n_docs, n_ids, d_bytes_out, bytes_cap, d_out_offsets, d_status,
                      d_err, stream):
        """Raw device pointers (ints); asynchronous on `stream`."""
        raise_for(load().hutk_decode_batch_device(self._h, d_ids, d_id_offsets, n_docs, n_ids, d_bytes_out, bytes_cap,
                                                  d_out_offsets, d_status, d_err, stream))

    def encode_one(self, data: bytes):
        """-> (ids list, return code)"""
        import numpy as np
        L = load()
        n = len(data)
        cap = self.ids_capacity(n, 1)
        ids = np.empty(max(cap, 1), dtype=np.int32)
        n_ids = C.c_int64(0)
        st = C.c_int32(0)
        buf = C.create_string_buffer(data, n) if n else None
        rc = L.hutk_encode(self._h, C.cast(buf, C.c_void_p) if n else None, n, ids.ctypes.data, cap,
                           C.byref(n_ids), C.byref(st))
        if rc not in (OK, E_WORD_TOO_LARGE):
            raise_for(rc)
        return ids[: n_ids.value].tolist(), rc

    def encode_device(self, d_bytes, d_offsets, n_docs, n_bytes, d_ids, ids_cap, d_out_offsets,
                      d_status=0, d_err=0, stream=0):
        """Raw device pointers (ints); asynchronous on `stream`."""
        rc = load().hutk_encode_batch_device(self._h, d_bytes, d_offsets, n_docs, n_bytes, d_ids, ids_cap,
                                             d_out_offsets, d_status or None, d_err or None, stream or None)
        raise_for(rc)

    def profile(self, enable):
        load().hutk_debug_profile(self._h, 1 if enable else 0)

    def profile_read(self, n_tiles):
        import numpy as np
        out = np.zeros(10, dtype=np.float64)
        raise_for(load().hutk_debug_profile_read(self._h, n_tiles, out.ctypes.data))
        return out.tolist()

    def profile_raw(self, n_tiles):
        import numpy as np
        out = np.zeros((n_tiles, 10), dtype=np.int64)
        raise_for(load().hutk_debug_profile_raw(self._h, n_tiles, out.ctypes.data))
        return out

    def last_timing(self):
        a, b = C.c_float(0), C.c_float(0)
        raise_for(load().hutk_last_timing(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value
