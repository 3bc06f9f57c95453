"""Loader for the reference itself, compiled by `make -C oracle ref` into
oracle/_ref/ (git-ignored; travels to the GPU box as a prebuilt file).

TEST INFRASTRUCTURE ONLY.  Used to validate the restatement, to generate the
golden fixtures (tools/make_golden.py) and, when present, as the "reference"
kind of bench.py's cpu_baseline.  Never imported by the product package.

Only the compiled `_hutoken` module is loaded: the wrapper functions of
/root/reference/hutoken.py:122-139 are one-line forwards and are restated in
RefTokenizer below so that nothing here needs /root/reference at run time.
"""
import glob
import importlib.util
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_REF_DIR = os.path.join(_HERE, "_ref")


def so_path():
    hits = sorted(glob.glob(os.path.join(_REF_DIR, "_hutoken*.so")))
    return hits[0] if hits else None


def build():
    """Compile the reference from /root/reference if it is there (container
    only; the GPU box uses the prebuilt file)."""
    if os.path.isdir("/root/reference/src"):
        subprocess.check_call(["make", "-s", "-C", _HERE, "ref"])
    return so_path()


def available():
    return so_path() is not None


_mod = None


def module():
    """The reference's `_hutoken` extension module (process-global state!)."""
    global _mod
    if _mod is None:
        p = so_path()
        if p is None:
            raise ImportError("oracle/_ref/_hutoken*.so not built (make -C oracle ref)")
        spec = importlib.util.spec_from_file_location("_hutoken", p)
        _mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(_mod)
    return _mod


class RefTokenizer:
    """The reference initialised on local vocab/special files.  The reference
    keeps ONE process-global context (lib.c:73-74): constructing a second
    RefTokenizer re-initialises it for every holder."""

    def __init__(self, vocab_path, special_path, prefix=None, is_byte_encoder=False, merges_path=None, pattern=None):
        self.m = module()
        # merges_path: the id-keyed merge path (lib.c:573-663, core.c:457-477); pattern: the regex pre-token path
        self.m.initialize(vocab_path, special_path, prefix, is_byte_encoder, -1, pattern, merges_path)

    def encode(self, text):
        return self.m.encode(text)

    def encode_bytes(self, data: bytes):
        """Marshalling-free call of the reference's internal seam
        `void encode(struct EncodeTask*)` (core.h:11, taskqueue.h:40-45) on raw
        bytes (no 0x00), which a Python str cannot carry when they are not
        valid UTF-8.  -> (ids, error message or None)"""
        import ctypes as C

        class IntVector(C.Structure):  # vector.h:6-10
            _fields_ = [("data", C.POINTER(C.c_int)), ("size", C.c_size_t),
                        ("capacity", C.c_size_t)]

        class EncodeTask(C.Structure):  # taskqueue.h:40-45
            _fields_ = [("text", C.c_char_p), ("ctx", C.c_void_p),
                        ("tokens", C.POINTER(IntVector)), ("error_msg", C.c_char_p)]

        L = C.CDLL(so_path())
        L.vector_init.argtypes = [C.POINTER(IntVector), C.c_size_t]
        L.vector_free.argtypes = [C.POINTER(IntVector)]
        L.encode.argtypes = [C.POINTER(EncodeTask)]
        L.encode.restype = None
        ctx = C.c_void_p.in_dll(L, "global_encode_context")
        vec = IntVector()
        L.vector_init(C.byref(vec), 256)
        task = EncodeTask(bytes(data), ctx, C.pointer(vec), None)
        L.encode(C.byref(task))
        ids = [vec.data[i] for i in range(vec.size)]
        L.vector_free(C.byref(vec))
        return ids, (task.error_msg.decode() if task.error_msg else None)

    def batch_encode(self, texts, num_threads=1):
        return self.m.batch_encode(texts, num_threads)

    def seam_batch(self, data, offsets, num_threads):
        """The reference's C core without its list marshalling: num_threads pthreads calling the internal
        seam encode(struct EncodeTask*) once per document of a packed batch (oracle/ref_seam.c).
        -> (total ids, seconds)"""
        import ctypes as C

        import numpy as np
        lib = os.path.join(_HERE, "_build", "libref_seam.so")
        src = os.path.join(_HERE, "ref_seam.c")
        if not os.path.exists(lib) or os.path.getmtime(lib) < os.path.getmtime(src):
            subprocess.check_call(["make", "-s", "-C", _HERE, "_build/libref_seam.so"])
        L = C.CDLL(lib)
        L.ref_seam_batch.restype = C.c_int
        L.ref_seam_batch.argtypes = [C.c_char_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int,
                                     C.POINTER(C.c_int64), C.POINTER(C.c_double)]
        data = np.ascontiguousarray(data, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        n_ids, sec = C.c_int64(0), C.c_double(0)
        rc = L.ref_seam_batch(os.fsencode(so_path()), data.ctypes.data, offsets.ctypes.data, len(offsets) - 1,
                              int(num_threads), C.byref(n_ids), C.byref(sec))
        if rc:
            raise RuntimeError("ref_seam_batch failed (%d)" % rc)
        return n_ids.value, sec.value

    def decode(self, ids):
        return self.m.decode(ids)

    def batch_decode(self, ids_lists, num_threads=1):
        return self.m.batch_decode(ids_lists, num_threads)
