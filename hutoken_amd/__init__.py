"""hutoken_amd -- MI355X-native batch BPE encode path behind huToken's Python surface.

`import hutoken_amd as hutoken` is the drop-in for the encode direction of the
reference's `hutoken` module (reference hutoken.py:22-43, 122-139):

    hutoken.initialize(vocab_file, special_chars_file, prefix=None, is_byte_encoder=False)
    hutoken.encode(text)                      -> list[int]
    hutoken.batch_encode(texts, num_threads)  -> list[list[int]]

Signatures, return types, exception classes and the messages the reference's tests
pin are kept.  Token ids are bit-exact with the reference's string-keyed path.
The work is done by hand-written HIP kernels behind the C ABI of
include/hutoken_amd.h; there is no CPU fallback: without the native library or a
GPU every call raises.

Also here (not in the reference): `encode_packed` / `encode_packed_device`, the
zero-marshalling entry points for packed UTF-8 + offsets.
"""
import os
import sys
import traceback

from . import _capi

__all__ = ["initialize", "encode", "batch_encode", "encode_packed", "encode_packed_device",
           "decode", "batch_decode", "context"]

_NOT_INIT = ("Vocabulary is not initialized for encoding. "
             "Call 'initialize_encode' function first.")
_BAD_INIT_ARGS = ("Invalid arguments. Expected a string "
                  "(vocab_file_path), a string (special_file_path), "
                  "a string or None (prefix) a bool an"
                  "optional integer (special_token_id), "
                  " an optional string (regex_pattern) and"
                  "a string or None (merges_file_path)")

# process-global context, like the reference's global_encode_context (lib.c:73-74)
_ctx = None


def context():
    """The current hutoken_amd._capi.Context (None before initialize())."""
    return _ctx


def _native_initialize(vocab_file_path, special_file_path, prefix=None, is_byte_encoder=False,
                       special_token_id=-1, pattern=None, merges_file_path=None, device=-1, devices=None):
    # mirrors the argument contract of _hutoken.initialize (lib.c:188-215, "ss|zpizz")
    # devices (not in the reference): several GPUs of this process behind batch_encode (hutk_ctx_add_device)
    global _ctx
    if devices is None and os.environ.get("HUTOKEN_DEVICES"):
        devices = [int(x) for x in os.environ["HUTOKEN_DEVICES"].split(",") if x.strip()]
    if devices:
        device = int(devices[0])
    if not isinstance(vocab_file_path, str) or not isinstance(special_file_path, str) \
            or not (prefix is None or isinstance(prefix, str)) \
            or not (pattern is None or isinstance(pattern, str)) \
            or not (merges_file_path is None or isinstance(merges_file_path, str)) \
            or not isinstance(special_token_id, int):
        raise TypeError(_BAD_INIT_ARGS)
    sh = _capi.shim()
    if sh is not None:
        # the compiled module with the reference's method table owns the context; this wrapper only looks at it
        sh.initialize(vocab_file_path, special_file_path, prefix, bool(is_byte_encoder), special_token_id, pattern,
                      merges_file_path, device)
        old, _ctx = _ctx, _capi.Context.from_handle(sh.handle())
        if old is not None:
            old.close()
        for d in (devices or [])[1:]:
            _ctx.add_device(int(d))
        return None
    # merges_file_path: the id-keyed merge path (lib.c:573-663, core.c:211-337) on the same kernels
    new = _capi.Context(vocab_file_path, special_file_path, prefix, bool(is_byte_encoder), device,
                        merges_path=merges_file_path, devices=devices)
    if pattern is not None:
        # the regex pre-token path (core.c:350-360): libc's regexec finds the words on the host, pretokenizer and
        # merge loop run on the GPU
        try:
            new.set_pattern(pattern)
        except Exception:
            new.close()
            raise
    old, _ctx = _ctx, new
    if old is not None:
        old.close()
    return None


def initialize(model_or_path, *args, **kwargs):
    """hutoken.initialize (reference hutoken.py:22-120): local vocabulary files, or a Hugging Face
    model id / directory that `transformers` can resolve offline (see hutoken_amd/hf.py).

    Merges file: the reference's local-file branch checks that args[6] exists and then
    drops it (hutoken.py:30-43 never hands it to _hutoken.initialize); only its Hugging
    Face branch passes `merges_file_path=`.  The positional form is kept as it is; the
    keyword `merges_file_path=` (the native function's own name for it, lib.c:188-205)
    selects the id-keyed merge path here."""
    if os.path.isfile(model_or_path):
        special_chars_file = args[0] if args else None
        merges_file = args[6] if len(args) > 6 else None
        if special_chars_file and not os.path.isfile(special_chars_file):
            raise ValueError(f"Special characters file '{special_chars_file}' does not exist.")
        if merges_file and not os.path.isfile(merges_file):
            raise ValueError(f"The provided merges file '{merges_file}' does not exist.")
        prefix = kwargs.get("prefix", None)
        is_byte_encoder = kwargs.get("is_byte_encoder", False)
        token_id = kwargs.get("token_id", -1)
        regex_pattern = kwargs.get("pattern", None)
        device = kwargs.get("device", int(os.environ.get("HUTOKEN_DEVICE", "-1")))
        merges_kw = kwargs.get("merges_file_path", None)
        if merges_kw and not os.path.isfile(merges_kw):
            raise ValueError(f"The provided merges file '{merges_kw}' does not exist.")
        return _native_initialize(model_or_path, special_chars_file, prefix, is_byte_encoder, token_id,
                                  regex_pattern, merges_kw, device=device, devices=kwargs.get("devices", None))
    # Hugging Face branch (hutoken.py:44-120): convert the tokenizer to huToken's files, then the same native
    # initialisation, on the id-keyed merge path when the tokenizer has merge rules
    from . import hf
    ex = hf.export(model_or_path, **kwargs)
    try:
        kw = {k: v for k, v in kwargs.items() if k not in ("is_byte_encoder",)}
        kw.setdefault("device", int(os.environ.get("HUTOKEN_DEVICE", "-1")))
        if "token_id" in kw:
            kw["special_token_id"] = kw.pop("token_id")
        return _native_initialize(ex["vocab_file"], ex["special_chars_file"], ex["prefix"], ex["is_byte_encoder"],
                                  *args, merges_file_path=ex["merges_file_path"], **kw)
    except Exception as e:
        traceback.print_exc(file=sys.stderr)
        raise RuntimeError("An unexpected error occured during "
                           f"initialization: {e}") from e


def _pack(texts):
    import numpy as np
    try:
        chunks = [t.encode("utf-8") for t in texts]  # a lone surrogate raises UnicodeEncodeError (reference: crash)
    except AttributeError:
        raise TypeError("bad argument type for built-in operation")
    data = b"".join(chunks)
    if b"\0" in data:  # rare: strdup() semantics of lib.c:770-772, a text ends at its first NUL
        chunks = [b if (z := b.find(b"\0")) < 0 else b[:z] for b in chunks]
        data = b"".join(chunks)
    offs = np.zeros(len(texts) + 1, dtype=np.int64)
    if chunks:
        np.cumsum(np.fromiter(map(len, chunks), dtype=np.int64, count=len(chunks)), out=offs[1:])
    return np.frombuffer(data, dtype=np.uint8), offs


def _native_encode(text):
    sh = _capi.shim()
    if sh is not None:
        return sh.encode(text)
    if _ctx is None:
        raise RuntimeError(_NOT_INIT)
    if not isinstance(text, str):
        raise TypeError(f"argument 1 must be str, not {type(text).__name__}")
    data = text.encode("utf-8")
    if b"\0" in data:
        raise ValueError("embedded null character")
    ids, _rc = _ctx.encode_one(data)  # an over-long word is not reported (lib.c:692-697)
    return ids


def _native_batch_encode(texts, num_threads=1):
    sh = _capi.shim()
    if sh is not None:
        return sh.batch_encode(texts, num_threads)
    if _ctx is None:
        raise RuntimeError(_NOT_INIT)
    if not isinstance(texts, list):
        raise TypeError("Invalid arguments. Expected a list of strings.")
    if not isinstance(num_threads, int):
        raise TypeError("Invalid arguments. Expected a list of strings.")
    data, offs = _pack(texts)
    if num_threads <= 0:
        return [[] for _ in texts]  # no worker starts (lib.c:784-791)
    # an over-long word ends its document silently, as in the reference (core.c:503)
    ids, oo, _st, _rc = _ctx.encode_packed(data, offs)
    flat = ids.tolist()
    bounds = oo.tolist()
    return [flat[bounds[i]:bounds[i + 1]] for i in range(len(texts))]


def encode(text):
    try:
        return _native_encode(text)
    except Exception as e:
        traceback.print_exc(file=sys.stderr)
        raise RuntimeError(f"hutoken: Error encoding text: {e}")


def batch_encode(texts, num_threads=1):
    try:
        return _native_batch_encode(texts, num_threads)
    except Exception as e:
        traceback.print_exc(file=sys.stderr)
        raise RuntimeError(f"hutoken: Error encoding texts: {e}")


def encode_packed(data, offsets):
    """Packed UTF-8 (uint8 array) + int64 offsets[n+1] on the host ->
    (ids int32 array, out_offsets int64[n+1], status int32[n]).  Raises like
    batch_encode."""
    if _ctx is None:
        raise RuntimeError(_NOT_INIT)
    ids, oo, st, _rc = _ctx.encode_packed(data, offsets)
    return ids, oo, st


def encode_packed_device(d_bytes, d_offsets, check=True):
    """Device-resident torch tensors in (uint8 bytes, int64 offsets[n+1]), device
    tensors out: (ids int32[capacity], out_offsets int64[n+1]).  The ids of
    document i are ids[out_offsets[i]:out_offsets[i+1]].  Asynchronous on the
    current torch stream unless check=True, which synchronises and raises on a
    device-side error."""
    import torch
    if _ctx is None:
        raise RuntimeError(_NOT_INIT)
    n_docs = d_offsets.numel() - 1
    n_bytes = d_bytes.numel()
    cap = _ctx.ids_capacity(n_bytes, n_docs)
    dev = d_bytes.device
    ids = torch.empty(max(cap, 1), dtype=torch.int32, device=dev)
    oo = torch.empty(n_docs + 1, dtype=torch.int64, device=dev)
    err = torch.zeros(1, dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream
    _ctx.encode_device(d_bytes.data_ptr(), d_offsets.data_ptr(), n_docs, n_bytes, ids.data_ptr(), cap,
                       oo.data_ptr(), 0, err.data_ptr(), stream)
    if check:
        code = int(err.item())
        if code not in (0, _capi.E_WORD_TOO_LARGE):
            raise RuntimeError(f"hutoken_amd: device-side error {code}")
    return ids, oo


_NOT_INIT_DECODE = ("Vocabulary is not initialized for decoding. "
                    "Call 'initialize_decode' function first.")


def _ids_to_text(raw):
    # PyUnicode_FromString (lib.c:938-939): the C string ends at its first 0x00, then strict UTF-8
    z = raw.find(b"\0")
    return (raw if z < 0 else raw[:z]).decode("utf-8")


def _native_decode(tokens):
    """_hutoken.decode (lib.c:876-951): list[int] -> str."""
    sh = _capi.shim()
    if sh is not None:
        return sh.decode(tokens)
    import numpy as np
    if _ctx is None:
        raise RuntimeError(_NOT_INIT_DECODE)
    if not isinstance(tokens, list):
        raise TypeError("Argument must be a list of integers")
    ids = np.asarray([int(t) for t in tokens], dtype=np.int64).astype(np.int32)  # (int)PyLong_AsLong
    out, oo, _st = _ctx.decode_packed(ids, np.array([0, len(ids)], dtype=np.int64))
    return _ids_to_text(out.tobytes())


def _native_batch_decode(tokens, num_threads=1):
    """_hutoken.batch_decode (lib.c:954-1126): list[list[int]] -> list[str]."""
    sh = _capi.shim()
    if sh is not None:
        return sh.batch_decode(tokens, num_threads)
    import numpy as np
    if _ctx is None:
        raise RuntimeError(_NOT_INIT_DECODE)
    if not isinstance(tokens, list):
        raise TypeError("Failed to parse arguments. Expected a single list of tokens.")
    if len(tokens) <= 0:
        raise ValueError("No tokens provided.")
    for item in tokens:
        if not isinstance(item, list):
            raise TypeError("Each item must be a list of integers.")
    offs = np.zeros(len(tokens) + 1, dtype=np.int64)
    np.cumsum(np.fromiter(map(len, tokens), dtype=np.int64, count=len(tokens)), out=offs[1:])
    flat = np.fromiter((int(t) for item in tokens for t in item), dtype=np.int64, count=int(offs[-1])).astype(np.int32)
    out, oo, _st = _ctx.decode_packed(flat, offs)
    raw = out.tobytes()
    bounds = oo.tolist()
    return [_ids_to_text(raw[bounds[i]:bounds[i + 1]]) for i in range(len(tokens))]


def decode(tokens):
    """hutoken.decode (reference hutoken.py:140-151)."""
    try:
        return _native_decode(tokens)
    except ValueError as e:
        traceback.print_exc(file=sys.stderr)
        raise ValueError(f"hutoken: Error decoding tokens {tokens}: {e}")
    except Exception as e:
        traceback.print_exc(file=sys.stderr)
        raise RuntimeError(f"hutoken: Error decoding tokens: {e}")


def batch_decode(tokens, num_threads=1):
    """hutoken.batch_decode (reference hutoken.py:153-160)."""
    try:
        return _native_batch_decode(tokens, num_threads)
    except Exception as e:
        traceback.print_exc(file=sys.stderr)
        raise RuntimeError(f"hutoken: Error decoding tokens: {e}")
