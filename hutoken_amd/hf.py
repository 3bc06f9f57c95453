"""The Hugging Face branch of hutoken.initialize (reference hutoken.py:44-120), host side only.

`initialize("org/model")` in the reference loads the tokenizer with `transformers`, writes huToken's own
vocabulary / special-characters files into `$XDG_CACHE_HOME/hutoken/<org>/<model>/`, works out `prefix`
and `is_byte_encoder` from the tokenizer, and initialises from those files (with the tokenizer's merges.txt,
i.e. on the id-keyed merge path).  `export()` does the same conversion step by step; `hutoken_amd.initialize`
then builds the device context from the files.  There is no network here or on the GPU box: a model id
resolves only if `transformers` finds it locally (a directory, or its own offline cache).

Two things the reference leaves to the `transformers` version it was written against, made explicit here:
  * byte-level tokenizers: the reference reads `tokenizer.byte_encoder` (slow GPT2Tokenizer, hutoken.py:90-91,
    109-111).  Tokenizers-backed classes of newer `transformers` have no such attribute, so a ByteLevel
    pre-tokenizer/decoder in the backend is recognised as well; the byte table is GPT-2's either way.
  * merges.txt: `save_pretrained` of newer `transformers` writes tokenizer.json only; the merge rules are then
    taken from the backend model and written as merges.txt in the same directory (same text format).
"""
import json
import os
import sys
import traceback

from . import vocab_files as vf

# the bytes the reference writes a replacement for (hutoken.py:15-20)
_SPECIAL_CHARS = vf.SPECIAL_BYTES


def _backend_json(tok):
    be = getattr(tok, "backend_tokenizer", None) or getattr(tok, "_tokenizer", None)
    if be is None or not hasattr(be, "to_str"):
        return None
    try:
        return json.loads(be.to_str())
    except Exception:
        return None


def _has_bytelevel(node):
    if isinstance(node, dict):
        if node.get("type") == "ByteLevel":
            return True
        return any(_has_bytelevel(v) for v in node.values())
    if isinstance(node, list):
        return any(_has_bytelevel(v) for v in node)
    return False


def is_byte_level(tok):
    """True for GPT-2-style byte-level tokenizers (hutoken.py:109-111 and the note in the module docstring)."""
    if getattr(tok, "byte_encoder", None) is not None:
        return True
    js = _backend_json(tok)
    return bool(js) and (_has_bytelevel(js.get("pre_tokenizer")) or _has_bytelevel(js.get("decoder")))


def _merges_text(tok):
    js = _backend_json(tok)
    if not js:
        return None
    merges = (js.get("model") or {}).get("merges")
    if not merges:
        return None
    lines = ["#version: 0.2"]
    for m in merges:  # "a b" (older tokenizer.json) or ["a", "b"]
        lines.append(m if isinstance(m, str) else " ".join(m))
    return "\n".join(lines) + "\n"


def _load(reference):
    """The `transformers` tokenizer behind a model id or directory; the reference's error texts (hutoken.py:46-54)."""
    try:
        from transformers import AutoTokenizer
    except ImportError as e:  # the reference would fail with a NameError here (hutoken.py:4-7, 46)
        raise RuntimeError("hutoken: the Hugging Face branch of initialize() needs the 'transformers' "
                           f"package: {e}") from e
    try:
        tok = AutoTokenizer.from_pretrained(reference)
    except (OSError, ValueError) as e:
        raise ValueError("Could not download Hugging Face tokenizer "
                         f"'{reference}': {e}")
    if not hasattr(tok, "vocab"):
        raise ValueError("Could not extract vocab from Hugging Face "
                         "tokenizer.")
    return tok


def _target(reference):
    """-> (folder, stem): $XDG_CACHE_HOME/hutoken/<org>/<model>/ and <model>, the layout of hutoken.py:55-60."""
    pieces = [x for x in str(reference).replace("\\", "/").split("/") if x]
    if len(pieces) < 2:  # the reference unpacks exactly "<org>/<model>" (hutoken.py:58)
        raise ValueError(f"not enough values to unpack (expected 2, got {len(pieces)})")
    root = os.getenv("XDG_CACHE_HOME", os.path.join(os.path.expanduser("~"), ".cache"))
    folder = os.path.join(root, "hutoken", pieces[-2], pieces[-1])
    os.makedirs(folder, exist_ok=True)
    return folder, pieces[-1]


def _entries(tok):
    """(token bytes, id) in id order; a token that has no UTF-8 form (a lone surrogate) is reported and left out."""
    for token, idx in sorted(tok.vocab.items(), key=lambda kv: kv[1]):
        try:
            yield token.encode("utf-8"), idx
        except Exception as e:  # noqa: BLE001
            sys.stderr.write(f"Failed to process token '{token}': {e}")


def _replacements(tok, byte_level):
    """byte -> replacement string for the bytes of hutoken.py:15-20: the byte-level table when there is one, else what
    the tokenizer itself makes of the character (hutoken.py:88-97)."""
    table = getattr(tok, "byte_encoder", None) or (vf.bytes_to_unicode() if byte_level else None)
    if table is not None:
        return {b: table[b] for b in _SPECIAL_CHARS}
    return {b: "".join(tok.tokenize(chr(b))) for b in _SPECIAL_CHARS}


def _merges_file(tok, folder, reference):
    """merges.txt as save_pretrained left it, or written from the backend model; None without merge rules."""
    path = os.path.join(folder, "merges.txt")
    if os.path.isfile(path):
        return path
    text = _merges_text(tok)
    if text is None:
        sys.stderr.write(f"No merges.txt found for '{reference}'. Continuing without merge rules.\n")
        return None
    with open(path, "w", encoding="utf-8", newline="") as f:
        f.write(text)
    return path


def _written(path, what, write):
    try:
        write()
    except IOError as e:
        traceback.print_exc(file=sys.stderr)
        raise IOError(f"Could not write {what} to '{path}': {e}")
    return path


def export(reference, **options):
    """Convert a Hugging Face tokenizer into huToken's files (what hutoken.py:44-107 does inline).

    Returns dict(vocab_file, special_chars_file, prefix, is_byte_encoder, merges_file_path, tokenizer)."""
    tok = _load(reference)
    folder, stem = _target(reference)
    tok.save_pretrained(folder)
    byte_level = is_byte_level(tok)
    vocab_path = os.path.join(folder, f"{stem}.txt")
    special_path = os.path.join(folder, f"{stem}_special_chars.txt")
    _written(vocab_path, "vocab file", lambda: vf.write_vocab_file(vocab_path, _entries(tok), encoding="utf-8"))
    _written(special_path, "special characters file",
             lambda: vf.write_special_file(special_path, _replacements(tok, byte_level)))
    # the marker a tokenizer puts in front of a word ("hu" -> "\u2581hu"): huToken's prefix (hutoken.py:75-76)
    first_piece = tok.tokenize("hu")[0]
    return dict(vocab_file=vocab_path, special_chars_file=special_path,
                prefix=None if first_piece == "hu" else first_piece[0],
                is_byte_encoder=1 if byte_level else options.get("is_byte_encoder", 0),
                merges_file_path=_merges_file(tok, folder, reference), tokenizer=tok)
