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


def export(model_or_path, **kwargs):
    """Convert a Hugging Face tokenizer into huToken's files (hutoken.py:44-107).

    Returns dict(vocab_file, special_chars_file, prefix, is_byte_encoder, merges_file_path, tokenizer)."""
    try:
        from transformers import AutoTokenizer
    except ImportError as e:  # the reference would fail with a NameError here (hutoken.py:4-7, 46)
        raise RuntimeError("hutoken: the Hugging Face branch of initialize() needs the 'transformers' "
                           f"package: {e}") from e
    try:
        hf_tokenizer = AutoTokenizer.from_pretrained(model_or_path)
    except (OSError, ValueError) as e:
        raise ValueError("Could not download Hugging Face tokenizer "
                         f"'{model_or_path}': {e}")
    if not hasattr(hf_tokenizer, "vocab"):
        raise ValueError("Could not extract vocab from Hugging Face "
                         "tokenizer.")

    cache_dir = os.getenv("XDG_CACHE_HOME", os.path.join(os.path.expanduser("~"), ".cache"))
    parts = [p for p in str(model_or_path).replace("\\", "/").split("/") if p]
    if len(parts) < 2:  # the reference unpacks exactly "<org>/<model>" (hutoken.py:58)
        raise ValueError(f"not enough values to unpack (expected 2, got {len(parts)})")
    org_name, model_name = parts[-2], parts[-1]
    vocab_dir = os.path.join(cache_dir, f"hutoken/{org_name}/{model_name}")
    os.makedirs(vocab_dir, exist_ok=True)
    vocab_file = os.path.join(vocab_dir, f"{model_name}.txt")
    hf_tokenizer.save_pretrained(vocab_dir)

    try:
        with open(vocab_file, "w", encoding="utf-8") as f:
            for token, idx in sorted(hf_tokenizer.vocab.items(), key=lambda item: item[1]):
                try:
                    f.write(vf.hex_line(token.encode("utf-8"), idx))
                except Exception as e:  # e.g. a lone surrogate in a token
                    sys.stderr.write(f"Failed to process token '{token}': {e}")
    except IOError as e:
        traceback.print_exc(file=sys.stderr)
        raise IOError(f"Could not write vocab file to '{vocab_file}': {e}")

    hu_tokenized = hf_tokenizer.tokenize("hu")[0]
    prefix = hu_tokenized[0] if hu_tokenized != "hu" else None

    byte_level = is_byte_level(hf_tokenizer)
    special_chars_file = os.path.join(vocab_dir, f"{model_name}_special_chars.txt")
    try:
        with open(special_chars_file, "w", encoding="utf-8") as f:
            table = getattr(hf_tokenizer, "byte_encoder", None) or (vf.bytes_to_unicode() if byte_level else None)
            for char in _SPECIAL_CHARS:
                if table is not None:
                    value = table[char]
                else:
                    value = "".join(hf_tokenizer.tokenize(chr(char)))
                f.write(f"{char} == {value}\n")
    except IOError as e:
        traceback.print_exc(file=sys.stderr)
        raise IOError("Could not write special characters file to "
                      f"'{special_chars_file}': {e}")

    merges_file_path = os.path.join(vocab_dir, "merges.txt")
    if not os.path.isfile(merges_file_path):
        text = _merges_text(hf_tokenizer)
        if text is not None:
            with open(merges_file_path, "w", encoding="utf-8", newline="") as f:
                f.write(text)
        else:
            merges_file_path = None
            sys.stderr.write(f"No merges.txt found for '{model_or_path}'. Continuing without merge rules.\n")

    is_byte_encoder = 1 if byte_level else kwargs.get("is_byte_encoder", 0)
    return dict(vocab_file=vocab_file, special_chars_file=special_chars_file, prefix=prefix,
                is_byte_encoder=is_byte_encoder, merges_file_path=merges_file_path, tokenizer=hf_tokenizer)
