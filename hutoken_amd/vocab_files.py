"""Writers for huToken's two text file formats and the GPT-2 byte alphabet.

Formats (reference README.md:269-319; writers hutoken.py:64-73, 88-97 and
scripts/convert.py:9-15):

  vocab file      one token per line:  0xHH0xHH... == <int id>\\n
                  (every line, the last one too, must end in '\\n':
                  lib.c:264-289 drops an unterminated final line)
  special file    one byte per line:   <decimal 0..255> == <utf-8 string>\\n
                  lines are read in 31-character chunks (lib.c:483)
"""

# bytes the GPT-2 byte encoder remaps (reference hutoken.py:15-20)
SPECIAL_BYTES = list(range(0, 33)) + list(range(127, 161)) + [173]


def bytes_to_unicode():
    """GPT-2's byte -> visible character table: printable Latin-1 bytes map to
    themselves, the other 68 bytes to U+0100.. in ascending byte order."""
    keep = list(range(33, 127)) + list(range(161, 173)) + list(range(174, 256))
    table = {b: chr(b) for b in keep}
    n = 0
    for b in range(256):
        if b not in table:
            table[b] = chr(256 + n)
            n += 1
    return table


def byte_token_order():
    """Byte values in GPT-2 id order (ids 0..255 of a GPT-2-shaped vocab)."""
    keep = list(range(33, 127)) + list(range(161, 173)) + list(range(174, 256))
    return keep + [b for b in range(256) if b not in set(keep)]


def encode_visible(raw: bytes, table=None) -> bytes:
    """Raw bytes -> UTF-8 of their GPT-2 visible characters (what a vocab key
    for those bytes looks like in a byte-encoder vocabulary)."""
    table = table or bytes_to_unicode()
    return "".join(table[b] for b in raw).encode("utf-8")


def hex_line(token: bytes, idx: int) -> str:
    return "".join("0x%02X" % b for b in token) + " == %d\n" % idx


def write_vocab_file(path, entries, encoding="ascii"):
    """entries: iterable of (token bytes, id)."""
    with open(path, "w", encoding=encoding) as f:
        for tok, idx in entries:
            f.write(hex_line(tok, idx))


def write_special_file(path, mapping):
    """mapping: {byte value: replacement str}."""
    with open(path, "w", encoding="utf-8") as f:
        for b in sorted(mapping):
            f.write("%d == %s\n" % (b, mapping[b]))


def gpt2_special_mapping():
    t = bytes_to_unicode()
    return {b: t[b] for b in SPECIAL_BYTES}


def llama_special_mapping():
    """SentencePiece/Llama-shaped special file: space -> U+2581, control bytes
    and DEL -> byte-fallback literals (what hutoken.py:88-97 writes for such a
    tokenizer)."""
    m = {32: "▁"}
    for b in list(range(0, 32)) + [127]:
        m[b] = "<0x%02X>" % b
    return m
