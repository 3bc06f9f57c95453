"""The Hugging Face branch of initialize() (reference hutoken.py:44-120).

A tiny GPT-2-shaped byte-level BPE tokenizer is trained in-process with `tokenizers`, saved as a local
"org/model" directory and loaded through `transformers` -- no network.  The reference's own tests pin this
branch by comparing ids with the Hugging Face tokenizer (tests/test_tokenizer.py:55-63, 152-163); so do these,
on text where huToken's word splitter and GPT-2's regex agree (no apostrophes, single spaces, no line ends).
"""
import json
import os
import random

import pytest

import helpers as H

transformers = pytest.importorskip("transformers")
tokenizers = pytest.importorskip("tokenizers")

TEXTS = [
    "hu",
    "hello world",
    "the quick brown fox jumps over the lazy dog",
    "szia uram, hogy vagy? 123 meg 4567!",
    "árvíztűrő tükörfúrógép és a többiek",
    "tokens: 12 + 30 = 42 (mostly) ok",
    "漢字 and 😂 emoji",
    "",
]


def _corpus():
    rng = random.Random(5)
    words = ("hu hello world the quick brown fox jumps over lazy dog szia uram hogy vagy meg mostly ok tokens "
             "emoji and árvíztűrő tükörfúrógép és a többiek there this that with from have").split()
    return [" ".join(rng.choice(words) for _ in range(20)) + " 123 4567 , ? ! ( ) : + =" for _ in range(300)]


@pytest.fixture(scope="module")
def hf_dir(tmp_path_factory):
    root = tmp_path_factory.mktemp("hfroot")
    d = root / "org" / "tiny"
    d.mkdir(parents=True)
    bpe = tokenizers.ByteLevelBPETokenizer()
    bpe.train_from_iterator(_corpus(), vocab_size=600, min_frequency=1, show_progress=False)
    bpe.save_model(str(d))  # vocab.json + merges.txt
    json.dump({"tokenizer_class": "GPT2Tokenizer", "model_max_length": 1024},
              open(d / "tokenizer_config.json", "w"))
    return root


@pytest.fixture()
def exported(hf_dir, tmp_path, monkeypatch):
    from hutoken_amd import hf
    monkeypatch.chdir(hf_dir)
    monkeypatch.setenv("XDG_CACHE_HOME", str(tmp_path / "cache"))
    return hf.export("org/tiny")


def test_export_files(exported, tmp_path):
    ex = exported
    base = os.path.join(str(tmp_path / "cache"), "hutoken", "org", "tiny")
    assert ex["vocab_file"] == os.path.join(base, "tiny.txt")
    assert ex["special_chars_file"] == os.path.join(base, "tiny_special_chars.txt")
    assert ex["merges_file_path"] == os.path.join(base, "merges.txt") and os.path.isfile(ex["merges_file_path"])
    assert ex["is_byte_encoder"] == 1
    assert ex["prefix"] is None  # "hu" is one token of this vocabulary (hutoken.py:75-76)
    tok = ex["tokenizer"]
    lines = open(ex["vocab_file"]).read().splitlines()
    assert len(lines) == len(tok.vocab)
    # ascending ids, every key in the 0xHH form
    ids = [int(l.split(" == ")[1]) for l in lines]
    assert ids == sorted(ids)
    key0 = bytes(int(h, 16) for h in lines[300].split(" == ")[0].split("0x")[1:])
    assert tok.vocab[key0.decode("utf-8")] == ids[300]
    sp = dict(l.split(" == ") for l in open(ex["special_chars_file"], encoding="utf-8").read().splitlines())
    assert sp["32"] == "Ġ" and sp["10"] == "Ċ" and len(sp) == 68


def test_oracle_on_exported_files_equals_hf(exported):
    """the CPU restatement (and the compiled reference, when present) on the exported files == the HF tokenizer"""
    from oracle.oracle import Oracle
    ex = exported
    tok = ex["tokenizer"]
    o = Oracle(ex["vocab_file"], ex["special_chars_file"], ex["prefix"], bool(ex["is_byte_encoder"]),
               merges_path=ex["merges_file_path"])
    assert o.has_merges
    for t in TEXTS:
        assert o.encode(t.encode("utf-8")) == tok.encode(t), t
    try:
        from oracle import ref
        r = ref.RefTokenizer(ex["vocab_file"], ex["special_chars_file"], ex["prefix"], bool(ex["is_byte_encoder"]),
                             merges_path=ex["merges_file_path"])
    except Exception:
        return  # /root/reference is not on this machine
    for t in TEXTS:
        if t:
            assert r.encode(t) == tok.encode(t), t


def test_unknown_model_raises(hf_dir, tmp_path, monkeypatch):
    from hutoken_amd import hf
    monkeypatch.chdir(hf_dir)
    monkeypatch.setenv("XDG_CACHE_HOME", str(tmp_path / "cache"))
    monkeypatch.setenv("HF_HUB_OFFLINE", "1")
    with pytest.raises(ValueError, match="Could not download Hugging Face tokenizer"):
        hf.export("org/absent")


@pytest.mark.gpu
def test_initialize_from_hf_dir_equals_hf(hf_dir, tmp_path, monkeypatch):
    import hutoken_amd as hutoken
    monkeypatch.chdir(hf_dir)
    monkeypatch.setenv("XDG_CACHE_HOME", str(tmp_path / "cache"))
    assert hutoken.initialize("org/tiny") is None
    assert hutoken.context().uses_merges
    tok = transformers.AutoTokenizer.from_pretrained("org/tiny")
    for t in TEXTS:
        assert hutoken.encode(t) == tok.encode(t), t
    assert hutoken.batch_encode(TEXTS, num_threads=2) == [tok.encode(t) for t in TEXTS]
    for t in TEXTS:
        assert hutoken.decode(tok.encode(t)) == t
    rng = random.Random(9)
    docs = [" ".join(rng.choice(_corpus()[0].split()) for _ in range(rng.randint(1, 40))) for _ in range(200)]
    assert hutoken.batch_encode(docs) == [tok.encode(t) for t in docs]
