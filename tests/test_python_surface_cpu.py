"""Python surface of the drop-in module: signatures, exception classes and the messages the
reference pins (reference hutoken.py:22-43, 122-139; tests/test_tokenizer.py:43-46, 137-141)."""
import importlib
import inspect
import os

import pytest

import helpers as H


@pytest.fixture()
def hutoken():
    import hutoken_amd
    m = importlib.reload(hutoken_amd)  # fresh, uninitialised module state
    return m


def test_signatures(hutoken):
    assert list(inspect.signature(hutoken.encode).parameters) == ["text"]
    sig = inspect.signature(hutoken.batch_encode)
    assert list(sig.parameters) == ["texts", "num_threads"] and sig.parameters["num_threads"].default == 1
    sig = inspect.signature(hutoken.initialize)
    assert list(sig.parameters) == ["model_or_path", "args", "kwargs"]


def test_encode_before_initialize(hutoken):
    msg = "Vocabulary is not initialized for encoding. Call 'initialize_encode' function first."
    with pytest.raises(RuntimeError, match=msg):
        hutoken.encode("szia")
    with pytest.raises(RuntimeError, match=msg):
        hutoken.batch_encode(["szia"])


def test_initialize_errors(hutoken, tmp_path):
    vp = os.path.join(str(tmp_path), "v.txt")
    open(vp, "w").write("invalid_line_format\n")
    sp = os.path.join(str(tmp_path), "s.txt")
    open(sp, "w").write("32 == x\n")
    with pytest.raises(ValueError, match="Invalid format in vocab file."):
        hutoken.initialize(vp, sp)
    with pytest.raises(ValueError, match="does not exist"):
        hutoken.initialize(vp, os.path.join(str(tmp_path), "missing.txt"))
    with pytest.raises(TypeError, match="Invalid arguments. Expected a string"):
        hutoken.initialize(vp)  # special file is a required str (lib.c:203 "ss|zpizz")
    with pytest.raises(ValueError, match="Could not download Hugging Face tokenizer"):
        hutoken.initialize("NYTK/PULI-LlumiX-32K")


def test_merges_file_arguments(hutoken, tmp_path):
    """hutoken.py:30-37: a merges file given positionally must exist (and is then dropped by the reference's
    local-file branch); the keyword of the native initialize selects the id-keyed path here."""
    ents, sp = H.random_byte_vocab(1, n_merges=20)
    vp, spath = H.write_vocab(tmp_path, "v", ents, sp)
    missing = str(tmp_path / "no_merges.txt")
    with pytest.raises(ValueError, match="The provided merges file"):
        hutoken.initialize(vp, spath, None, None, None, None, None, missing)  # args[6] of *args
    with pytest.raises(ValueError, match="The provided merges file"):
        hutoken.initialize(vp, spath, merges_file_path=missing)


def test_out_of_path_features_fail_loudly(hutoken, tmp_path):
    ents, sp = H.random_byte_vocab(1, n_merges=20)
    vp, spath = H.write_vocab(tmp_path, "v", ents, sp)
    with pytest.raises(RuntimeError):  # (no GPU here: the context cannot be created)
        hutoken.initialize(vp, spath, pattern="[a-z]+")
    with pytest.raises(RuntimeError):
        hutoken.decode([1, 2, 3])


def test_pattern_argument(tmp_path):
    """initialize(pattern=...): a POSIX ERE (core.c:350-360).  One that does not compile is refused (the reference
    goes on with an uncompiled pattern for most error codes).  A pattern together with a prefix is fine (core.c:362-366)."""
    from hutoken_amd import _capi
    ents, sp = H.random_byte_vocab(1, n_merges=20)
    vp, spath = H.write_vocab(tmp_path, "v", ents, sp)
    ctx = _capi.Context(vp, spath, None, True, device=-2)
    ctx.set_pattern("[ ]?[a-z]+|[ ]+")
    ctx.set_pattern(None)
    for bad in ("(", "[a-", "a{2,1}", ""):
        with pytest.raises(ValueError, match="Regex could not be compiled."):
            ctx.set_pattern(bad)
    ctx = _capi.Context(vp, spath, "x", True, device=-2)
    ctx.set_pattern("[a-z]+")
