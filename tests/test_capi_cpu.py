"""The C-ABI library: builds, loads, exports every symbol include/hutoken_amd.h declares,
and fails loudly (never computes) without a GPU.  No compute calls here."""
import os
import re

import pytest

import helpers as H
from hutoken_amd import _capi


def test_library_exports_every_declared_symbol():
    lib = _capi.load()
    header = open(os.path.join(H.ROOT, "include", "hutoken_amd.h")).read()
    declared = sorted(set(re.findall(r"\b(hutk_[a-z0-9_]+)\s*\(", header)))
    assert declared, "no declarations found"
    assert sorted(_capi.EXPORTS) == declared
    for name in declared:
        assert hasattr(lib, name), name


def test_header_cites_reference_interfaces():
    header = open(os.path.join(H.ROOT, "include", "hutoken_amd.h")).read()
    for cite in ("src/lib.c:185-571", "src/lib.c:779-794", "src/lib.c:668-720", "include/hutoken/core.h:11"):
        assert cite in header


def test_no_gpu_means_loud_failure(tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    ents, sp = H.random_byte_vocab(1, n_merges=50)
    vp, spath = H.write_vocab(tmp_path, "v", ents, sp)
    with pytest.raises(RuntimeError, match="no HIP device"):
        _capi.Context(vp, spath, None, True)
    host = _capi.Context(vp, spath, None, True, device=-2)  # tables only
    import numpy as np
    with pytest.raises(RuntimeError, match="host-only"):
        host.encode_packed(np.frombuffer(b"abc", dtype=np.uint8), np.array([0, 3], dtype=np.int64))
    with pytest.raises(RuntimeError, match="host-only"):  # no device to add a second one to
        host.add_device(0)
    assert host.device_count == 0
    with pytest.raises(RuntimeError, match="no HIP device"):
        _capi.Context(vp, spath, None, True, devices=[0, 1])


def test_product_never_imports_the_oracle():
    pkg = os.path.join(H.ROOT, "hutoken_amd")
    for dirpath, _dirs, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", ".c")):
                src = open(os.path.join(dirpath, f), errors="replace").read()
                assert "import oracle" not in src and "from oracle" not in src, f
                assert "hutk_oracle" not in src and "hto_" not in src, f
