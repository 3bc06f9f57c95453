"""Host fuzz of the device word splitter (hutoken_amd/csrc/hutk_classify.h: the table-driven form,
the SWAR form and the exact form) against the oracle's sequential splitter."""
import os
import subprocess

import helpers as H


def test_swar_and_exact_splitter_match_the_oracle(tmp_path):
    exe = os.path.join(str(tmp_path), "classify_check")
    obj = os.path.join(str(tmp_path), "oracle.o")
    subprocess.check_call(["gcc", "-O2", "-c", os.path.join(H.ROOT, "oracle", "hutk_oracle.c"), "-o", obj])
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I" + os.path.join(H.ROOT, "hutoken_amd", "csrc"),
                           "-o", exe, os.path.join(H.ROOT, "tests", "cpu", "classify_check.cpp"), obj, "-lpthread"])
    for seed, mode in ((1, 0), (2, 1), (3, 1), (4, 2)):
        out = subprocess.run([exe, "30000", str(seed), str(mode)], capture_output=True, text=True)
        assert out.returncode == 0, out.stdout + out.stderr
        assert "mismatches 0" in out.stdout
