#!/usr/bin/env python3
"""Tile-kernel time of several library builds on ONE box (k_ptiles via HUTK_PTILES=1 unless the name says 'old').
usage: ptiles_ab.py CORPUS N_DOCS ROUNDS lib1.so lib2.so ..."""
import os, re, subprocess, sys, statistics
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
corpus, n_docs, rounds = sys.argv[1], sys.argv[2], int(sys.argv[3])
libs = sys.argv[4:]
res = {l: [] for l in libs}
for r in range(rounds):
    for l in libs:
        old = l.startswith("old:")
        path = l[4:] if old else l
        env = dict(os.environ, HUTOKEN_AMD_LIB=os.path.join(root, path), HUTK_PTILES="0" if old else "1")
        out = subprocess.run([sys.executable, os.path.join(root, "tools", "profile_phases.py"), corpus, n_docs, "VG"],
                             env=env, capture_output=True, text=True).stdout
        m = re.search(r"([0-9.]+) GB/s  tile-kernel ([0-9.]+) ms", out)
        res[l].append((float(m.group(1)), float(m.group(2))) if m else (float("nan"), float("nan")))
for l in libs:
    v = res[l]
    print(f"{l}: best {max(x[0] for x in v):.2f} GB/s, tile kernel min {min(x[1] for x in v):.3f} median {statistics.median(x[1] for x in v):.3f} ms")
