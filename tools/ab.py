#!/usr/bin/env python3
"""A/B of library builds on ONE GPU box (boxes differ by a few per cent, so numbers from two gpurun calls
do not compare).  usage: ab.py CORPUS N_DOCS VOCAB ROUNDS lib1.so lib2.so ...  -> best and median GB/s each."""
import os, re, subprocess, sys, statistics
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
corpus, n_docs, vocab, rounds = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
libs = sys.argv[5:]
res = {l: [] for l in libs}
for r in range(rounds):
    for l in libs:
        env = dict(os.environ, HUTOKEN_AMD_LIB=os.path.join(root, l))
        out = subprocess.run([sys.executable, os.path.join(root, "tools", "profile_phases.py"), corpus, n_docs, vocab],
                             env=env, capture_output=True, text=True).stdout
        m = re.search(r"([0-9.]+) GB/s", out)
        res[l].append(float(m.group(1)) if m else float("nan"))
for l in libs:
    v = res[l]
    print(f"{l}: best {max(v):.2f} median {statistics.median(v):.2f} GB/s  {v}")
