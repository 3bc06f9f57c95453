#!/usr/bin/env python3
"""A/B of ENVIRONMENT settings of one library on one GPU box.
usage: ab_env.py CORPUS N_DOCS VOCAB ROUNDS "VAR=a" "VAR=b VAR2=c" ...  ("-" = nothing set) -> best and median GB/s each."""
import os, re, subprocess, sys, statistics
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
corpus, n_docs, vocab, rounds = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
sets = sys.argv[5:]
res = {s: [] for s in sets}
for r in range(rounds):
    for s in sets:
        env = dict(os.environ)
        if s != "-":
            env.update(kv.split("=", 1) for kv in s.split())
        out = subprocess.run([sys.executable, os.path.join(root, "tools", "profile_phases.py"), corpus, n_docs, vocab],
                             env=env, capture_output=True, text=True).stdout
        m = re.search(r"([0-9.]+) GB/s", out)
        res[s].append(float(m.group(1)) if m else float("nan"))
for s in sets:
    v = res[s]
    print(f"{s}: best {max(v):.2f} median {statistics.median(v):.2f} GB/s  {v}")
