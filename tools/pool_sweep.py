#!/usr/bin/env python3
"""the unit limit of k_tiles' pool (words of more units go to the exception kernels instead), per-kernel
times from the library's own events and a kernel trace.  usage (GPU): pool_sweep.py CORPUS N_DOCS VOCAB lib.so ..."""
import os, re, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
corpus, n_docs, vocab = sys.argv[1:4]
for lib in sys.argv[4:]:
    env = dict(os.environ, HUTOKEN_AMD_LIB=os.path.join(root, lib), HUTK_PTILES="0")
    best = None
    for _ in range(3):
        out = subprocess.run([sys.executable, os.path.join(root, "tools", "profile_phases.py"), corpus, n_docs, vocab], env=env, capture_output=True, text=True).stdout
        m = re.search(r"([0-9.]+) ms/step\s+([0-9.]+) GB/s\s+tile-kernel ([0-9.]+) ms\s+ids (\d+)", out)
        if m and (best is None or float(m.group(1)) < best[0]):
            best = (float(m.group(1)), float(m.group(2)), float(m.group(3)), int(m.group(4)))
    print(f"{os.path.basename(lib):28s} {corpus} {n_docs} {vocab}: step {best[0]:.3f} ms  {best[1]:.1f} GB/s  k_tiles {best[2]:.3f} ms  rest {best[0]-best[2]:.3f} ms  ids {best[3]}", flush=True)
