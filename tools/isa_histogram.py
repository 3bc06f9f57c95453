#!/usr/bin/env python3
"""Static opcode histogram of one kernel of hutk_kernels.hip (device assembly from `hipcc -S --cuda-device-only`).
usage: isa_histogram.py [mangled-name-substring]   default: the byte-mode k_tiles instantiation the benchmark runs.
Prints JSON {opcode: count}; bench.py's issue model weights tools/valu_issue_bench's per-opcode costs with it."""
import collections, json, os, re, subprocess, sys, tempfile
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
csrc = os.path.join(root, "hutoken_amd", "csrc")
want = sys.argv[1] if len(sys.argv) > 1 else "k_tilesItLb1ELb1ELi4ELb0EEE"
out = os.path.join(tempfile.gettempdir(), "hutk_kernels_%d.s" % os.getuid())
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(root, "include"), "-I" + csrc,
                       "--cuda-device-only", "-S", os.path.join(csrc, "hutk_kernels.hip"), "-o", out] + sys.argv[2:],
                      stderr=subprocess.DEVNULL)
c = collections.Counter()
inside = False
for l in open(out):
    s = l.strip()
    if not inside:
        inside = s.startswith("_ZN") and want in s.split(":")[0] and ":" in s
        continue
    if s.startswith(".Lfunc_end"):
        break
    m = re.match(r"^((?:v|s|ds|global|buffer|scratch|flat)_[a-z0-9_]+)\b", s)
    if m:
        c[m.group(1)] += 1
print(json.dumps(dict(c.most_common()), indent=0))
