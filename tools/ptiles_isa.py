#!/usr/bin/env python3
"""Static instruction counts of k_ptiles between the source's PT_MARK comments (loops count once).
usage: ptiles_isa.py [extra hipcc flags]"""
import os, re, subprocess, sys, collections
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "hutoken_amd", "csrc", "hutk_ptiles.hip")
out = "/tmp/ptiles_isa.s"
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(root, "include"),
                       "-I" + os.path.dirname(src), "-S", "--cuda-device-only", "-DHUTK_PT_MARKS=1", "-o", out, src] + sys.argv[1:], stderr=subprocess.DEVNULL)
lines = open(out).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_ZN4hutk8k_ptiles"))
cur = "prologue"
acc = collections.OrderedDict()
for l in lines[start:]:
    t = l.strip()
    m = re.match(r"; PTMARK (\w+)", t)
    if m:
        cur = m.group(1)
        continue
    if t.startswith("s_endpgm"):
        break
    a = acc.setdefault(cur, [0, 0, 0, 0, 0])
    if t.startswith("v_"):
        a[0] += 1
        if "readlane" in t or "writelane" in t:
            a[4] += 1
    elif t.startswith("s_"):
        a[1] += 1
    elif t.startswith("ds_"):
        a[2] += 1
    elif t.startswith(("global_", "scratch_", "flat_", "buffer_")):
        a[3] += 1
print(f"{'region':14s} {'VALU':>6s} {'SALU':>6s} {'LDS':>5s} {'VMEM':>5s} {'lane-spill':>10s}")
for k, a in acc.items():
    print(f"{k:14s} {a[0]:6d} {a[1]:6d} {a[2]:5d} {a[3]:5d} {a[4]:10d}")
