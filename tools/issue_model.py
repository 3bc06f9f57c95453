#!/usr/bin/env python3
"""What a VALU instruction of k_tiles costs a gfx950 SIMD: the static opcode mix of the shipped kernel
(tools/isa_histogram.py) priced with the measured per-opcode issue costs (tools/valu_issue_bench.hip ->
profiles/r03_valu_issue_costs.json).

The microbenchmark shows two classes at two and more wavefronts per SIMD: a few simple VOP1/VOP2 opcodes (mov, not, and, or,
xor, add, sub, lshrrev, ashrrev, bitop3, f32 add/fma) take 2 cycles per wave-instruction, everything else (lshlrev, min/max,
mul, every other VOP3, compares, selects, DPP, SDWA, lane reads/writes, 64-bit forms) takes 4 -- and a stream that ALTERNATES
the two classes runs at ~3.8 per instruction, i.e. the 2-cycle rate needs two wavefronts that both have a simple opcode
next.  So a kernel's VALU cost per instruction lies between
    paired   = (2 * n_fast + 4 * n_slow) / n     every simple opcode finds a partner
    unpaired = 4                                 none does
and bench.py reports the issue fraction for both bounds.  usage: issue_model.py [costs.json] > profiles/r03_issue_model.json"""
import json, os, re, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
costs = json.load(open(sys.argv[1] if len(sys.argv) > 1 else os.path.join(root, "profiles", "r03_valu_issue_costs.json")))
table = costs["cycles_per_inst_per_simd_8_waves"]
# microbenchmark name -> ISA base opcodes it stands for
FAST = {"v_mov_b32": "mov", "v_not_b32": "not", "v_xor_b32": "xor", "v_and_b32": "and", "v_or_b32": "or", "v_add_u32": "add",
        "v_sub_u32": "sub", "v_subrev_u32": "sub", "v_lshrrev_b32": "lshr", "v_ashrrev_i32": "ashr", "v_bitop3_b32": "bitop3",
        "v_add_f32": "addf", "v_fma_f32": "fmaf", "v_accvgpr_read_b32": "mov", "v_accvgpr_write_b32": "mov"}
hist = json.loads(subprocess.check_output([sys.executable, os.path.join(root, "tools", "isa_histogram.py")] + sys.argv[2:]))
n_fast = n_slow = 0
by = {}
for op, n in hist.items():
    if not op.startswith("v_"):
        continue
    base = re.sub(r"_(e32|e64|sdwa|dpp|e64_dpp)$", "", op)
    plain = op.endswith("_e32") or op.endswith("_e64") or op == base
    if base in FAST and plain and table.get(FAST[base], 9) < 3.0:
        n_fast += n
        by[op] = ["fast", n]
    else:
        n_slow += n
        by[op] = ["slow", n]
n = n_fast + n_slow
out = {"kernel": "k_tiles<uint16_t, true, true, 4> (static mix)", "valu_static": n, "fast": n_fast, "slow": n_slow,
       "salu_static": sum(v for k, v in hist.items() if k.startswith("s_")),
       "cycles_per_valu_inst_paired": round((2.0 * n_fast + 4.0 * n_slow) / n, 3), "cycles_per_valu_inst_unpaired": 4.0,
       "measured_mix_fast_slow_alternating": table.get("mix_fast_slow"), "measured_fast": table.get("xor"),
       "measured_slow": table.get("lshl"), "source_costs": "profiles/r03_valu_issue_costs.json (tools/valu_issue_bench.hip)",
       "classes": by}
print(json.dumps(out, indent=1))
