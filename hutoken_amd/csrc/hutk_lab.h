// hutk_lab.h -- MEASUREMENT ONLY switches of hutk_kernels.hip.  None is set in a build that ships; tools/build_variant.sh
// sets one per library for the A/B runs (tools/ab.py, tools/pmc_ab.sh) whose numbers DESIGN.md section 5 quotes.
//   HUTK_PERTURB_VALU=n   n extra VALU instructions per tile (a dependent chain in every lane)
//   HUTK_PERTURB_SLEEP=n  n x ~8 k idle cycles per wavefront before the merge phase
//   HUTK_PERTURB_MEM=n    n extra 16-byte table gathers per word
//   HUTK_ABLATE_MERGE=1   no word is merged (WRONG IDS): instruction count of the other phases
//   HUTK_MERGE_STAMPS=1   the ten clock stamps of the diagnostic profile (hutk_debug_profile) are spent inside the
//                         merge phase instead of at the phase boundaries (tools/profile_phases.py)
#pragma once
#ifndef HUTK_PERTURB_VALU
#define HUTK_PERTURB_VALU 0
#endif
#ifndef HUTK_PERTURB_MEM
#define HUTK_PERTURB_MEM 0
#endif
#ifndef HUTK_PERTURB_SLEEP
#define HUTK_PERTURB_SLEEP 0
#endif
#ifndef HUTK_ABLATE_MERGE
#define HUTK_ABLATE_MERGE 0
#endif
#ifndef HUTK_MERGE_STAMPS
#define HUTK_MERGE_STAMPS 0
#endif
#ifndef HUTK_LAB_NO_CUTFLAGS
#define HUTK_LAB_NO_CUTFLAGS 0  // 1: k_tiles does not look for tiles without a word start (k_cut then never cuts: WRONG for over-long words)
#endif
#ifndef HUTK_LAB_ALIGN
#define HUTK_LAB_ALIGN 0  // n: the round loop and the trip loop of k_tiles start on a 2^n-byte boundary (is a few per cent of difference between two builds code placement?)
#endif
#ifndef HUTK_LAB_SLIM_ROUNDS
#define HUTK_LAB_SLIM_ROUNDS 0  // 1: words of up to 14 bytes in rounds of their own (unaligned LDS reads, no branches: 40 % fewer instructions in phase 5; measured: no faster, the merge phase is what a workgroup lives for -- profiles/r04_slim_rounds_ab.txt)
#endif
#ifndef HUTK_LAB_POOL_UNITS
#define HUTK_LAB_POOL_UNITS 32  // = LANE_MAX_UNITS; lower: words of more units leave k_tiles' pool for the exception kernels (profiles/r04_pool_unit_sweep.txt)
#endif
#ifndef HUTK_LAB_NO_COLD
#define HUTK_LAB_NO_COLD 0  // 1: MEASUREMENT ONLY (wrong for overlong encodings and over-long words): k_tiles without its two out-of-line calls, i.e. without scratch memory -- what does declaring scratch cost a launch?
#endif
#ifndef HUTK_LAB_EXC_FAST
#define HUTK_LAB_EXC_FAST 1  // 0: words of up to 1024 units merge with round 3's bpe_wave / bpe_wave_big (A/B of bpe_wave_fast)
#endif
#ifndef HUTK_LAB_EXC_GROUP
#define HUTK_LAB_EXC_GROUP 1  // 0: words of 65..256 units one LANE per word (d_exc_lane_fast<2>, <4>: round 3) instead of four / eight lanes
#endif
#ifndef HUTK_LAB_EXC_STAMPS
#define HUTK_LAB_EXC_STAMPS 0  // 1: d_exc_group_fast<2> leaves its trips' cycle sums in the profile buffer (tools/exc_stamps.py)
#endif
#ifndef HUTK_LAB_SEAM2_TILES
#define HUTK_LAB_SEAM2_TILES 0  // 1: the seam map's second level (whole characters) in k_tiles too, not in k_ptiles only: C3 226.0 -> 223.4 GB/s (8 rounds, one box), and text dense in three-byte characters is k_ptiles' anyway
#endif
#define HUTK_STR2(x) #x
#define HUTK_STR(x) HUTK_STR2(x)
#ifndef HUTK_LAB_LDS_PAD
#define HUTK_LAB_LDS_PAD 0  // bytes of unused LDS per workgroup of k_tiles: fewer resident workgroups (is the kernel bound by latency or by issue?)
#endif
#define HUTK_STAMP_AT(k)                                                       \
    do {                                                                       \
        if (W.prof && lane == 0) W.prof[tile * N_PHASE + (k)] = clock64();     \
    } while (0)
#if HUTK_MERGE_STAMPS
#define HUTK_STAMP(k) do {} while (0)
#define HUTK_MSTAMP(k) do { if (tile_ok) HUTK_STAMP_AT(k); } while (0)
#else
#define HUTK_STAMP(k) HUTK_STAMP_AT(k)
#define HUTK_MSTAMP(k) do {} while (0)
#endif
