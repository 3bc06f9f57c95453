// hutk_kdev.h -- device-side helpers shared by the kernels of hutk_kernels.hip and hutk_ptiles.hip: table lookups,
// wavefront-level LDS ordering, the splitter restated per byte position, symbol widths, bitmap and scan helpers.
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#include "hutk_classify.h"
#include "hutk_device.h"

namespace hutk {

// ------------------------------------------------------------------------
// table lookups
// ------------------------------------------------------------------------
// Pair table (hutk_internal.h): one 16-byte bucket of two entries per lookup.  The pair is in its first bucket, or
// -- rarely, and announced by a filter bit in the first bucket -- in its second, or absent.
struct PairProbe { uint4 b; uint32_t t; };
__device__ __forceinline__ PairProbe pair_issue(const DevTables& T, uint32_t l, uint32_t r) {
    PairProbe p;
    p.t = pair_mix(l, r);
    p.b = T.pair_buckets[pair_bucket1(p.t, T.pair_shift)];
    return p;
}
__device__ __forceinline__ uint32_t pair_match(const uint4 b, uint32_t k0, uint32_t k1) {
    // w1 of the matching entry, or all ones
    // (bitwise on purpose: with && the compiler loads .x first and .y only on a match -- a second round trip)
    const uint32_t da = (b.x ^ k0) | ((b.y ^ k1) & 0xFFu), db = (b.z ^ k0) | ((b.w ^ k1) & 0xFFu);
    return da == 0 ? b.y : db == 0 ? b.w : 0xFFFFFFFFu;
}
__device__ __forceinline__ uint32_t pair_resolve(const DevTables& T, const PairProbe& p, uint32_t l, uint32_t r) {
    const uint32_t k0 = l | ((r & 0xFFFu) << 20), k1 = r >> 12;
    uint32_t y = pair_match(p.b, k0, k1);
    if (y == 0xFFFFFFFFu) {
        const uint32_t filter = (p.b.y >> 28) | ((p.b.w >> 28) << 4);
        if ((filter >> (p.t & 7u)) & 1u)  // some pair with this filter bit moved on to its second bucket
            y = pair_match(T.pair_buckets[pair_bucket2(p.t, T.pair_shift)], k0, k1);
    }
    const uint32_t m = (y >> 8) & 0xFFFFFu;  // PAIR_ABSENT for an empty entry matched by (SYM_UNK, SYM_UNK) and for y == all ones
    return m == PAIR_ABSENT ? SYM_NONE : m;
}
__device__ __forceinline__ uint32_t pair_lookup(const DevTables& T, uint32_t l, uint32_t r) {
    return pair_resolve(T, pair_issue(T, l, r), l, r);
}

__device__ __forceinline__ uint32_t char_lookup(const DevTables& T, uint32_t packed) {
    // two-choice cuckoo: both candidate slots loaded together, no dependent probe sequence
    const uint32_t h = char_hash(packed);
    const uint64_t a = T.char_slots[char_slot1(h, T.char_shift)], b = T.char_slots[char_slot2(h, T.char_mask)];
    const uint32_t ka = (uint32_t)(a >> 32) ^ packed, kb = (uint32_t)(b >> 32) ^ packed;  // (bitwise: see pair_match)
    return ka == 0 ? (uint32_t)a : kb == 0 ? (uint32_t)b : SYM_UNK;
}

// rank used for comparisons (smaller merges first; ties resolved by position)
__device__ __forceinline__ uint32_t rank_of(const DevTables& T, uint32_t merged) {
    if (T.rank_is_sym) return merged;
    return (uint32_t)T.sym_id[merged] ^ 0x80000000u;  // signed id order as unsigned
}

__device__ __forceinline__ int32_t sym_to_id(const DevTables& T, uint32_t s) {
    if (T.ident_ids) return s < T.n_vocab_sym ? (int32_t)s : -1;
    return s < T.n_sym ? T.sym_id[s] : -1;
}

// One wavefront per workgroup: LDS instructions of a wavefront execute in order, so lanes exchange data
// through LDS without s_barrier -- and without the "wait for every outstanding global load and STORE"
// that __syncthreads() implies.  This only stops the compiler from moving LDS accesses across the point.
// index of the lowest set bit; -1 (all ones) for 0 -- the hardware's own answer, which the merge loop's scan relies on
__device__ __forceinline__ int ffbl_raw(uint32_t x) {
    int r;
    asm("v_ffbl_b32 %0, %1" : "=v"(r) : "v"(x));
    return r;
}
// LDS byte address of a pointer into a __shared__ object (for the hand-issued reads below)
__device__ __forceinline__ uint32_t lds_addr(const void* p) {
    return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void*)p;
}
// Four 16-bit LDS reads IN FLIGHT TOGETHER.  Written out because the compiler, short of registers in k_tiles, gives the four
// reads of the merge loop's scan one destination register and waits for each before it issues the next (seen in the ISA:
// four LDS latencies per step where one would do).
__device__ __forceinline__ void lds_read4_u16(uint32_t a0, uint32_t a1, uint32_t a2, uint32_t a3, uint32_t& v0, uint32_t& v1,
                                              uint32_t& v2, uint32_t& v3) {
    asm volatile(
        "ds_read_u16 %0, %4\n\tds_read_u16 %1, %5\n\tds_read_u16 %2, %6\n\tds_read_u16 %3, %7\n\ts_waitcnt lgkmcnt(0)"
        : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3)
        : "v"(a0), "v"(a1), "v"(a2), "v"(a3)
        : "memory");
}
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Seam map, second level (Tables::seam2_*): 24 bits at bit offset s (0..127) of the 16 bytes hi:lo, and the verdict for the
// boundary between the three-byte characters a3 | b3 where the first level says "may join": true = no token spans it.
__device__ __forceinline__ uint32_t win24(uint64_t lo, uint64_t hi, int s) {
    const uint64_t r = s == 0 ? lo : s < 64 ? ((lo >> s) | (hi << (64 - s))) : (hi >> (s - 64));
    return (uint32_t)r & 0xFFFFFFu;
}
__device__ __forceinline__ bool seam2_cuts(const DevTables& T, uint32_t a3, uint32_t b3) {
    if (!seam2_char3(a3) || !seam2_char3(b3)) return false;
    uint32_t hit = (T.seam2_part[(a3 >> 16) & 0xFFu] >> (b3 & 31u)) & 1u;
    auto ask = [&](uint32_t ka, uint32_t kb) {
        if (!(T.seam2_cats & seam2_cat_bit(ka, kb))) return;
        const uint32_t h = seam2_hash(ka == 3 ? a3 : (a3 >> 16) & 0xFFu, kb == 3 ? b3 : kb == 2 ? (b3 & 0xFFFFu) : (b3 & 0xFFu), ka, kb) >> T.seam2_shift;
        hit |= (T.seam2_bits[h >> 5] >> (h & 31u)) & 1u;
    };
    ask(3, 3); ask(3, 2); ask(3, 1); ask(1, 3); ask(1, 2);
    return !hit;
}

// __syncthreads() of a workgroup that is ONE wavefront, without the s_barrier: the same fences, so that what one lane wrote
// (LDS or global memory) the others read afterwards -- for code that also runs as one of two independent wavefronts of a
// workgroup (k_exc_b), where an s_barrier would tie them together.
__device__ __forceinline__ void wave_wg_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

__device__ __forceinline__ void raise(int32_t* err, int32_t code) { atomicCAS(err, 0, code); }

// Is the word at byte ws (of the document that starts at ds) the one the prefix goes with (core.c:364-366, 421-451: the
// first word, or the first match of the regex pre-token path), and does its document begin with a space?
__device__ __forceinline__ bool gbit(const uint32_t* m, int64_t p) { return (m[p >> 5] >> (p & 31)) & 1u; }
__device__ __forceinline__ bool word_is_first(const BatchArgs& A, int64_t ws, int64_t ds) {
    return A.first_bits ? gbit(A.first_bits, ws) : ws == ds;
}
__device__ __forceinline__ bool doc_begins_with_space(const BatchArgs& A, int64_t ws) {  // (for a first word)
    return A.alone_bits ? gbit(A.alone_bits, ws) : A.bytes[ws] == ' ';
}

// ------------------------------------------------------------------------
// splitter, restated per byte position (src/parser.c:24-183)
// ------------------------------------------------------------------------
__device__ __forceinline__ bool is_cont(uint32_t b) { return (b & 0xC0u) == 0x80u; }
__device__ __forceinline__ bool bit_at(const uint32_t* m, int i) { return (m[i >> 5] >> (i & 31)) & 1u; }

__device__ __forceinline__ bool cp_alpha(uint32_t cp) {  // parser.c:102-129
    if ((cp | 0x20u) - 'a' < 26u) return true;
    switch (cp) {
        case 0xE1: case 0xE9: case 0xED: case 0xF3: case 0xFA: case 0x151: case 0x171:
        case 0xFC: case 0xF6: case 0xC1: case 0xC9: case 0xCD: case 0xD3: case 0xDA:
        case 0x150: case 0x170: case 0xDC: case 0xD6:
            return true;
        default:
            return false;
    }
}

// A lead byte at window index j: its class when the whole character is present
// inside the document (parser.c:144-183), else C_BAD.  A structurally complete
// sequence that decodes to U+0000 or to ASCII whitespace is never consumed by
// any of the splitter's runs, so each of its bytes is a word of its own: C_BAD.
__device__ __forceinline__ uint8_t lead_class(const uint8_t* sb, const uint32_t* docm, int j, int* len) {
    const uint32_t b0 = sb[j];
    int L;
    uint32_t cp;
    *len = 1;
    if ((b0 & 0xE0u) == 0xC0u) { L = 2; cp = b0 & 0x1Fu; }
    else if ((b0 & 0xF0u) == 0xE0u) { L = 3; cp = b0 & 0x0Fu; }
    else if ((b0 & 0xF8u) == 0xF0u) { L = 4; cp = b0 & 0x07u; }
    else return C_BAD;
    for (int k = 1; k < L; k++) {
        const uint32_t b = sb[j + k];
        if (!is_cont(b) || bit_at(docm, j + k)) return C_BAD;
        cp = (cp << 6) | (b & 0x3Fu);
    }
    if (cp == 0 || cp == 0x20u || (cp - 9u) < 5u) return C_BAD;
    *len = L;
    if (cp_alpha(cp)) return C_ALPHA;
    if (cp - '0' < 10u) return C_DIGIT;
    return C_OTHER;
}

__device__ __forceinline__ uint8_t code_at(const uint8_t* sb, const uint32_t* docm, int li) {
    const uint32_t b = sb[li];
    if (b < 0x80u) {
        if ((b | 0x20u) - 'a' < 26u) return C_ALPHA;
        if (b - '0' < 10u) return C_DIGIT;
        if (b == 0x20u) return C_SPACE;
        if (b - 9u < 5u) return C_WS;
        if (b == 0) return C_BAD;
        return C_OTHER;
    }
    int len;
    if (is_cont(b)) {
        for (int k = 1; k <= 3; k++) {
            const uint32_t bl = sb[li - k];
            if (is_cont(bl)) continue;
            if (bl >= 0xC0u && lead_class(sb, docm, li - k, &len) != C_BAD && len > k) return C_INTERIOR;
            break;
        }
        return C_BAD;
    }
    return lead_class(sb, docm, li, &len);
}

// does a word start at window index li?  (scode = code_at of every index)
__device__ __forceinline__ bool word_starts(const uint8_t* scode, const uint32_t* docm, int li) {
    const uint8_t c = scode[li];
    if (c == C_INTERIOR) return false;
    if (bit_at(docm, li)) return true;
    int pj = li - 1;
    while (scode[pj] == C_INTERIOR) pj--;  // at most 3 steps: a lead precedes interior bytes
    const uint8_t pc = scode[pj];
    if (c >= C_WS || pc >= C_WS) return true;      // whitespace and stray bytes stand alone
    if (c == C_SPACE) return pc != C_SPACE;        // a run of spaces starts after a non-space
    if (pc == c) return false;                     // same class: the run continues
    if (pc == C_SPACE)                             // "[ ]?" prefix: ONE space attaches forward
        return !(bit_at(docm, pj) || scode[pj - 1] != C_SPACE);
    return true;
}

// ------------------------------------------------------------------------
// symbol storage in LDS: 16-bit when the vocabulary has fewer than 65520 symbols
// (halves the LDS footprint of the tile kernel -> more resident wavefronts)
// ------------------------------------------------------------------------
// With fewer than 0xFFF0 symbols the 16-bit form of a symbol is just its low half:
// SYM_NONE -> 0xFFFF ("no rank" in the pair array), SYM_UNK -> 0xFFFF (a unit that is no
// symbol: no table key has it, and it maps to id -1), so narrow/widen are plain casts.
template <typename SymT> struct Sym;
template <> struct Sym<uint32_t> {
    static __device__ __forceinline__ uint32_t narrow(uint32_t v) { return v; }
    static __device__ __forceinline__ uint32_t widen(uint32_t v) { return v; }
    static constexpr uint32_t NONE = SYM_NONE;
    typedef uint2 Pair;  // entry of the (byte, next byte) table: {symbol of the byte, merged symbol}
    static __device__ __forceinline__ uint32_t pair_sym(uint2 e) { return e.x; }
    static __device__ __forceinline__ uint32_t pair_merged(uint2 e) { return e.y; }
};
template <> struct Sym<uint16_t> {
    static __device__ __forceinline__ uint16_t narrow(uint32_t v) { return (uint16_t)v; }
    static __device__ __forceinline__ uint32_t widen(uint16_t v) { return (uint32_t)v; }
    static constexpr uint16_t NONE = 0xFFFFu;
    typedef uint32_t Pair;
    static __device__ __forceinline__ uint16_t pair_sym(uint32_t e) { return (uint16_t)e; }
    static __device__ __forceinline__ uint16_t pair_merged(uint32_t e) { return (uint16_t)(e >> 16); }
};

// 64 bits of a bitmap starting at bit `start` (the bitmap has 2 words of slack)
__device__ __forceinline__ uint64_t bits64(const uint32_t* m, int start) {
    const int k = start >> 5, sh = start & 31;
    uint64_t v = ((uint64_t)m[k] | ((uint64_t)m[k + 1] << 32)) >> sh;
    if (sh) v |= (uint64_t)m[k + 2] << (64 - sh);
    return v;
}

// byte k (0..31) of a 32-byte register window
struct Win { uint64_t a, b, c, d; };
__device__ __forceinline__ uint32_t win_byte(const Win w, int k) {
    const uint64_t v = (k < 16) ? ((k < 8) ? w.a : w.b) : ((k < 24) ? w.c : w.d);
    return (uint32_t)(v >> ((k & 7) * 8)) & 0xFFu;
}

// exclusive prefix sum over the 64 lanes with DPP row shifts and broadcasts (12 VALU instructions, no LDS)
__device__ __forceinline__ uint32_t wave_excl_scan(uint32_t v, int lane, uint32_t* total) {
    (void)lane;
    uint32_t inc = v;
    // inclusive scan inside each row of 16 lanes: row_shr:1, 2, 4, 8 (lanes shifted in from outside read 0)
    inc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x111, 0xf, 0xf, false);
    inc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x112, 0xf, 0xf, false);
    inc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x114, 0xf, 0xf, false);
    inc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x118, 0xf, 0xf, false);
    // row_bcast:15 into rows 1 and 3, then row_bcast:31 into rows 2 and 3
    inc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x142, 0xa, 0xf, false);
    inc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x143, 0xc, 0xf, false);
    *total = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
    return inc - v;
}

}  // namespace hutk
