// hutk_classify.h -- the reference's word splitter (src/parser.c:24-183) restated
// for 16 consecutive positions at a time, in three forms:
//
//   classify16_dfa    a 31-state automaton over byte classes, one table lookup per byte (the form k_tiles
//                     uses: ~160 integer instructions + 46 LDS reads per 16 positions)
//   classify16        byte-parallel (SWAR) mask algebra, ~650 integer ops per 16 positions (k_tiles with
//                     -DHUTK_SPLIT_SWAR=1; kept as the independent second implementation the fuzz compares)
//   classify16_exact  per-position decode; handles the overlong encodings the other two
//                     defer (they report them through *exotic)
//
// Compiled for the device (k_tiles) AND for the host (tests/cpu/classify_check.cpp
// fuzzes all three against the oracle's sequential splitter), so it is plain integer C++.
//
// Input: a 32-byte window d[0..7] (little-endian dwords; window byte k is the text
// byte at position p0 - 8 + k, zero outside the data) and `dbits`, bit k set when a
// document starts at window byte k.  Output: bit j set when a word starts at
// position p0 + j, j = 0..15 (window bytes 8..23).
//
// The splitter is a local function of the bytes:
//   * a byte >= 0x80 is part of a character iff a structurally complete lead
//     (all continuation bytes present, inside the same document) covers it;
//     every other byte >= 0x80 is a one-byte word (class X), and so is each byte of
//     a sequence that decodes to U+0000 or to ASCII whitespace (parser.c:94: cp == 0
//     stops every run; whitespace is in no run, and only a raw ' ' starts a space run)
//   * classes: A letters (ASCII + 18 Hungarian code points, parser.c:102-129),
//     D digits, S space, W \t\n\v\f\r (each its own word, parser.c:84-87), O the rest
//   * a character starts a word iff: document start; or it or its predecessor is W/X;
//     or S after non-S; or A/D/O after a different class, unless that predecessor is a
//     LONE space (the "[ ]?" prefix of parser.c:34-36)
#pragma once
#include <cstdint>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define HUTK_CLS_HD __host__ __device__ __forceinline__
#define HUTK_CLS_UNROLL _Pragma("unroll")
#else
#define HUTK_CLS_HD inline
#define HUTK_CLS_UNROLL
#endif

namespace hutk {

enum : uint32_t { K_INTERIOR = 0, K_ALPHA = 1, K_DIGIT = 2, K_OTHER = 3, K_SPACE = 4, K_WS = 5, K_BAD = 6 };

// ---------------------------------------------------------------------------
// exact per-position form
// ---------------------------------------------------------------------------
HUTK_CLS_HD uint32_t cls_win_byte(const uint32_t (&d)[8], int k) {
#if defined(__HIP_DEVICE_COMPILE__)
    // select chain instead of a dynamically indexed register array (which would go to scratch)
    const int q = k >> 2;
    const uint32_t lo = (q & 2) ? ((q & 1) ? d[3] : d[2]) : ((q & 1) ? d[1] : d[0]);
    const uint32_t hi = (q & 2) ? ((q & 1) ? d[7] : d[6]) : ((q & 1) ? d[5] : d[4]);
    return (((q & 4) ? hi : lo) >> (8 * (k & 3))) & 0xFFu;
#else
    return (d[k >> 2] >> (8 * (k & 3))) & 0xFFu;
#endif
}

HUTK_CLS_HD bool cls_alpha_cp(uint32_t cp) {  // parser.c:102-129
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

HUTK_CLS_HD uint32_t cls_ascii(uint32_t b) {
    if ((b | 0x20u) - 'a' < 26u) return K_ALPHA;
    if (b - '0' < 10u) return K_DIGIT;
    if (b == 0x20u) return K_SPACE;
    if (b - 9u < 5u) return K_WS;
    if (b == 0) return K_BAD;
    return K_OTHER;
}

// lead at window byte k: class when the whole character is present inside the document
// (parser.c:144-183), else K_BAD; *len = its length (1 when bad)
HUTK_CLS_HD uint32_t cls_lead(const uint32_t (&d)[8], uint32_t dbits, int k, int* len) {
    const uint32_t b0 = cls_win_byte(d, k);
    int L;
    uint32_t cp;
    *len = 1;
    if ((b0 & 0xE0u) == 0xC0u) { L = 2; cp = b0 & 0x1Fu; }
    else if ((b0 & 0xF0u) == 0xE0u) { L = 3; cp = b0 & 0x0Fu; }
    else if ((b0 & 0xF8u) == 0xF0u) { L = 4; cp = b0 & 0x07u; }
    else return K_BAD;
    for (int q = 1; q < L; q++) {
        const uint32_t b = cls_win_byte(d, k + q);
        if ((b & 0xC0u) != 0x80u || ((dbits >> (k + q)) & 1u)) return K_BAD;
        cp = (cp << 6) | (b & 0x3Fu);
    }
    if (cp == 0 || cp == 0x20u || (cp - 9u) < 5u) return K_BAD;
    *len = L;
    if (cls_alpha_cp(cp)) return K_ALPHA;
    if (cp - '0' < 10u) return K_DIGIT;
    return K_OTHER;
}

HUTK_CLS_HD uint32_t cls_code(const uint32_t (&d)[8], uint32_t dbits, int k) {  // 3 <= k <= 23
    const uint32_t x = cls_win_byte(d, k);
    if (x < 0x80u) return cls_ascii(x);
    int len;
    if ((x & 0xC0u) == 0x80u) {
        for (int q = 1; q <= 3; q++) {
            const uint32_t bl = cls_win_byte(d, k - q);
            if ((bl & 0xC0u) == 0x80u) continue;
            if (bl >= 0xC0u && cls_lead(d, dbits, k - q, &len) != K_BAD && len > q) return K_INTERIOR;
            break;
        }
        return K_BAD;
    }
    return cls_lead(d, dbits, k, &len);
}

HUTK_CLS_HD uint32_t classify16_exact(const uint32_t (&d)[8], uint32_t dbits) {
    uint64_t lo = 0, hi = 0;  // 4-bit codes of window bytes 3..23
    for (int k = 3; k <= 23; k++) {
        const uint64_t c = cls_code(d, dbits, k);
        if (k < 16) lo |= c << (4 * k); else hi |= c << (4 * (k - 16));
    }
    auto code = [&](int k) -> uint32_t { return (uint32_t)((k < 16 ? lo >> (4 * k) : hi >> (4 * (k - 16))) & 7u); };
    uint32_t flags = 0;
    for (int j = 0; j < 16; j++) {
        const int k = 8 + j;
        const uint32_t c = code(k);
        bool f;
        if (c == K_INTERIOR) f = false;
        else if ((dbits >> k) & 1u) f = true;
        else {
            int pk = k - 1;
            while (code(pk) == K_INTERIOR) pk--;  // at most 3 steps
            const uint32_t pc = code(pk);
            if (c >= K_WS || pc >= K_WS) f = true;
            else if (c == K_SPACE) f = pc != K_SPACE;
            else if (pc == c) f = false;
            else if (pc == K_SPACE) f = !(((dbits >> pk) & 1u) || code(pk - 1) != K_SPACE);
            else f = true;
        }
        flags |= (uint32_t)f << j;
    }
    return flags;
}

// ---------------------------------------------------------------------------
// byte-parallel form.  Every per-byte predicate lives in bit 7 of its byte; a shift
// by one position is a funnel shift by 8 bits across neighbouring dwords.
// ---------------------------------------------------------------------------
// ({hi,lo} >> s) truncated to 32 bits, 0 < s < 32
HUTK_CLS_HD uint32_t funnel_r(uint32_t hi, uint32_t lo, int s) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_alignbit(hi, lo, s);
#else
    return (uint32_t)((((uint64_t)hi << 32) | lo) >> s);
#endif
}
// predicate of position k + n seen at k (n = 1..3): needs the following dword
HUTK_CLS_HD uint32_t nxt(uint32_t following, uint32_t cur, int n) { return funnel_r(following, cur, 8 * n); }
// predicate of position k - n seen at k (n = 1..3): needs the preceding dword
HUTK_CLS_HD uint32_t prv(uint32_t cur, uint32_t preceding, int n) { return funnel_r(cur, preceding, 32 - 8 * n); }
// bit j of the result = bit 7 of byte j of m
HUTK_CLS_HD uint32_t movemask4(uint32_t m) { return ((((m >> 7) & 0x01010101u) * 0x00204081u) >> 21) & 0xFu; }
// 4 bits -> hi-bit mask
HUTK_CLS_HD uint32_t spread4(uint32_t b) { return (((b & 0xFu) * 0x00204081u) & 0x01010101u) << 7; }

// byte after a C3 / C5 lead that makes a Hungarian letter (parser.c:107-124):
// C3 81 89 8D 93 96 9A 9C A1 A9 AD B3 B6 BA BC, C5 90 91 B0 B1; bit index = byte - 0x80
constexpr uint64_t HUN_AFTER_C3 = (1ull << 0x01) | (1ull << 0x09) | (1ull << 0x0D) | (1ull << 0x13) | (1ull << 0x16) |
                                  (1ull << 0x1A) | (1ull << 0x1C) | (1ull << 0x21) | (1ull << 0x29) | (1ull << 0x2D) |
                                  (1ull << 0x33) | (1ull << 0x36) | (1ull << 0x3A) | (1ull << 0x3C);
constexpr uint64_t HUN_AFTER_C5 = (1ull << 0x10) | (1ull << 0x11) | (1ull << 0x30) | (1ull << 0x31);

HUTK_CLS_HD uint32_t classify16(const uint32_t (&d)[8], uint32_t dbits, bool* exotic) {
    const uint32_t H = 0x80808080u;
    // ---- per-byte predicates of dwords 1..6 (window bytes 4..27) ----
    uint32_t cont[7], Dm[7], lowA0[7], low90[7];
    uint32_t l2[6], l3[6], l4[6], alpha[6], digit[6], space[6], wsp[6], oth[6], bad0[6], c3c5[6], eE0[6], eF0[6];
    uint32_t suspect = 0;
    HUTK_CLS_UNROLL
    for (int i = 1; i <= 6; i++) {
        const uint32_t x = d[i];
        const uint32_t hi = x & H;
        const uint32_t b6 = (x << 1) & H, b5 = (x << 2) & H, b4 = (x << 3) & H, b3 = (x << 4) & H;
        cont[i] = hi & ~b6;
        lowA0[i] = cont[i] & ~b5;        // 80..9F
        low90[i] = cont[i] & ~b5 & ~b4;  // 80..8F
        Dm[i] = spread4(dbits >> (4 * i));
        if (i < 6) {
            l2[i] = hi & b6 & ~b5;
            l3[i] = hi & b6 & b5 & ~b4;
            l4[i] = hi & b6 & b5 & b4 & ~b3;
            const uint32_t y = x & 0x7F7F7F7Fu;  // 7-bit values: the adds below cannot carry across bytes
            const uint32_t lower = y | 0x20202020u;
            const uint32_t al = (lower + 0x1F1F1F1Fu) & ~(lower + 0x05050505u);  // 'a'..'z' after case folding
            const uint32_t dg = (y + 0x50505050u) & ~(y + 0x46464646u);          // '0'..'9'
            const uint32_t ws = (y + 0x77777777u) & ~(y + 0x72727272u);          // 9..13
            const uint32_t nsp = (y ^ 0x20202020u) + 0x7F7F7F7Fu;                // bit 7 set iff y != 0x20
            const uint32_t nz = y + 0x7F7F7F7Fu;                                  // bit 7 set iff y != 0
            const uint32_t asc = ~hi & H;
            alpha[i] = al & asc;
            digit[i] = dg & asc;
            wsp[i] = ws & asc;
            space[i] = ~nsp & asc;
            const uint32_t nul = ~nz & asc;
            oth[i] = asc & ~(alpha[i] | digit[i] | wsp[i] | space[i] | nul);
            bad0[i] = nul | (hi & b6 & b5 & b4 & b3);  // 0x00 and F8..FF
            // exact byte tests among bytes >= 0x80 (y = their low 7 bits)
            c3c5[i] = hi & (~((y ^ 0x43434343u) + 0x7F7F7F7Fu) | ~((y ^ 0x45454545u) + 0x7F7F7F7Fu));
            eE0[i] = hi & ~((y ^ 0x60606060u) + 0x7F7F7F7Fu);
            eF0[i] = hi & ~((y ^ 0x70707070u) + 0x7F7F7F7Fu);
            suspect |= hi & ~(((y & 0x7E7E7E7Eu) ^ 0x40404040u) + 0x7F7F7F7Fu);  // C0, C1
        }
    }
    // overlong three- and four-byte forms: E0 + 80..9F, F0 + 80..8F
    HUTK_CLS_UNROLL
    for (int i = 1; i <= 5; i++)
        suspect |= (eE0[i] & nxt(lowA0[i + 1], lowA0[i], 1)) | (eF0[i] & nxt(low90[i + 1], low90[i], 1));
    *exotic = suspect != 0;

    // ---- structurally complete leads of dwords 1..5 ----
    uint32_t v2[6], v3[6], v4[6], hunA[6];
    HUTK_CLS_UNROLL
    for (int q = 1; q <= 5; q++) {
        const uint32_t n1c = nxt(cont[q + 1], cont[q], 1), n2c = nxt(cont[q + 1], cont[q], 2),
                       n3c = nxt(cont[q + 1], cont[q], 3);
        const uint32_t n1d = nxt(Dm[q + 1], Dm[q], 1), n2d = nxt(Dm[q + 1], Dm[q], 2), n3d = nxt(Dm[q + 1], Dm[q], 3);
        v2[q] = l2[q] & n1c & ~n1d;
        v3[q] = l3[q] & n1c & n2c & ~(n1d | n2d);
        v4[q] = l4[q] & n1c & n2c & n3c & ~(n1d | n2d | n3d);
        // Hungarian letters: C3/C5 lead + listed follower -> class A (few per window: per occurrence)
        uint32_t h = 0;
        uint32_t todo = movemask4(v2[q] & c3c5[q]);
        while (todo) {
            const int j = __builtin_ctz(todo);
            todo &= todo - 1;
            // lead at byte j of dword q, follower right after it (possibly byte 0 of dword q + 1); q is a
            // compile-time constant here, so no register array is indexed dynamically (scratch)
            const uint32_t w = j ? funnel_r(d[q + 1], d[q], 8 * j) : d[q];
            const uint32_t lead = w & 0xFFu, fol = (w >> 8) & 0xFFu;
            const uint64_t set = (lead == 0xC3u) ? HUN_AFTER_C3 : HUN_AFTER_C5;
            if ((set >> (fol & 63u)) & 1ull) h |= 0x80u << (8 * j);
        }
        hunA[q] = h;
    }

    uint32_t flags = 0;
    HUTK_CLS_UNROLL
    for (int i = 2; i <= 5; i++) {
        const uint32_t any_c = v2[i] | v3[i] | v4[i], any_p = v2[i - 1] | v3[i - 1] | v4[i - 1];
        const uint32_t interior =
            prv(any_c, any_p, 1) | prv(v3[i] | v4[i], v3[i - 1] | v4[i - 1], 2) | prv(v4[i], v4[i - 1], 3);
        const uint32_t hi = d[i] & H;
        const uint32_t A = alpha[i] | hunA[i];
        const uint32_t O = oth[i] | (any_c & ~hunA[i]);
        const uint32_t X = bad0[i] | (hi & ~interior & ~any_c);  // stray continuation / broken lead / F8+ / NUL
        const uint32_t WX = wsp[i] | X;
        // W/X status of the byte just before this dword: its interior flag depends only on leads
        // at most 3 back, i.e. in the same (previous) dword
        const uint32_t int_p3 = prv(any_p, 0u, 1) | prv(v3[i - 1] | v4[i - 1], 0u, 2) | prv(v4[i - 1], 0u, 3);
        const uint32_t WX_p = wsp[i - 1] | bad0[i - 1] | ((d[i - 1] & H) & ~int_p3 & ~any_p);
        const uint32_t pWX = prv(WX, WX_p, 1);
        // class of the character that ENDS right before position k
        const uint32_t pA = prv(alpha[i], alpha[i - 1], 1) | prv(hunA[i], hunA[i - 1], 2);
        const uint32_t pD = prv(digit[i], digit[i - 1], 1);
        const uint32_t pS = prv(space[i], space[i - 1], 1);
        const uint32_t pO = prv(oth[i], oth[i - 1], 1) | prv(v2[i] & ~hunA[i], v2[i - 1] & ~hunA[i - 1], 2) |
                            prv(v3[i], v3[i - 1], 3) | v4[i - 1];
        const uint32_t lone = pS & (prv(Dm[i], Dm[i - 1], 1) | ~prv(space[i], space[i - 1], 2));
        const uint32_t start = ~interior & H &
                               (Dm[i] | WX | pWX | (space[i] & ~pS) | (A & ~pA & ~lone) | (digit[i] & ~pD & ~lone) |
                                (O & ~pO & ~lone));
        flags |= movemask4(start) << (4 * (i - 2));
    }
    return flags;
}

// ---------------------------------------------------------------------------
// table-driven form: the splitter as a finite automaton over (byte class, "a document starts here").
//
// The mask algebra above spends ~800 instructions per 16 positions; the same function is a 31-state
// automaton, and a lane that walks it over window bytes 4..26 needs one LDS lookup and three integer
// instructions per byte.  State = what the next start decision depends on:
//   no character pending: class of the previous character -- A, D, O, lone space (a space after a non-space
//     or at a document start: the "[ ]?" of the next word), further space, or W/X (each its own word);
//   inside a multi-byte character: how many bytes are still due, which lead it was (C3 / C5 followers decide
//     between Hungarian letter and class O; E0 / F0 followers decide overlong or not), and the start flag the
//     character will get when it completes as class A / class O (that is all the earlier context it needs).
// A start is known when a character completes, up to three bytes after its lead, so a transition emits four
// flags: bit j <-> the byte 3 - j before the current one.  A sequence that breaks (a byte that is no
// continuation, or a document start) turns every byte consumed so far into a one-byte word (class X).
// Overlong forms (C0/C1 leads, E0 + 80..9F, F0 + 80..8F) end in an absorbing EXOTIC state: the caller falls
// back to classify16_exact, as with the mask algebra.
//
// Exactness for positions 0..15 from a cold start at window byte 4: continuation bytes at the front of the
// window are taken as stray although a lead before the window may own them; that can only change flags
// before position 0, because the first byte that is not a continuation (window byte 7 at the latest)
// resynchronises the character structure, and the character in front of position 0 then starts inside the
// window.  Whether a space at window byte 4 is lone needs window byte 3, which picks the initial state.
// ---------------------------------------------------------------------------
namespace dfa {
enum : int {
    C_ALPHA, C_DIGIT, C_SPACE, C_WS, C_NUL, C_OTHER,                    // bytes < 0x80
    C_80, C_80_H3, C_90, C_90_H3, C_90_H5, C_A0, C_A0_H3, C_A0_H5,       // continuation bytes; H3 / H5: Hungarian
    C_L_C0C1, C_L_C3, C_L_C5, C_L_2, C_L_E0, C_L_3, C_L_F0, C_L_4, C_L_BAD,  // letter after a C3 / C5 lead
    N_CLS
};
enum : int { S_A, S_D, S_O, S_SL, S_SM, S_WX, S_P2 = 6, S_PC3 = 8, S_PC5 = 12, S_P3 = 16, S_EXOTIC = 30, N_STATES = 31 };
enum : int { P3a, PE0, P3b, P4a, PF0, P4b, P4c };  // pending three- and four-byte characters: S_P3 + 2 * p + flag
constexpr int ROW_BYTES = 128;                     // one row per state: 2 * N_CLS 16-bit entries, padded
constexpr int DOC_COL_BYTES = 2 * N_CLS;           // the columns for "a document starts at this byte"
constexpr int TABLE_BYTES = N_STATES * ROW_BYTES;
static_assert(2 * DOC_COL_BYTES <= ROW_BYTES, "row holds both column sets");
// The 36 bytes a row leaves free hold the seam map (hutk_internal.h, Tables::seam_hi; 256 dwords) of the context, nine
// entries per row, so that k_tiles finds it in LDS beside the automaton: entry x at seam_offset(x).
constexpr int ROW_FREE = 2 * DOC_COL_BYTES, SEAM_PER_ROW = (ROW_BYTES - ROW_FREE) / 4;
static_assert(ROW_FREE % 4 == 0 && SEAM_PER_ROW * N_STATES >= 256, "the seam map fits the rows' padding");
HUTK_CLS_HD uint32_t seam_offset(uint32_t x) {  // byte offset into the table of entry x (x < 256)
    const uint32_t r = (x * 57u) >> 9;          // x / 9
    return r * (uint32_t)ROW_BYTES + (uint32_t)ROW_FREE + 4u * (x - 9u * r);
}
static_assert(SEAM_PER_ROW == 9, "seam_offset divides by nine");

inline int byte_class(uint32_t b) {
    if (b < 0x80u) {
        switch (cls_ascii(b)) {
            case K_ALPHA: return C_ALPHA;
            case K_DIGIT: return C_DIGIT;
            case K_SPACE: return C_SPACE;
            case K_WS: return C_WS;
            case K_BAD: return C_NUL;
            default: return C_OTHER;
        }
    }
    if (b < 0xC0u) {
        const bool h3 = (HUN_AFTER_C3 >> (b - 0x80u)) & 1ull, h5 = (HUN_AFTER_C5 >> (b - 0x80u)) & 1ull;
        if (b < 0x90u) return h3 ? C_80_H3 : C_80;
        if (b < 0xA0u) return h3 ? C_90_H3 : h5 ? C_90_H5 : C_90;
        return h3 ? C_A0_H3 : h5 ? C_A0_H5 : C_A0;
    }
    if (b < 0xC2u) return C_L_C0C1;
    if (b == 0xC3u) return C_L_C3;
    if (b == 0xC5u) return C_L_C5;
    if (b < 0xE0u) return C_L_2;
    if (b == 0xE0u) return C_L_E0;
    if (b < 0xF0u) return C_L_3;
    if (b == 0xF0u) return C_L_F0;
    if (b < 0xF8u) return C_L_4;
    return C_L_BAD;
}

// start flag of a character of class A / D / O (cls = S_A / S_D / S_O) after context ctx (parser.c:24-58)
inline bool start_after(int ctx, int cls) { return !(ctx == cls || ctx == S_SL); }

struct Step { int next; unsigned emit; };

// no character pending, context ctx, byte of class c; ds: a document starts at this byte
inline Step fresh(int ctx, int c, bool ds) {
    const unsigned HERE = 8u;  // emission bit of the current byte
    const bool sp = ctx == S_SL || ctx == S_SM;
    switch (c) {
        case C_ALPHA: return {S_A, (ds || start_after(ctx, S_A)) ? HERE : 0u};
        case C_DIGIT: return {S_D, (ds || start_after(ctx, S_D)) ? HERE : 0u};
        case C_OTHER: return {S_O, (ds || start_after(ctx, S_O)) ? HERE : 0u};
        case C_SPACE: return {(ds || !sp) ? S_SL : S_SM, (ds || !sp) ? HERE : 0u};
        case C_L_C0C1: return {S_EXOTIC, 0u};
        case C_L_C3: case C_L_C5: case C_L_2: case C_L_E0: case C_L_3: case C_L_F0: case C_L_4: {
            const int cx = ds ? S_WX : ctx;  // a document start forces the start whatever the class turns out to be
            const int fa = start_after(cx, S_A), fo = start_after(cx, S_O);
            switch (c) {
                case C_L_C3: return {S_PC3 + 2 * fa + fo, 0u};
                case C_L_C5: return {S_PC5 + 2 * fa + fo, 0u};
                case C_L_2: return {S_P2 + fo, 0u};
                case C_L_E0: return {S_P3 + 2 * PE0 + fo, 0u};
                case C_L_3: return {S_P3 + 2 * P3a + fo, 0u};
                case C_L_F0: return {S_P3 + 2 * PF0 + fo, 0u};
                default: return {S_P3 + 2 * P4a + fo, 0u};
            }
        }
        default: return {S_WX, HERE};  // \t\n\v\f\r, 0x00, a stray continuation byte, F8..FF: one-byte words
    }
}

inline Step step(int st, int c, bool ds) {
    if (st == S_EXOTIC) return {S_EXOTIC, 0u};
    if (st <= S_WX) return fresh(st, c, ds);
    // a character is pending: which one, with which flags, and how many of its bytes are behind us
    int prog = -1, fa = 0, fo = 0, consumed = 1;
    if (st < S_PC3) { fo = st - S_P2; }
    else if (st < S_PC5) { fa = (st - S_PC3) >> 1; fo = (st - S_PC3) & 1; }
    else if (st < S_P3) { fa = (st - S_PC5) >> 1; fo = (st - S_PC5) & 1; }
    else {
        prog = (st - S_P3) >> 1;
        fo = (st - S_P3) & 1;
        consumed = (prog == P3b || prog == P4b) ? 2 : (prog == P4c) ? 3 : 1;
    }
    const bool cont = c >= C_80 && c <= C_A0_H5;
    if (ds || !cont) {  // broken: the bytes consumed so far are one-byte words, then this byte on its own
        const unsigned flush = ((1u << consumed) - 1u) << (3 - consumed);
        const Step f = fresh(S_WX, c, ds);
        return {f.next, f.next == S_EXOTIC ? 0u : (flush | f.emit)};
    }
    if (st < S_PC3) return {S_O, fo ? 4u : 0u};
    if (st < S_PC5) {
        const bool hun = c == C_80_H3 || c == C_90_H3 || c == C_A0_H3;
        return {hun ? S_A : S_O, (hun ? fa : fo) ? 4u : 0u};
    }
    if (st < S_P3) {
        const bool hun = c == C_90_H5 || c == C_A0_H5;
        return {hun ? S_A : S_O, (hun ? fa : fo) ? 4u : 0u};
    }
    switch (prog) {
        case P3a: return {S_P3 + 2 * P3b + fo, 0u};
        case PE0: return {c >= C_A0 ? S_P3 + 2 * P3b + fo : S_EXOTIC, 0u};  // E0 80..9F: overlong
        case P3b: return {S_O, fo ? 2u : 0u};
        case P4a: return {S_P3 + 2 * P4b + fo, 0u};
        case PF0: return {c >= C_90 ? S_P3 + 2 * P4b + fo : S_EXOTIC, 0u};  // F0 80..8F: overlong
        case P4b: return {S_P3 + 2 * P4c + fo, 0u};
        default: return {S_O, fo ? 1u : 0u};
    }
}

// table[TABLE_BYTES / 2] and lut[256] for classify16_dfa (built once on the host, staged in LDS by k_tiles)
inline void build(uint16_t* table, uint8_t* lut) {
    for (int b = 0; b < 256; b++) lut[b] = (uint8_t)(2 * byte_class((uint32_t)b));
    for (int st = 0; st < N_STATES; st++)
        for (int col = 0; col < ROW_BYTES / 2; col++) {
            Step r{S_EXOTIC, 0u};
            if (col < 2 * N_CLS) r = step(st, col % N_CLS, col >= N_CLS);
            table[st * (ROW_BYTES / 2) + col] = (uint16_t)((r.next * ROW_BYTES) | r.emit);
        }
}
}  // namespace dfa

HUTK_CLS_HD uint32_t classify16_dfa(const uint32_t (&d)[8], uint32_t dbits, const uint16_t* table, const uint8_t* lut,
                                    bool* exotic) {
    // column offsets of window bytes 4..26 first: these lookups do not depend on the state
    uint32_t col[23];
    HUTK_CLS_UNROLL
    for (int k = 4; k <= 26; k++) {
        const uint32_t b = (d[k >> 2] >> (8 * (k & 3))) & 0xFFu;
        col[k - 4] = lut[b] + ((dbits >> k) & 1u) * (uint32_t)dfa::DOC_COL_BYTES;
    }
    // (two walks side by side, the second starting cold at window byte 12, were no faster: measured)
    uint32_t st = ((d[0] >> 24) == 0x20u ? dfa::S_SM : dfa::S_WX) * dfa::ROW_BYTES;
    uint32_t acc = 0;  // bit i <-> window byte i + 1
    const uint8_t* t8 = reinterpret_cast<const uint8_t*>(table);
    HUTK_CLS_UNROLL
    for (int k = 4; k <= 26; k++) {
        const uint32_t e = *reinterpret_cast<const uint16_t*>(t8 + (st | col[k - 4]));
        acc |= (e & 15u) << (k - 4);
        st = e & 0xFF80u;
    }
    *exotic = st == (uint32_t)(dfa::S_EXOTIC * dfa::ROW_BYTES);
    return (acc >> 7) & 0xFFFFu;
}

// The same automaton as TWO walks side by side, for callers that are bound by the latency of the walk's chain of
// dependent lookups rather than by instruction count (k_ptiles): the first covers window bytes 4..18 and yields the
// starts of positions 0..7, the second starts cold at window byte 12 -- the same seven bytes of run-in in front of its
// first used output as the first walk has -- covers bytes 12..26 and yields positions 8..15.  15 steps deep instead of 23.
HUTK_CLS_HD uint32_t classify16_dfa2(const uint32_t (&d)[8], uint32_t dbits, const uint16_t* table, const uint8_t* lut,
                                     bool* exotic) {
    uint32_t col[23];
    HUTK_CLS_UNROLL
    for (int k = 4; k <= 26; k++) {
        const uint32_t b = (d[k >> 2] >> (8 * (k & 3))) & 0xFFu;
        col[k - 4] = lut[b] + ((dbits >> k) & 1u) * (uint32_t)dfa::DOC_COL_BYTES;
    }
    uint32_t sa = ((d[0] >> 24) == 0x20u ? dfa::S_SM : dfa::S_WX) * dfa::ROW_BYTES;
    uint32_t sb = ((d[2] >> 24) == 0x20u ? dfa::S_SM : dfa::S_WX) * dfa::ROW_BYTES;  // window byte 11
    uint32_t acca = 0, accb = 0;  // bit i <-> window byte i + 1
    const uint8_t* t8 = reinterpret_cast<const uint8_t*>(table);
    HUTK_CLS_UNROLL
    for (int i = 0; i < 15; i++) {
        const int ka = 4 + i, kb = 12 + i;
        const uint32_t ea = *reinterpret_cast<const uint16_t*>(t8 + (sa | col[ka - 4]));
        const uint32_t eb = *reinterpret_cast<const uint16_t*>(t8 + (sb | col[kb - 4]));
        acca |= (ea & 15u) << (ka - 4);
        accb |= (eb & 15u) << (kb - 4);
        sa = ea & 0xFF80u;
        sb = eb & 0xFF80u;
    }
    *exotic = sa == (uint32_t)(dfa::S_EXOTIC * dfa::ROW_BYTES) || sb == (uint32_t)(dfa::S_EXOTIC * dfa::ROW_BYTES);
    const uint32_t acc = (acca & 0x7FFFu) | (accb & ~0x7FFFu);
    return (acc >> 7) & 0xFFFFu;
}

}  // namespace hutk
