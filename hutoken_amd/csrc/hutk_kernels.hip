// hutk_kernels.hip -- hand-written HIP kernels (gfx950 / CDNA4, wave64) for
// huToken's batch BPE encode path.  No MFMA: this is integer / indexing work.
//
// Pipeline of one batch (all on one stream, no host round trip):
//
//   k_pre      per tile: first document that can touch the tile (binary search)
//   k_tiles    THE hot kernel.  One 256-thread workgroup owns 2048 input bytes:
//                1. coalesced 16-byte loads of the bytes (+16 before, +272 after) into LDS
//                2. document-start bitmap of the window
//                3. per-byte character code from a +-3 byte neighbourhood
//                   (the reference's splitter, src/parser.c:24-183, is locally decidable)
//                4. word-start flags -> __ballot -> bitmap -> compact word list
//                5. ONE LANE PER WORD: initial symbols (byte -> symbol LUT / character
//                   hash), pair ranks from the device pair table, then the reference's
//                   merge rule "leftmost pair of minimal rank" (src/core.c:66-209,
//                   src/queue.c:152-199) with symbols and pair results held in LDS
//                6. workgroup scan of per-word id counts, dense id run written out
//                7. ids-before-document-start for every document that starts in the tile
//              Words a lane cannot take (more than 48 units, end outside the staged
//              window, or first word of a document when a prefix is configured) become
//              exception records.
//   k_exc      one wavefront per exception word, work pulled from a device counter:
//              the same merge rule, cooperatively (parallel min over the pair array,
//              __shfl_xor reduction), arrays in LDS up to 1024 units, else in HBM.
//   k_scan     exclusive scan of per-tile id counts
//   k_gather   tile runs (+ exception words) -> caller's ids array
//   k_doc_off  out_offsets[]
//
// Rank of a pair = vocabulary id of the concatenated bytes (src/core.c:700-722).
// On the device every possible token is a 20-bit symbol numbered in id order, and
// (left, right) -> merged is one 8-byte slot of an open-addressing table built by
// hutk_loader.cpp; with unique ids "smaller merged symbol" == "smaller rank".
#include <hip/hip_runtime.h>

#include "hutk_device.h"

namespace hutk {

// ------------------------------------------------------------------------
// table lookups
// ------------------------------------------------------------------------
__device__ __forceinline__ uint32_t pair_lookup(const DevTables& T, uint32_t l, uint32_t r) {
    const uint64_t key = ((uint64_t)l << 20) | r;
    uint32_t h = pair_hash(l, r) >> T.pair_shift;
    for (;;) {
        const uint64_t s = T.pair_slots[h];
        if ((s >> 20) == key) return (uint32_t)s & 0xFFFFFu;
        if (s == SLOT_EMPTY) return SYM_NONE;
        h = (h + 1) & T.pair_mask;
    }
}

__device__ __forceinline__ uint32_t char_lookup(const DevTables& T, uint32_t packed) {
    uint32_t h = char_hash(packed) >> T.char_shift;
    for (;;) {
        const uint64_t s = T.char_slots[h];
        if ((uint32_t)(s >> 32) == packed) return (uint32_t)s;
        if (s == SLOT_EMPTY) return SYM_UNK;
        h = (h + 1) & T.char_mask;
    }
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

__device__ __forceinline__ void raise(int32_t* err, int32_t code) { atomicCAS(err, 0, code); }

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
template <typename SymT> struct Sym;
template <> struct Sym<uint32_t> {
    static __device__ __forceinline__ uint32_t narrow(uint32_t v) { return v; }
    static __device__ __forceinline__ uint32_t widen(uint32_t v) { return v; }
    static constexpr uint32_t NONE = SYM_NONE;
};
template <> struct Sym<uint16_t> {
    // SYM_NONE -> 0xFFFF, SYM_UNK -> 0xFFFE
    static __device__ __forceinline__ uint16_t narrow(uint32_t v) {
        return v == SYM_NONE ? (uint16_t)0xFFFFu : v == SYM_UNK ? (uint16_t)0xFFFEu : (uint16_t)v;
    }
    static __device__ __forceinline__ uint32_t widen(uint16_t v) {
        return v == 0xFFFFu ? SYM_NONE : v == 0xFFFEu ? SYM_UNK : (uint32_t)v;
    }
    static constexpr uint16_t NONE = 0xFFFFu;
};

// two independent lookups: both first probes are issued before either is examined,
// so their latencies overlap (the left and right neighbour of a fresh merge)
__device__ __forceinline__ uint32_t pair_resolve(const DevTables& T, uint64_t s, uint64_t key, uint32_t h) {
    for (;;) {
        if ((s >> 20) == key) return (uint32_t)s & 0xFFFFFu;
        if (s == SLOT_EMPTY) return SYM_NONE;
        h = (h + 1) & T.pair_mask;
        s = T.pair_slots[h];
    }
}
__device__ __forceinline__ void pair_lookup2(const DevTables& T, bool on1, uint32_t l1, uint32_t r1, bool on2,
                                             uint32_t l2, uint32_t r2, uint32_t& m1, uint32_t& m2) {
    const uint32_t h1 = pair_hash(l1, r1) >> T.pair_shift;
    const uint32_t h2 = pair_hash(l2, r2) >> T.pair_shift;
    const uint64_t s1 = on1 ? T.pair_slots[h1] : SLOT_EMPTY;
    const uint64_t s2 = on2 ? T.pair_slots[h2] : SLOT_EMPTY;
    m1 = on1 ? pair_resolve(T, s1, ((uint64_t)l1 << 20) | r1, h1) : SYM_NONE;
    m2 = on2 ? pair_resolve(T, s2, ((uint64_t)l2 << 20) | r2, h2) : SYM_NONE;
}

// 64 bits of a bitmap starting at bit `start` (the bitmap has 2 words of slack)
__device__ __forceinline__ uint64_t bits64(const uint32_t* m, int start) {
    const int k = start >> 5, sh = start & 31;
    uint64_t v = ((uint64_t)m[k] | ((uint64_t)m[k + 1] << 32)) >> sh;
    if (sh) v |= (uint64_t)m[k + 2] << (64 - sh);
    return v;
}

// ------------------------------------------------------------------------
// merge loop, one lane per word, arrays in LDS (src/core.c:66-209)
//   Sw[0..n): symbols;  Mw[i]: merged symbol of (live unit i, next live unit)
//   cand: bit i set <=> Mw[i] holds a rank (initial pairs are ranked before the call)
// returns the number of surviving symbols, compacted to Sw[0..cnt)
// ------------------------------------------------------------------------
template <typename SymT>
__device__ int bpe_lane(const DevTables& T, SymT* Sw, SymT* Mw, int n, uint64_t cand) {
    if (n <= 1) return n;
    uint64_t live = (n >= 64) ? ~0ull : ((1ull << n) - 1ull);
    while (cand) {
        uint32_t best = 0xFFFFFFFFu;
        int p = 0;
        for (uint64_t c = cand; c; c &= c - 1) {
            const int i = __builtin_ctzll(c);
            const uint32_t r = rank_of(T, Sym<SymT>::widen(Mw[i]));
            if (r < best) {  // strict: the leftmost pair of equal rank wins (queue.c:162-164)
                best = r;
                p = i;
            }
        }
        const int q = p + 1 + __builtin_ctzll(live >> (p + 1));
        const SymT merged_n = Mw[p];
        const uint32_t merged = Sym<SymT>::widen(merged_n);
        Sw[p] = merged_n;
        live &= ~(1ull << q);
        cand &= ~((1ull << q) | (1ull << p));
        const uint64_t right = (q >= 63) ? 0ull : (live >> (q + 1));
        const uint64_t left = live & ((1ull << p) - 1ull);
        const int q2 = right ? q + 1 + __builtin_ctzll(right) : 0;
        const int p0 = left ? 63 - __builtin_clzll(left) : 0;
        uint32_t mr, ml;
        pair_lookup2(T, right != 0, merged, right ? Sym<SymT>::widen(Sw[q2]) : 0u, left != 0,
                     left ? Sym<SymT>::widen(Sw[p0]) : 0u, merged, mr, ml);
        if (right) {
            Mw[p] = Sym<SymT>::narrow(mr);
            if (mr != SYM_NONE) cand |= 1ull << p;
        }
        if (left) {
            Mw[p0] = Sym<SymT>::narrow(ml);
            if (ml != SYM_NONE) cand |= 1ull << p0;
            else cand &= ~(1ull << p0);
        }
    }
    int cnt = 0;
    for (uint64_t c = live; c; c &= c - 1) Sw[cnt++] = Sw[__builtin_ctzll(c)];
    return cnt;
}

// ------------------------------------------------------------------------
// k_pre
// ------------------------------------------------------------------------
__global__ void k_pre(BatchArgs A, Workspace W) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= A.n_tiles) return;
    const int64_t gw = t * TILE_BYTES - LOOKBACK;
    int64_t lo = 0, hi = A.n_docs + 1;  // first d in [0, n_docs] with offsets[d] >= gw
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (A.offsets[mid] < gw) lo = mid + 1; else hi = mid;
    }
    W.tile_first_doc[t] = lo;
}

// ------------------------------------------------------------------------
// k_tiles
// ------------------------------------------------------------------------
constexpr int WM_WORDS = (TILE_BYTES + HALO) / 32;  // 72 words of per-position bits
constexpr uint32_t EXC_FLAG = 0x8000u;
constexpr int N_PHASE = 10;

#define HUTK_STAMP(k)                                                          \
    do {                                                                       \
        if (W.prof && tid == 0) W.prof[tile * N_PHASE + (k)] = clock64();      \
    } while (0)

template <typename SymT, bool BYTE_MODE>
__global__ __launch_bounds__(TILE_THREADS) void k_tiles(DevTables T, BatchArgs A, Workspace W) {
    __shared__ __attribute__((aligned(16))) uint8_t sb[WINDOW];
    __shared__ uint8_t scode[WINDOW];
    __shared__ uint32_t docm[WINDOW / 32 + 1];
    __shared__ uint32_t wmask[WM_WORDS + 2];   // word starts
    __shared__ uint32_t rmask[WM_WORDS + 2];   // position r: pair (r, r+1) has a rank
    __shared__ uint32_t wpref[WM_WORDS + 1];
    __shared__ uint16_t wstart[TILE_BYTES + 2];
    __shared__ uint16_t wcnt[TILE_BYTES];    // unit count, then id count; EXC_FLAG marks an exception word
    __shared__ uint16_t wpos[TILE_BYTES + 1];  // ids before word w
    __shared__ uint16_t order[TILE_BYTES];   // lane words, longest first
    __shared__ SymT S[TILE_BYTES + HALO];
    __shared__ SymT M[TILE_BYTES + HALO];
    __shared__ SymT s_item_sym[256];
    __shared__ uint8_t s_item_direct[256];
    __shared__ uint32_t s_scan[TILE_THREADS];
    __shared__ uint32_t hist[64], hbase[64];
    __shared__ uint32_t s_misc[4];

    const int tid = threadIdx.x;
    const int64_t tile = blockIdx.x;
    const int64_t t0 = tile * TILE_BYTES;
    const int64_t gw = t0 - LOOKBACK;  // global offset of window index 0
    HUTK_STAMP(0);

    // ---- 1. stage bytes, tables -------------------------------------------------
    for (int c = tid; c < WINDOW / 16; c += TILE_THREADS) {
        const int64_t p = gw + 16 * c;
        if (p >= 0 && p + 16 <= A.n_bytes) {
            *reinterpret_cast<uint4*>(sb + 16 * c) = *reinterpret_cast<const uint4*>(A.bytes + p);
        } else {
            for (int k = 0; k < 16; k++) {
                const int64_t q = p + k;
                sb[16 * c + k] = (q >= 0 && q < A.n_bytes) ? A.bytes[q] : (uint8_t)0;
            }
        }
    }
    s_item_sym[tid] = Sym<SymT>::narrow(T.item_sym[tid]);
    s_item_direct[tid] = T.item_direct[tid];
    if (tid < WINDOW / 32 + 1) docm[tid] = 0;
    if (tid < 64) hist[tid] = 0;
    if (tid < 2) { wmask[WM_WORDS + tid] = 0xFFFFFFFFu; rmask[WM_WORDS + tid] = 0; }
    for (int i = tid; i < TILE_BYTES + 2; i += TILE_THREADS) wstart[i] = 0xFFFFu;
    __syncthreads();

    // ---- 2. document starts inside the window -------------------------------
    const int64_t dfirst = W.tile_first_doc[tile];
    for (int64_t d = dfirst + tid; d <= A.n_docs; d += TILE_THREADS) {
        const int64_t o = A.offsets[d];
        if (o >= gw + WINDOW) break;
        const int li = (int)(o - gw);
        if (li >= 0) atomicOr(&docm[li >> 5], 1u << (li & 31));
    }
    __syncthreads();
    HUTK_STAMP(1);

    // ---- 3. character codes -----------------------------------------------------
    const int64_t tile_end = (t0 + TILE_BYTES < A.n_bytes) ? t0 + TILE_BYTES : A.n_bytes;
    for (int li = tid; li < WINDOW; li += TILE_THREADS) {
        uint8_t c = C_BAD;
        if (li >= 4 && li < WINDOW - 4) c = code_at(sb, docm, li);
        scode[li] = c;
        const int64_t p = gw + li;
        if (p >= t0 && p < tile_end && sb[li] == 0) raise(A.err, HUTK_E_NUL_BYTE);
    }
    __syncthreads();
    HUTK_STAMP(2);

    // ---- 4. word-start flags -> bitmap --------------------------------------
    for (int r0 = 0; r0 < TILE_BYTES + HALO; r0 += TILE_THREADS) {
        const int r = r0 + tid;
        const bool f = word_starts(scode, docm, r + LOOKBACK);
        const unsigned long long bal = __ballot(f);
        if ((tid & 63) == 0) {
            wmask[r >> 5] = (uint32_t)bal;
            wmask[(r >> 5) + 1] = (uint32_t)(bal >> 32);
        }
    }
    __syncthreads();
    HUTK_STAMP(3);

    // ---- 5. word list; byte mode: symbols and initial pair ranks per POSITION ---
    if (tid <= WM_WORDS) {
        uint32_t acc = 0;
        for (int k = 0; k < tid; k++) acc += __popc(wmask[k]);
        wpref[tid] = acc;
    }
    if (BYTE_MODE) {
        // every adjacent byte pair of the window at once: no per-word loop, no divergence,
        // one direct-indexed load each (the 65536-entry byte-pair table is L1/L2 resident)
        const SymT* bp = reinterpret_cast<const SymT*>(T.bytepair);
        for (int r0 = 0; r0 < TILE_BYTES + HALO; r0 += TILE_THREADS) {
            const int r = r0 + tid;
            const uint32_t b = sb[r + LOOKBACK], b2 = sb[r + LOOKBACK + 1];
            S[r] = s_item_sym[b];
            SymT m = Sym<SymT>::NONE;
            if (!((wmask[(r + 1) >> 5] >> ((r + 1) & 31)) & 1u)) m = bp[(b << 8) | b2];
            M[r] = m;
            const unsigned long long bal = __ballot(m != Sym<SymT>::NONE);
            if ((tid & 63) == 0) {
                rmask[r >> 5] = (uint32_t)bal;
                rmask[(r >> 5) + 1] = (uint32_t)(bal >> 32);
            }
        }
    }
    __syncthreads();
    const int limit = (int)(tile_end - t0);  // words are starts at tile offsets < limit
    const int nW = (limit <= 0) ? 0
                                : (int)(wpref[limit >> 5] + __popc(wmask[limit >> 5] & ((1u << (limit & 31)) - 1u)));
    if (tid < WM_WORDS) {
        uint32_t idx = wpref[tid];
        for (uint32_t m = wmask[tid]; m && idx < TILE_BYTES + 2; m &= m - 1)
            wstart[idx++] = (uint16_t)(tid * 32 + __builtin_ctz(m));
    }
    __syncthreads();
    HUTK_STAMP(4);

    // ---- 6a. classify words, count units, histogram of unit counts -------------
    for (int w = tid; w < nW; w += TILE_THREADS) {
        const int ws = wstart[w];
        const int we = wstart[w + 1];
        const int lw = ws + LOOKBACK;
        const bool docfirst = bit_at(docm, lw);
        bool exc = (we == 0xFFFF) || (we - ws > LANE_MAX_BYTES) || (T.has_prefix && docfirst);
        int n = 0;
        if (!exc) {
            const int nb = we - ws;
            if (BYTE_MODE) {
                n = nb;
            } else {
                int i = 0;
                while (i < nb) {
                    const uint32_t b = sb[lw + i];
                    int L = 1;
                    if (b >= 0x80u) {
                        if (scode[lw + i] == C_BAD) raise(A.err, HUTK_E_INVALID_UTF8);
                        else L = (b >= 0xF0u) ? 4 : (b >= 0xE0u) ? 3 : 2;
                    }
                    uint32_t sym;
                    if (s_item_direct[b]) sym = Sym<SymT>::widen(s_item_sym[b]);
                    else if (L == 1) sym = SYM_UNK;
                    else {
                        uint32_t packed = b | ((uint32_t)sb[lw + i + 1] << 8);
                        if (L > 2) packed |= (uint32_t)sb[lw + i + 2] << 16;
                        if (L > 3) packed |= (uint32_t)sb[lw + i + 3] << 24;
                        sym = char_lookup(T, packed);
                    }
                    if (n < LANE_MAX_UNITS) S[ws + n] = Sym<SymT>::narrow(sym);
                    n++;
                    i += L;
                }
            }
            if (n > LANE_MAX_UNITS) exc = true;
        }
        if (exc) {
            wcnt[w] = (uint16_t)EXC_FLAG;
        } else {
            wcnt[w] = (uint16_t)n;
            atomicAdd(&hist[n], 1u);
        }
    }
    __syncthreads();
    if (tid < 64) {  // longest first: bucket n starts after all longer buckets
        uint32_t acc = 0;
        for (int m = 63; m > tid; m--) acc += hist[m];
        hbase[tid] = acc;
        if (tid == 0) s_misc[1] = acc + hist[0];
    }
    __syncthreads();
    const int nL = (int)s_misc[1];
    if (tid < 64) hist[tid] = 0;
    __syncthreads();
    for (int w = tid; w < nW; w += TILE_THREADS) {
        const uint32_t c = wcnt[w];
        if (!(c & EXC_FLAG)) order[hbase[c] + atomicAdd(&hist[c], 1u)] = (uint16_t)w;
    }
    __syncthreads();
    HUTK_STAMP(5);

    // ---- 6b. one lane per word, longest words first -----------------------------
    for (int k = tid; k < nL; k += TILE_THREADS) {
        const int w = order[k];
        const int ws = wstart[w];
        const int n = wcnt[w];
        uint64_t cand;
        if (BYTE_MODE) {
            cand = (n > 1) ? (bits64(rmask, ws) & ((1ull << (n - 1)) - 1ull)) : 0ull;
        } else {
            cand = 0;
            for (int i = 0; i + 1 < n; i++) {
                const uint32_t m = pair_lookup(T, Sym<SymT>::widen(S[ws + i]), Sym<SymT>::widen(S[ws + i + 1]));
                M[ws + i] = Sym<SymT>::narrow(m);
                if (m != SYM_NONE) cand |= 1ull << i;
            }
        }
        wcnt[w] = (uint16_t)bpe_lane<SymT>(T, S + ws, M + ws, n, cand);
    }
    __syncthreads();
    HUTK_STAMP(6);

    // ---- 7. scan of id counts (low half) and exception counts (high half) ---
    const int chunk = (nW + TILE_THREADS - 1) / TILE_THREADS;
    const int wa = tid * chunk < nW ? tid * chunk : nW;
    const int wb = wa + chunk < nW ? wa + chunk : nW;
    uint32_t mine = 0;
    for (int w = wa; w < wb; w++) {
        const uint32_t c = wcnt[w];
        mine += (c & EXC_FLAG) ? 0x10000u : c;
    }
    s_scan[tid] = mine;
    __syncthreads();
    for (int off = 1; off < TILE_THREADS; off <<= 1) {
        const uint32_t v = (tid >= off) ? s_scan[tid - off] : 0u;
        __syncthreads();
        s_scan[tid] += v;
        __syncthreads();
    }
    {
        uint32_t run = s_scan[tid] - mine;  // exclusive
        for (int w = wa; w < wb; w++) {
            wpos[w] = (uint16_t)run;
            const uint32_t c = wcnt[w];
            run += (c & EXC_FLAG) ? 0x10000u : c;
        }
    }
    const uint32_t total = s_scan[TILE_THREADS - 1];
    const uint32_t n_dense = total & 0xFFFFu, n_exc = total >> 16;
    if (tid == 0) {
        wpos[nW] = (uint16_t)n_dense;
        uint32_t first = 0;
        if (n_exc) first = atomicAdd(&W.counters[0], n_exc);
        s_misc[0] = first;
        W.tile_count[tile] = n_dense;
        W.tile_dense[tile] = n_dense;
        W.tile_run_start[tile] = nW ? wstart[0] : 0u;
        W.tile_exc_first[tile] = first;
        W.tile_nexc[tile] = n_exc;
    }
    __syncthreads();
    HUTK_STAMP(7);

    // ---- 8. dense run + exception records -----------------------------------
    int32_t* run_out = W.run + t0 + (nW ? wstart[0] : 0);
    for (int w = tid; w < nW; w += TILE_THREADS) {
        const uint32_t c = wcnt[w];
        const uint32_t pos = wpos[w];
        const int ws = wstart[w];
        if (!(c & EXC_FLAG)) {
            for (uint32_t j = 0; j < c; j++) run_out[pos + j] = sym_to_id(T, Sym<SymT>::widen(S[ws + j]));
        }
    }
    if (n_exc) {  // rare: records in word order (one thread walks the tile's words)
        if (tid == 0) {
            uint64_t slot = s_misc[0];
            for (int w = 0; w < nW; w++) {
                if (!(wcnt[w] & EXC_FLAG)) continue;
                if ((int64_t)slot < W.cap_exc) {
                    ExcRec rec;
                    const int ws = wstart[w], we = wstart[w + 1];
                    rec.ws = t0 + ws;
                    rec.tok_base = 0;
                    rec.len = (we == 0xFFFF) ? -1 : (we - ws);
                    rec.wpos = wpos[w];
                    rec.cnt = 0;
                    rec.tile = (uint32_t)tile;
                    W.exc[slot] = rec;
                } else {
                    raise(A.err, HUTK_E_MEMORY);
                }
                slot++;
            }
        }
    }
    HUTK_STAMP(8);

    // ---- 9. ids emitted before each document that starts in this tile ----------
    for (int64_t d = dfirst + tid; d <= A.n_docs; d += TILE_THREADS) {
        const int64_t o = A.offsets[d];
        if (o >= tile_end) break;
        if (o < t0) continue;
        const int r = (int)(o - t0);
        const uint32_t widx = wpref[r >> 5] + __popc(wmask[r >> 5] & ((1u << (r & 31)) - 1u));
        W.doc_tile_pos[d] = wpos[widx];
    }
    HUTK_STAMP(9);
}

// ------------------------------------------------------------------------
// k_exc: one wavefront per exception word
// ------------------------------------------------------------------------
struct LdsArr {
    uint32_t* p;
    __device__ __forceinline__ uint32_t get(int64_t i) const { return p[i]; }
    __device__ __forceinline__ void set(int64_t i, uint32_t v) const { p[i] = v; }
};
struct HbmArr {  // L1-bypassing accesses: lanes of the wave exchange data through it
    uint32_t* p;
    __device__ __forceinline__ uint32_t get(int64_t i) const {
        return __hip_atomic_load(p + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __device__ __forceinline__ void set(int64_t i, uint32_t v) const {
        __hip_atomic_store(p + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
};

__device__ __forceinline__ uint64_t wave_min_u64(uint64_t v) {
    for (int off = 32; off; off >>= 1) {
        const uint64_t o = __shfl_xor(v, off, 64);
        v = o < v ? o : v;
    }
    return v;
}

// Cooperative merge of n symbols held in Sa (pairs in Ma) by one wavefront.
// Dense arrays: a merge removes element p+1 by shifting the tail left.
template <class Arr>
__device__ int64_t bpe_wave(const DevTables& T, Arr Sa, Arr Ma, int64_t n, int lane) {
    for (int64_t i = lane; i < n; i += 64)
        Ma.set(i, (i + 1 < n) ? pair_lookup(T, Sa.get(i), Sa.get(i + 1)) : SYM_NONE);
    __syncthreads();
    while (n > 1) {
        uint64_t best = ~0ull;
        for (int64_t i = lane; i + 1 < n; i += 64) {
            const uint32_t m = Ma.get(i);
            if (m != SYM_NONE) {
                const uint64_t k = ((uint64_t)rank_of(T, m) << 32) | (uint64_t)i;
                best = k < best ? k : best;
            }
        }
        best = wave_min_u64(best);
        if (best == ~0ull) break;
        const int64_t p = (int64_t)(best & 0xFFFFFFFFull);
        const uint32_t merged = Ma.get(p);
        const bool has_left = p > 0, has_right = p + 2 < n;
        const uint32_t sl = has_left ? Sa.get(p - 1) : 0u;
        const uint32_t sr = has_right ? Sa.get(p + 2) : 0u;
        __syncthreads();
        for (int64_t base = p + 1; base + 1 < n; base += 64) {
            const int64_t i = base + lane;
            uint32_t s = 0, m = 0;
            const bool on = i + 1 < n;
            if (on) {
                s = Sa.get(i + 1);
                m = Ma.get(i + 1);
            }
            __syncthreads();
            if (on) {
                Sa.set(i, s);
                Ma.set(i, m);
            }
            __syncthreads();
        }
        n -= 1;
        if (lane == 0) {
            Sa.set(p, merged);
            Ma.set(p, has_right ? pair_lookup(T, merged, sr) : SYM_NONE);
        }
        if (lane == 1 && has_left) Ma.set(p - 1, pair_lookup(T, sl, merged));
        __syncthreads();
    }
    return n;
}

constexpr int EXC_CHUNK = 256;                    // positions examined per step when a word end is unknown
constexpr int EXC_WIN = 16 + EXC_CHUNK + 16;      // staged bytes per step

__global__ __launch_bounds__(64) void k_exc(DevTables T, BatchArgs A, Workspace W) {
    __shared__ uint32_t Sl[EXC_LDS_UNITS];
    __shared__ uint32_t Ml[EXC_LDS_UNITS];
    __shared__ __attribute__((aligned(16))) uint8_t sb[EXC_WIN];
    __shared__ uint8_t scode[EXC_WIN];
    __shared__ uint32_t docm[EXC_WIN / 32 + 1];
    __shared__ uint32_t s_idx;

    const int lane = threadIdx.x;
    const uint32_t n_exc = W.counters[0];
    // first record by block index, further ones from the device cursor: no atomic
    // traffic at all when there are fewer records than wavefronts
    for (uint32_t round = 0;; round++) {
        uint32_t idx = blockIdx.x;
        if (round) {
            if (lane == 0) s_idx = gridDim.x + atomicAdd(&W.counters[1], 1u);
            __syncthreads();
            idx = s_idx;
            __syncthreads();
        }
        if (idx >= n_exc || (int64_t)idx >= W.cap_exc) break;
        ExcRec rec = W.exc[idx];
        const int64_t ws = rec.ws;

        // document holding ws: last d with offsets[d] <= ws
        int64_t lo = 0, hi = A.n_docs;  // invariant: offsets[lo] <= ws, answer in [lo, hi)
        while (hi - lo > 1) {
            const int64_t mid = (lo + hi) >> 1;
            if (A.offsets[mid] <= ws) lo = mid; else hi = mid;
        }
        const int64_t d = lo, ds = A.offsets[d], de = A.offsets[d + 1];

        // word end, when the tile could not see it
        int64_t we = ws + rec.len;
        bool too_large = false;
        if (rec.len < 0) {
            we = -1;
            for (int64_t base = ws + 1; we < 0; base += EXC_CHUNK) {
                if (base - ws > MAX_WORD_BYTES + 1) { too_large = true; break; }
                const int64_t g0 = base - 16;  // global offset of window index 0
                for (int i = lane; i < EXC_WIN; i += 64) {
                    const int64_t q = g0 + i;
                    sb[i] = (q >= ds && q < de) ? A.bytes[q] : (uint8_t)0;
                }
                if (lane < EXC_WIN / 32 + 1) docm[lane] = 0;
                __syncthreads();
                if (lane == 0) {
                    if (ds >= g0 && ds < g0 + EXC_WIN) docm[(ds - g0) >> 5] |= 1u << ((ds - g0) & 31);
                    if (de >= g0 && de < g0 + EXC_WIN) docm[(de - g0) >> 5] |= 1u << ((de - g0) & 31);
                }
                __syncthreads();
                for (int i = lane; i < EXC_WIN; i += 64)
                    scode[i] = (i >= 4 && i < EXC_WIN - 4) ? code_at(sb, docm, i) : (uint8_t)C_BAD;
                __syncthreads();
                for (int r0 = 0; r0 < EXC_CHUNK && we < 0; r0 += 64) {
                    const int64_t q = base + r0 + lane;
                    const bool f = (q <= de) && word_starts(scode, docm, 16 + r0 + lane);
                    const unsigned long long bal = __ballot(f);
                    if (bal) we = base + r0 + __builtin_ctzll(bal);
                }
                __syncthreads();
            }
        }
        const int64_t nb = too_large ? 0 : we - ws;
        if (too_large || nb > MAX_WORD_BYTES) {
            if (lane == 0) {
                raise(A.err, HUTK_E_WORD_TOO_LARGE);
                if (A.status) A.status[d] = HUTK_DOC_WORD_TOO_LARGE;
                rec.cnt = 0;
                rec.tok_base = -(ws - ds) - 1;  // where the document is cut (negative marks "no ids")
                W.exc[idx] = rec;
            }
            continue;
        }

        const bool docfirst = (ws == ds);
        const bool with_prefix = T.has_prefix && docfirst;
        const bool alone = with_prefix && A.bytes[ws] == ' ';  // core.c:365-366, 421-446
        const int kp = (with_prefix && !alone) ? T.n_prefix : 0;
        const int64_t gbase = ws + (int64_t)W.pad_per_doc * (docfirst ? d : d + 1);

        // unit count
        int64_t n_units;
        if (T.is_byte_encoder) {
            n_units = nb;
        } else {
            int64_t cnt = 0;
            for (int64_t i0 = 0; i0 < nb; i0 += 64) {
                const int64_t i = i0 + lane;
                const bool lead = i < nb && !is_cont(A.bytes[ws + i]);
                cnt += __popcll(__ballot(lead));
            }
            n_units = cnt;
        }
        const int64_t n = n_units + kp;
        const bool in_lds = n <= EXC_LDS_UNITS;
        LdsArr Sl_a{Sl}, Ml_a{Ml};
        HbmArr Sg_a{W.exc_sym + gbase}, Mg_a{W.exc_mrg + gbase};

        // initial symbols
        for (int i = lane; i < kp; i += 64) {
            if (in_lds) Sl_a.set(i, T.prefix_syms[i]); else Sg_a.set(i, T.prefix_syms[i]);
        }
        if (T.is_byte_encoder) {
            for (int64_t i = lane; i < nb; i += 64) {
                const uint32_t sym = T.item_sym[A.bytes[ws + i]];
                if (in_lds) Sl_a.set(kp + i, sym); else Sg_a.set(kp + i, sym);
            }
        } else {
            int64_t ubase = kp;
            for (int64_t i0 = 0; i0 < nb; i0 += 64) {
                const int64_t i = i0 + lane;
                const uint32_t b = i < nb ? A.bytes[ws + i] : 0x80u;
                const bool lead = i < nb && !is_cont(b);
                const unsigned long long bal = __ballot(lead);
                if (lead) {
                    const int L = (b < 0x80u) ? 1 : (b >= 0xF0u) ? 4 : (b >= 0xE0u) ? 3 : 2;
                    uint32_t sym;
                    if (b >= 0xF8u || i + L > nb) {
                        raise(A.err, HUTK_E_INVALID_UTF8);
                        sym = SYM_UNK;
                    } else if (T.item_direct[b]) {
                        sym = T.item_sym[b];
                    } else if (L == 1) {
                        sym = SYM_UNK;
                    } else {
                        uint32_t packed = b | ((uint32_t)A.bytes[ws + i + 1] << 8);
                        if (L > 2) packed |= (uint32_t)A.bytes[ws + i + 2] << 16;
                        if (L > 3) packed |= (uint32_t)A.bytes[ws + i + 3] << 24;
                        sym = char_lookup(T, packed);
                    }
                    const int64_t u = ubase + __popcll(bal & ((1ull << lane) - 1ull));
                    if (in_lds) Sl_a.set(u, sym); else Sg_a.set(u, sym);
                } else if (i == 0 && i < nb) {
                    raise(A.err, HUTK_E_INVALID_UTF8);  // a word cannot begin inside a character
                }
                ubase += __popcll(bal);
            }
        }
        __syncthreads();

        const int64_t left = in_lds ? bpe_wave(T, Sl_a, Ml_a, n, lane) : bpe_wave(T, Sg_a, Mg_a, n, lane);
        const int na = alone ? T.n_prefix_alone : 0;
        int32_t* out = W.exc_tok + gbase;
        for (int i = lane; i < na; i += 64) out[i] = T.prefix_alone_ids[i];
        for (int64_t i = lane; i < left; i += 64)
            out[na + i] = sym_to_id(T, in_lds ? Sl_a.get(i) : Sg_a.get(i));
        if (lane == 0) {
            rec.cnt = (uint32_t)(left + na);
            rec.len = (int32_t)nb;
            rec.tok_base = gbase;
            W.exc[idx] = rec;
            atomicAdd(&W.tile_count[rec.tile], rec.cnt);
        }
        __syncthreads();
    }
}

// one-off: merge a short symbol sequence (the prefix encoded as its own word)
__global__ __launch_bounds__(64) void k_bpe_symbols(DevTables T, const uint32_t* syms, int n, int32_t* ids_out,
                                                     int32_t* n_out) {
    __shared__ uint32_t Sl[EXC_LDS_UNITS];
    __shared__ uint32_t Ml[EXC_LDS_UNITS];
    const int lane = threadIdx.x;
    if (n > EXC_LDS_UNITS) n = EXC_LDS_UNITS;
    for (int i = lane; i < n; i += 64) Sl[i] = syms[i];
    __syncthreads();
    const int64_t left = bpe_wave(T, LdsArr{Sl}, LdsArr{Ml}, n, lane);
    for (int i = lane; i < left; i += 64) ids_out[i] = sym_to_id(T, Sl[i]);
    if (lane == 0) *n_out = (int32_t)left;
}

// ------------------------------------------------------------------------
// k_scan: exclusive scan of tile_count -> tile_base (one workgroup)
// ------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_scan(BatchArgs A, Workspace W) {
    __shared__ int64_t part[1024];
    const int tid = threadIdx.x;
    const int64_t chunk = (A.n_tiles + 1023) / 1024;
    const int64_t a = tid * chunk < A.n_tiles ? tid * chunk : A.n_tiles;
    const int64_t b = a + chunk < A.n_tiles ? a + chunk : A.n_tiles;
    int64_t sum = 0;
    for (int64_t t = a; t < b; t++) sum += W.tile_count[t];
    part[tid] = sum;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const int64_t v = tid >= off ? part[tid - off] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    int64_t run = part[tid] - sum;
    for (int64_t t = a; t < b; t++) {
        W.tile_base[t] = run;
        run += W.tile_count[t];
    }
    if (tid == 1023) W.tile_base[A.n_tiles] = part[1023];
}

// ------------------------------------------------------------------------
// k_gather: tile runs and exception words -> ids_out
// ------------------------------------------------------------------------
constexpr int GATHER_EXC_LDS = 2048;

__global__ __launch_bounds__(256) void k_gather(BatchArgs A, Workspace W) {
    __shared__ uint32_t e_pos[GATHER_EXC_LDS];
    __shared__ uint32_t e_cum[GATHER_EXC_LDS + 1];
    const int tid = threadIdx.x;
    const int64_t tile = blockIdx.x;
    const int64_t base = W.tile_base[tile];
    const uint32_t dense = W.tile_dense[tile];
    const uint32_t nexc = W.tile_nexc[tile];
    const int32_t* run = W.run + tile * TILE_BYTES + W.tile_run_start[tile];
    if (base + (int64_t)W.tile_count[tile] > A.ids_cap) {
        if (tid == 0) raise(A.err, HUTK_E_CAPACITY);
        return;
    }
    if (nexc == 0) {
        for (uint32_t k = tid; k < dense; k += 256) A.ids_out[base + k] = run[k];
        return;
    }
    const ExcRec* recs = W.exc + W.tile_exc_first[tile];
    for (uint32_t e = tid; e < nexc; e += 256) e_pos[e] = recs[e].wpos;
    __syncthreads();
    if (tid == 0) {
        uint32_t acc = 0;
        for (uint32_t e = 0; e < nexc; e++) {
            e_cum[e] = acc;
            acc += recs[e].cnt;
        }
        e_cum[nexc] = acc;
    }
    __syncthreads();
    for (uint32_t k = tid; k < dense; k += 256) {
        // exceptions that come before dense id k: those with wpos <= k
        uint32_t lo = 0, hi = nexc;
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if (e_pos[mid] <= k) lo = mid + 1; else hi = mid;
        }
        A.ids_out[base + k + e_cum[lo]] = run[k];
    }
    for (uint32_t e = 0; e < nexc; e++) {
        const ExcRec r = recs[e];
        if (r.tok_base < 0) continue;
        const int32_t* src = W.exc_tok + r.tok_base;
        const int64_t dst = base + r.wpos + e_cum[e];
        for (uint32_t j = tid; j < r.cnt; j += 256) A.ids_out[dst + j] = src[j];
    }
}

// ------------------------------------------------------------------------
// k_doc_off: out_offsets[d] for d in [0, n_docs]
// ------------------------------------------------------------------------
__global__ void k_doc_off(BatchArgs A, Workspace W) {
    const int64_t d = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (d > A.n_docs) return;
    const int64_t o = A.offsets[d];
    if (o >= A.n_bytes) {
        A.out_offsets[d] = W.tile_base[A.n_tiles];
        return;
    }
    const int64_t tile = o / TILE_BYTES;
    int64_t v = W.tile_base[tile] + W.doc_tile_pos[d];
    const uint32_t nexc = W.tile_nexc[tile];
    if (nexc) {
        const ExcRec* recs = W.exc + W.tile_exc_first[tile];
        for (uint32_t e = 0; e < nexc; e++)
            if (recs[e].ws < o) v += recs[e].cnt;
    }
    A.out_offsets[d] = v;
}

// ------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------
void launch_pre(const BatchArgs& a, const Workspace& w, hipStream_t s) {
    const unsigned g = (unsigned)((a.n_tiles + 255) / 256);
    hipLaunchKernelGGL(k_pre, dim3(g), dim3(256), 0, s, a, w);
}
void launch_tiles(const DevTables& t, const BatchArgs& a, const Workspace& w, hipStream_t s) {
    const dim3 g((unsigned)a.n_tiles), b(TILE_THREADS);
    if (t.sym16) {
        if (t.is_byte_encoder) hipLaunchKernelGGL((k_tiles<uint16_t, true>), g, b, 0, s, t, a, w);
        else hipLaunchKernelGGL((k_tiles<uint16_t, false>), g, b, 0, s, t, a, w);
    } else {
        if (t.is_byte_encoder) hipLaunchKernelGGL((k_tiles<uint32_t, true>), g, b, 0, s, t, a, w);
        else hipLaunchKernelGGL((k_tiles<uint32_t, false>), g, b, 0, s, t, a, w);
    }
}
void launch_exceptions(const DevTables& t, const BatchArgs& a, const Workspace& w, hipStream_t s) {
    // fixed grid; every wavefront pulls records until the device counter runs out
    hipLaunchKernelGGL(k_exc, dim3(4096), dim3(64), 0, s, t, a, w);
}
void launch_scan(const BatchArgs& a, const Workspace& w, hipStream_t s) {
    hipLaunchKernelGGL(k_scan, dim3(1), dim3(1024), 0, s, a, w);
}
void launch_gather(const BatchArgs& a, const Workspace& w, hipStream_t s) {
    hipLaunchKernelGGL(k_gather, dim3((unsigned)a.n_tiles), dim3(256), 0, s, a, w);
}
void launch_doc_offsets(const BatchArgs& a, const Workspace& w, hipStream_t s) {
    const unsigned g = (unsigned)((a.n_docs + 1 + 255) / 256);
    hipLaunchKernelGGL(k_doc_off, dim3(g), dim3(256), 0, s, a, w);
}
void launch_bpe_symbols(const DevTables& t, const uint32_t* d_syms, int n, int32_t* d_ids_out,
                        int32_t* d_n_out, hipStream_t s) {
    hipLaunchKernelGGL(k_bpe_symbols, dim3(1), dim3(64), 0, s, t, d_syms, n, d_ids_out, d_n_out);
}

}  // namespace hutk
