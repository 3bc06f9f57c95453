// hutk_kernels.hip -- hand-written HIP kernels (gfx950 / CDNA4, wave64) for
// huToken's batch BPE encode path.  No MFMA: this is integer / indexing work.
//
// Pipeline of one batch (all on one stream, no host round trip):
//
//   k_pre      per tile: first document that can touch the tile (binary search)
//   k_tiles    THE hot kernel.  One wavefront per tile of 960 input bytes, TILE_WAVES tiles per
//              workgroup:
//                1. coalesced 16-byte loads of the bytes (+16 before, +80 after) into LDS,
//                   document-start bitmap of the window; the workgroup's copy of the splitter's
//                   transition table (4 KB, in the LDS the merge phase uses later)
//                2. each lane classifies its 16 positions from a 32-byte register window (the
//                   reference's splitter, src/parser.c:24-183, is a function of a +-5 byte
//                   neighbourhood) by walking a 31-state automaton, one LDS lookup per byte
//                   (hutk_classify.h), and emits 16 word-start bits
//                3. words go to the lanes round-robin; one round = one memory round trip:
//                   whole-word table probe (raw bytes -> the single token of the word),
//                   first-byte symbol, unit count; outcome: one symbol / needs merging /
//                   exception
//                4. the merge-loop words of the workgroup's tiles are pooled (long ones
//                   first) and taken 64 at a time, ONE WORD PER LANE, one merge per trip:
//                   the reference's rule "leftmost pair of minimal rank" (src/core.c:66-209,
//                   src/queue.c:152-199); symbols and pair results in LDS, the two new
//                   neighbour pairs looked up together in the cuckoo pair table
//                5. epilogue on bitmaps of surviving units and exception words: popcounts,
//                   DPP wave scan, symbols stored to the tile's run, ids-before-document-
//                   start for every document that starts in the tile
//              Words a lane cannot take (more than 32 units or 63 bytes, end outside the
//              staged window, beyond the tile's prefix budget) become exception records.
//   k_exc_a    exception words of up to 63 bytes, one LANE per word (d_exc_lane_fast<1>: one dword per unit, rows read
//              16 bytes at a time; d_exc_medium for vocabularies without the short form); the ends of the words whose
//              end no tile saw (d_exc_ends: the first word start of the tiles behind, Workspace::tile_first_start) and
//              the five lists by length for k_exc_b
//   k_exc_b    words of up to 1024 units: TWO .. SIXTEEN lanes per word, 32 .. 4 words per wavefront (d_exc_group_fast<2>
//              .. <16>: block minima in registers; d_exc_quad, sixteen lanes per word up to 256 units, for 32-bit
//              symbols); longer ones: one wavefront per word (d_exc: up to 2046 / 1024 units in LDS, beyond that in HBM
//              with dead-unit marks and per-chunk minima, bpe_wave_big)
//   k_cut      the reference's over-long-word rule (a document ends in front of a word of more than 262144 bytes)
//   k_scan     exclusive scan of per-tile id counts, one launch, decoupled look-back with tickets
//   k_finish   tile runs -> caller's ids array (symbol -> id), also for the tiles that hold exception words; out_offsets[]
//   k_tail_small  all of the above behind k_tiles for a batch of up to 32 tiles, in one launch
//
// Rank of a pair = vocabulary id of the concatenated bytes (src/core.c:700-722), or the rule's
// line order on the id-keyed path (src/core.c:211-337).  On the device every possible token is a
// 20-bit symbol numbered so that "smaller merged symbol" == "smaller rank", and
// (left, right) -> merged is one 8-byte slot of a two-choice cuckoo table built by hutk_loader.cpp.
#include <hip/hip_runtime.h>

#include <type_traits>

#include "hutk_kdev.h"
#include "hutk_lab.h"

namespace hutk {

// k_cut's notes from a tile (rare, out of line so that the hot path does not carry them): no word start of the
// reference's own among the tile's positions -> its bit in noreal_bits; starts, but none in the halo -> this may be the
// tile in front of a run, and its last start is where k_cut would cut: 1 + position into *cutpos (LDS; the ids in front
// of it are counted in the epilogue).
__device__ __noinline__ void cut_note_cold(uint32_t* noreal_bits, uint32_t* counters, uint32_t tile, unsigned long long mine,
                                                  uint32_t last16, uint32_t* cutpos) {
    // (the two pointers, not the Workspace: a structure passed by value to a function that is not inlined travels through
    // the stack, i.e. scratch memory -- and a kernel that declares scratch makes the queue drain and set it up at its launch)
    if ((threadIdx.x & 63) != 0) return;
    if (mine == 0) {
        atomicOr(&noreal_bits[tile >> 5], 1u << (tile & 31));
        atomicAdd(&counters[6], 1u);
    } else {  // last16: the starts (without seams) of the tile's last lane that has one
        *cutpos = 1u + (uint32_t)(16 * (63 - __builtin_clzll(mine)) + 31 - __builtin_clz(last16 & 0xFFFFu));
    }
}

// Out of line on purpose (the hot path of every tile does not carry it).  The per-position form indexes its window
// dynamically: the window goes to LDS (`tmp`: 8 dwords of the calling lane's own) rather than to a stack array, which
// would be scratch memory.
struct Win8 { uint32_t d[8]; };  // by value: the window travels in registers
__device__ __noinline__ uint32_t classify16_exact_cold(Win8 w, uint32_t dbits, uint32_t* tmp) {
#pragma unroll
    for (int i = 0; i < 8; i++) tmp[i] = w.d[i];
    return classify16_exact(*reinterpret_cast<const uint32_t(*)[8]>(tmp), dbits);
}


// ------------------------------------------------------------------------
// k_pre
// ------------------------------------------------------------------------
__global__ void k_pre(BatchArgs A, Workspace W) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    // what the kernels behind this one add to or publish in: error word, counters, per-document status, scan states
    // (here rather than in memsets of their own: a batch is a handful of launches)
    if (t == 0) *A.err = 0;
    if (t < 16 && t != 10) W.counters[t] = 0;  // ([10]: the sample below adds to it from every workgroup; zeroed for the next batch by k_scan / k_tail_small, behind the tile kernels that read it)
    const int64_t n_threads = (int64_t)gridDim.x * blockDim.x;
    if (A.status)
        for (int64_t d = t; d < A.n_docs; d += n_threads) A.status[d] = 0;
    for (int64_t b = t; b < W.n_scan_blocks; b += n_threads) W.scan_state[b] = 0;
    for (int64_t b = t; b < (A.n_tiles + 31) / 32; b += n_threads) W.noreal_bits[b] = 0;
    // which tile kernel this batch is for (Workspace::select): 16 bytes from the middle of the tiles of every
    // SELECT_BLOCK_STRIDE-th workgroup, how many of them are lead bytes of three- and four-byte characters (text in which
    // they are many is cut into short words by the seams, most of them no tokens of a Latin-trained vocabulary: a
    // merge-loop word every few bytes).  One atomic per sampling workgroup: adds to one address queue up at ~12 ns each.
    if (blockIdx.x % SELECT_BLOCK_STRIDE == 0) {  // (uniform)
        __shared__ uint32_t s_hi;
        if (threadIdx.x == 0) s_hi = 0;
        __syncthreads();
        uint32_t hi = 0;
        const int64_t p = t * TILE_BYTES + TILE_BYTES / 2;
        if (t < A.n_tiles && p + SELECT_SAMPLE <= A.n_bytes) {
            const uint4 v = *reinterpret_cast<const uint4*>(A.bytes + p);  // (16-byte aligned: TILE_BYTES / 2 is)
            const uint32_t K = 0x80808080u;
            hi = (uint32_t)(__popc(v.x & (v.x << 1) & (v.x << 2) & K) + __popc(v.y & (v.y << 1) & (v.y << 2) & K) +
                            __popc(v.z & (v.z << 1) & (v.z << 2) & K) + __popc(v.w & (v.w << 1) & (v.w << 2) & K));
        }
        for (int o = 32; o; o >>= 1) hi += (uint32_t)__shfl_xor((int)hi, o, 64);
        if ((threadIdx.x & 63) == 0 && hi) atomicAdd(&s_hi, hi);
        __syncthreads();
        if (threadIdx.x == 0 && s_hi) atomicAdd(&W.counters[10], s_hi);
    }
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
// k_tiles: ONE WAVEFRONT per tile of 960 input bytes (+64 bytes of halo in which a
// word that starts in the tile may end).  Lane l owns positions 16l .. 16l+15.
// Workgroup barriers only around the pooled merge phase.
//
// Build switches: HUTK_TILE_WAVES tiles per workgroup (4; 2, 6, 8 and 16 measured no faster), HUTK_WAVES_EU resident
// wavefronts per SIMD the byte-mode kernel is compiled for (8 = 64 VGPRs), HUTK_CHAR_EU the same outside byte-encoder
// mode (7: LDS-limited).  Measurement-only switches: hutk_lab.h.  What was tried and dropped: DESIGN.md section 5.
// ------------------------------------------------------------------------
constexpr int N_PHASE = 10;
#ifndef HUTK_CHAR_EU
#define HUTK_CHAR_EU 7
#endif
#ifndef HUTK_WAVES_EU
#define HUTK_WAVES_EU 8
#endif
#ifndef HUTK_TILE_WAVES
#define HUTK_TILE_WAVES 4
#endif
constexpr int TILE_WAVES = HUTK_TILE_WAVES;  // tiles (= wavefronts) per workgroup of k_tiles
constexpr int NPOS = TILE_BYTES + HALO;  // 1024 classified positions, 16 per lane
static_assert(NPOS == 64 * 16, "16 positions per lane");


// Per-tile LDS state.  A workgroup is WAVES wavefronts and owns WAVES consecutive tiles; each wavefront works on
// its own tile except in the merge phase, where the words of all tiles are pooled (phase 6).
template <typename SymT, bool BYTE_MODE>
struct TileLds {
    static constexpr int ARENA_WORDS = 4, ARENA_W = LANE_MAX_UNITS + 4;
    __attribute__((aligned(16))) uint8_t sb[WINDOW];
    uint32_t docm[WINDOW / 32 + 3];
    __attribute__((aligned(8))) uint16_t wmask16[64 + 8];       // word starts, 16 positions per entry
    __attribute__((aligned(8))) uint32_t mergem[NPOS / 32 + 2];  // words still waiting for the merge loop
    __attribute__((aligned(8))) uint32_t excm[NPOS / 32 + 2];    // starts of exception words
    __attribute__((aligned(8))) uint32_t livem[NPOS / 32 + 2];   // surviving units (see phase 5)
    uint16_t stage[64];  // word starts handed to the lanes, 64 at a time; the epilogue's lane prefix later
    __attribute__((aligned(16))) SymT S[NPOS];  // symbol of unit i of the word at ws: S[ws + i]
    // non-byte mode with a prefix: the first word of a document gets the prefix units in front of its own,
    // which does not fit its byte span; up to ARENA_WORDS such words per tile keep their units in this side
    // arena (more than that: exception path)
    SymT arenaS[BYTE_MODE ? 1 : ARENA_WORDS * ARENA_W];
    SymT arenaM[BYTE_MODE ? 1 : ARENA_WORDS * ARENA_W];
    uint32_t arena_live[ARENA_WORDS];  // their surviving units
    uint16_t arena_ws[ARENA_WORDS], arena_n[ARENA_WORDS];
    uint32_t arena_used, extra;  // arena slots taken; extra ids granted (<= RUN_EXTRA)
    uint32_t cutpos;             // k_cut: 1 + position of the tile's last word start of the reference's own, when none follows in the halo; else 0
};

// MULTI: the vocabulary has special-character replacements of several units (rare): a word that holds such an item is an
// exception word.  A build of its own, because even as a uniform branch the test costs the ordinary kernel 2.5 %.
template <typename RunT>
__device__ __forceinline__ void d_gather(const DevTables& T, const BatchArgs& A, const Workspace& W, int64_t vwave);
__device__ __forceinline__ void d_doc_off(const BatchArgs& A, const Workspace& W, int64_t vblock);

// ONE: the whole pipeline of a batch of at most WAVES tiles in THIS launch (hutk_encode(): a sentence): one workgroup does what
// k_pre does for its tiles, the tile kernel, and -- when the batch has no exception word, which is the rule -- the scan, the copy
// of the runs to the caller's ids and the documents' offsets; then it raises Workspace::one_flag (page-locked host memory
// the caller polls: no second and third launch, no hipStreamSynchronize).  With exception words it raises the flag with 2 and the
// host launches k_tail_small behind it.  (The reference does such a call in ~20 us on one core, lib.c:668-720.)
template <typename SymT, bool BYTE_MODE, bool RANK_IS_SYM, int WAVES, bool MULTI = false, bool ONE = false>
__global__ __launch_bounds__(64 * WAVES) __attribute__((amdgpu_waves_per_eu(sizeof(SymT) == 2 ? (BYTE_MODE ? (RANK_IS_SYM ? HUTK_WAVES_EU : 7) : HUTK_CHAR_EU) : 3))) void k_tiles(DevTables T, BatchArgs A_in, Workspace W) {
    BatchArgs A = A_in;  // (ONE: its input pointers are replaced below)
    typedef TileLds<SymT, BYTE_MODE> Tile;
    constexpr int ARENA_WORDS = Tile::ARENA_WORDS, ARENA_W = Tile::ARENA_W;
    constexpr int POOL_CAP = 64 * WAVES;
    constexpr int POOL_LONG_CAP = POOL_CAP * 3 / 8, POOL_LONG = 8;
    static_assert(WAVES <= 32, "pool entries keep the tile-in-workgroup index in 6 bits");
    // the merge loop's short form: 16-bit symbols, rank == symbol order (GPT-2-shaped files, and the id-keyed path)
    constexpr bool FAST_TRIPS = RANK_IS_SYM && sizeof(SymT) == 2;
    __shared__ Tile L[WAVES];
    __shared__ uint32_t pool_cnt[3];  // long words, other words, entries of the arena handed out
    // Exception records and places on k_gather_exc's list are claimed per WORKGROUP (one 64-bit atomic for its tiles, in the
    // shadow of the merge phase's barriers), not per tile: a batch full of exception words was bound by those same-address
    // atomics (100 k of them at ~12 ns: 1.2 ms for 97 MB of 33-62-letter words).  s_exc_cnt: records | tiles << 16 of
    // the workgroup's tiles, s_exc_base: what the atomic returned.
    __shared__ uint32_t s_exc_cnt;
    __shared__ unsigned long long s_exc_base;
    __shared__ SymT s_item_sym[BYTE_MODE ? 2 : 256];      // non-byte mode only: lead byte -> symbol
    __shared__ uint32_t s_item_direct[BYTE_MODE ? 1 : 8];  // one bit per lead byte (a byte each would cost a resident workgroup)
    // Merge phase: the pool of words and m, the pair results of units (i, next live) of a pooled word at
    // m[its offset + i]; a word gets its stretch of m when it enters the pool (a position-indexed array per tile
    // would be four times the size, and LDS is what limits the resident wavefronts).
    // The merge phase begins behind a workgroup barrier that every wavefront passes after its classification, so until
    // then the same LDS holds the splitter automaton (hutk_classify.h): transition table, then byte classes.
    constexpr int M_ARENA = 352 * WAVES + 192;
    struct MergeLds {
        __attribute__((aligned(16))) SymT m[M_ARENA];
        uint32_t pool[POOL_CAP];
    };
    struct SplitLds {
        uint8_t dfa[dfa::TABLE_BYTES];
        uint8_t lut[256];
    };
    __shared__ __attribute__((aligned(16))) union {
        MergeLds g;
        SplitLds c;
    } s_m;
    static_assert(M_ARENA >= 2 * LANE_MAX_UNITS, "arena size");
    uint32_t* const pool = s_m.g.pool;
#if HUTK_LAB_LDS_PAD
    __shared__ uint32_t s_lab_pad[HUTK_LAB_LDS_PAD / 4];  // MEASUREMENT ONLY
    if (A.n_docs < 0) s_lab_pad[threadIdx.x] = 1;
#endif

    // The wavefront's index is uniform, and told so the compiler keeps what derives from it in scalar registers: 6 vector
    // registers fewer and no spill in byte-encoder mode (+1 %); outside it the extra scalar work costs 2 %, so not there.
    if (select_skips(W, A.n_tiles)) return;  // (both tile kernels are enqueued and this batch is the other one's; uniform)
    if constexpr (ONE) {
        // The batch's offsets and bytes are in the caller's page-locked host memory (offsets, then -- 16-byte aligned -- the
        // bytes): copied into device memory ONCE, every later read of them would be a round trip over PCIe (five in a row
        // for a one-tile batch: ~8 us of a 27 us launch).
        const size_t off_bytes = (((size_t)A.n_docs + 1) * 8 + 15) & ~(size_t)15;
        const size_t n16 = (off_bytes + (size_t)A.n_bytes + 15) / 16;
        const uint4* src = reinterpret_cast<const uint4*>(A.offsets);
        uint4* dst = reinterpret_cast<uint4*>(W.one_in);
        for (size_t i = threadIdx.x; i < n16; i += 64 * WAVES) dst[i] = src[i];
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
        __syncthreads();
        A.offsets = reinterpret_cast<const int64_t*>(W.one_in);
        A.bytes = W.one_in + off_bytes;
    }
    const int lane = threadIdx.x & 63;
    const int wv = BYTE_MODE ? __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) : (int)(threadIdx.x >> 6);
    // XCD-aware order: workgroups are dealt round-robin to the 8 XCDs, each with its own L2.  Workgroup b works on
    // tile group (b % 8) * (groups / 8) + b / 8, so every XCD walks one contiguous eighth of the batch and the
    // lines two neighbouring tiles share (halo, lookback, offsets) meet in one L2.  The grid is a multiple of 8.
    const int64_t wg = (int64_t)(blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const int64_t tile = wg * WAVES + wv;
    const bool tile_ok = tile < A.n_tiles;  // a wavefront without a tile still merges pooled words
    Tile& me = L[wv];
    uint8_t* const sb = me.sb;
    uint32_t* const docm = me.docm;
    uint16_t* const wmask16 = me.wmask16;
    uint32_t* const mergem = me.mergem;
    uint32_t* const excm = me.excm;
    uint32_t* const livem = me.livem;
    uint16_t* const stage = me.stage;
    SymT* const S = me.S;
    SymT* const arenaS = me.arenaS;
    uint32_t* const arena_live = me.arena_live;
    uint16_t* const arena_ws = me.arena_ws;
    uint16_t* const arena_n = me.arena_n;
    uint32_t& s_arena_used = me.arena_used;
    uint32_t& s_extra = me.extra;
    auto RK = [&](uint32_t merged) -> uint32_t {  // rank used in comparisons
        if (RANK_IS_SYM) return merged;
        return (uint32_t)T.sym_id[merged] ^ 0x80000000u;  // signed id order as unsigned
    };
    // best (lowest rank, leftmost) among the candidate pairs `cand` of a word whose pair results are at Mw[].
    // Four candidates per step: their LDS reads are independent, so a step costs one LDS latency, not four.
    // Strict "<" keeps the leftmost pair of equal rank (queue.c:162-164); a step that runs out of
    // candidates repeats its first one, which cannot win against itself.
    auto scan_best = [&](uint32_t c, const SymT* Mw, uint32_t& br, int& bp, SymT& bm) {
        while (c) {
            const uint32_t c1 = c & (c - 1), c2 = c1 & (c1 - 1), c3 = c2 & (c2 - 1);
            const int i0 = __builtin_ctz(c);
            const int i1 = c1 ? __builtin_ctz(c1) : i0, i2 = c2 ? __builtin_ctz(c2) : i0,
                      i3 = c3 ? __builtin_ctz(c3) : i0;
            const SymT m0 = Mw[i0], m1 = Mw[i1], m2 = Mw[i2], m3 = Mw[i3];
            const uint32_t r0 = RK(Sym<SymT>::widen(m0)), r1 = RK(Sym<SymT>::widen(m1)),
                           r2 = RK(Sym<SymT>::widen(m2)), r3 = RK(Sym<SymT>::widen(m3));
            if (r0 < br) { br = r0; bp = i0; bm = m0; }
            if (r1 < br) { br = r1; bp = i1; bm = m1; }
            if (r2 < br) { br = r2; bp = i2; bm = m2; }
            if (r3 < br) { br = r3; bp = i3; bm = m3; }
            c = c3 & (c3 - 1);
        }
    };
    // units of the lane-path word that starts at position ws of tile X
    auto word_units = [&](Tile& X, int ws) -> int {
        const int nb = 1 + __builtin_ctzll(bits64(reinterpret_cast<const uint32_t*>(X.wmask16), ws + 1));
        if (BYTE_MODE) return nb;
        for (int a = 0; a < ARENA_WORDS; a++)
            if (X.arena_ws[a] == ws) return X.arena_n[a];
        int n = 0;  // units = characters = lead bytes
        for (int i = 0; i < nb; i++) n += !is_cont(X.sb[ws + LOOKBACK + i]);
        return n;
    };
    const int64_t t0 = tile * TILE_BYTES;
    const int64_t gw = t0 - LOOKBACK;  // global offset of window index 0
    const uint32_t* wmask32 = reinterpret_cast<const uint32_t*>(wmask16);
    const int64_t tile_end = (t0 + TILE_BYTES < A.n_bytes) ? t0 + TILE_BYTES : A.n_bytes;
    int64_t dfirst = 0;
    if constexpr (ONE) {
        // what k_pre does: the error word, the counters, the documents' status (used behind the barrier below), and the first
        // document that can touch my tile: the number of documents that begin in front of the tile's window
        if (threadIdx.x == 0) { *A.err = 0; W.noreal_bits[0] = 0; }
        if (threadIdx.x < 16) W.counters[threadIdx.x] = 0;
        if (A.status)
            for (int64_t d = threadIdx.x; d < A.n_docs; d += 64 * WAVES) A.status[d] = 0;
        if (tile_ok) {
            int64_t cnt = 0;
            for (int64_t d0 = 0; d0 <= A.n_docs; d0 += 64) {
                const int64_t d = d0 + lane;
                cnt += __popcll(__ballot(d <= A.n_docs && A.offsets[d] < gw));
            }
            dfirst = cnt;
            if (lane == 0) W.tile_first_doc[tile] = cnt;
        }
    } else {
        dfirst = tile_ok ? W.tile_first_doc[tile] : 0;
    }
    uint32_t own = 0;  // word starts of my 16 positions that are words of this tile
    // The tile's bytes are requested BEFORE the automaton's table is copied, and the offsets of its first documents before
    // that copy is waited for: one memory round trip for the three instead of three in a row (the kernel is bound by
    // the latency of its chains of dependent accesses).  A tile at either end of the data is staged
    // byte by byte below.
    static_assert(WINDOW / 16 > 64 && WINDOW / 16 <= 128, "two chunks per lane");
    const bool whole = tile_ok && gw >= 0 && gw + WINDOW <= A.n_bytes;
    uint4 pre0 = make_uint4(0, 0, 0, 0), pre1 = pre0;
    if (whole) {
        pre0 = *reinterpret_cast<const uint4*>(A.bytes + gw + 16 * lane);
        if (lane < WINDOW / 16 - 64) pre1 = *reinterpret_cast<const uint4*>(A.bytes + gw + 16 * (lane + 64));
    }
    constexpr int DFA_CHUNKS = (dfa::TABLE_BYTES + 256) / 16;
    static_assert(DFA_CHUNKS <= 2 * 64 * WAVES || WAVES < 4, "two chunks of the automaton's table per thread");
    uint4 dfa0 = make_uint4(0, 0, 0, 0), dfa1 = dfa0;
    if (WAVES >= 4) {
        if ((int)threadIdx.x < DFA_CHUNKS) dfa0 = T.split_dfa[threadIdx.x];
        if ((int)threadIdx.x + 64 * WAVES < DFA_CHUNKS) dfa1 = T.split_dfa[threadIdx.x + 64 * WAVES];
    }
    int64_t pre_o = 0;  // offsets[dfirst + lane]
    if (tile_ok && !A.first_bits && dfirst + lane <= A.n_docs) pre_o = A.offsets[dfirst + lane];
    if (WAVES >= 4) {
        if ((int)threadIdx.x < DFA_CHUNKS) reinterpret_cast<uint4*>(&s_m.c)[threadIdx.x] = dfa0;  // table, then byte classes, as uploaded
        if ((int)threadIdx.x + 64 * WAVES < DFA_CHUNKS) reinterpret_cast<uint4*>(&s_m.c)[threadIdx.x + 64 * WAVES] = dfa1;
    } else {
        for (int i = threadIdx.x; i < DFA_CHUNKS; i += 64 * WAVES) reinterpret_cast<uint4*>(&s_m.c)[i] = T.split_dfa[i];
    }
    if (!BYTE_MODE) {
        for (int i = threadIdx.x; i < 256; i += 64 * WAVES) {
            s_item_sym[i] = Sym<SymT>::narrow(T.item_sym[i]);
            const unsigned long long direct = __ballot(T.item_direct[i] != 0);  // (a wavefront's 64 values of i are consecutive)
            if (lane == 0) {
                s_item_direct[i >> 5] = (uint32_t)direct;
                s_item_direct[(i >> 5) + 1] = (uint32_t)(direct >> 32);
            }
        }
    }
    if (threadIdx.x == 0) s_exc_cnt = 0;
    __syncthreads();
    uint32_t my_exc_off = 0;  // my tile's share of the workgroup's claim: records before mine | tiles before mine << 16
    if (tile_ok) {
        HUTK_STAMP(0);

        // ---- 1. stage bytes and tables ------------------------------------------------
        if (whole) {
            *reinterpret_cast<uint4*>(sb + 16 * lane) = pre0;
            if (lane < WINDOW / 16 - 64) *reinterpret_cast<uint4*>(sb + 16 * (lane + 64)) = pre1;
        } else
        for (int c = lane; c < WINDOW / 16; c += 64) {
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
        if (lane < WINDOW / 32 + 3) docm[lane] = 0;
        if (lane < NPOS / 32 + 2) { mergem[lane] = 0; excm[lane] = 0; livem[lane] = 0; }
        if (lane == 0) { s_arena_used = 0; s_extra = 0; me.cutpos = 0; }
        if (lane < ARENA_WORDS) { arena_ws[lane] = 0xFFFFu; arena_live[lane] = 0; }
        if (lane < 8) wmask16[64 + lane] = 0xFFFFu;
        wave_sync();

        // ---- 2. document starts inside the window -------------------------------
        // (regex pre-token path with a prefix: the FIRST MATCHES of the documents instead, from the host's bitmap; the
        // classifier, which needs the real starts, is not run on that path)
        if (A.first_bits) {
            if (lane < WINDOW / 32) {
                const int64_t g = gw + 32 * lane;  // window bits [32 lane, 32 lane + 32); gw = 16 (mod 32), or -16 for tile 0
                const int64_t w = g >> 5;          // (floor)
                const uint32_t lo = w >= 0 ? A.first_bits[w] : 0u, hi = A.first_bits[w + 1];
                docm[lane] = (lo >> 16) | (hi << 16);
            }
        } else
        for (int64_t d = dfirst + lane; d <= A.n_docs; d += 64) {
            const int64_t o = d == dfirst + lane ? pre_o : A.offsets[d];
            if (o >= gw + WINDOW) break;
            const int li = (int)(o - gw);
            if (li >= 0) atomicOr(&docm[li >> 5], 1u << (li & 31));
        }
        wave_sync();
        HUTK_STAMP(1);

        // ---- 3. classification in registers: 32-byte window per lane ------------------
        // window-local index k <-> window index kb - 8 + k; own positions are k = 8..23
        const int kb = LOOKBACK + 16 * lane;
        Win w;
        {
            const uint64_t* src = reinterpret_cast<const uint64_t*>(sb + kb - 8);
            w.a = src[0]; w.b = src[1]; w.c = src[2]; w.d = src[3];
        }
        const uint32_t dbits = (uint32_t)bits64(docm, kb - 8);
        uint32_t flags;  // word starts of my 16 positions
        unsigned long long with_start;
        uint32_t last_real16;
        {
            const uint32_t dw[8] = {(uint32_t)w.a, (uint32_t)(w.a >> 32), (uint32_t)w.b, (uint32_t)(w.b >> 32),
                                    (uint32_t)w.c, (uint32_t)(w.c >> 32), (uint32_t)w.d, (uint32_t)(w.d >> 32)};
            bool exotic;
            if (A.word_bits) {
                // regex pre-token path (core.c:350-360, 372-378): the host ran the pattern over the documents; bit p of
                // word_bits = a word (or a stretch no match covers, see gap_bits) starts at byte p
                flags = reinterpret_cast<const uint16_t*>(A.word_bits)[(t0 >> 4) + lane];
                exotic = false;
            } else {
                flags = classify16_dfa(dw, dbits, reinterpret_cast<const uint16_t*>(s_m.c.dfa), s_m.c.lut,
                                       &exotic);                   // the automaton: one LDS lookup per byte
            }
            if (exotic && !HUTK_LAB_NO_COLD) {  // overlong encodings: per-position decode
                Win8 w8;
    #pragma unroll
                for (int i = 0; i < 8; i++) w8.d[i] = dw[i];
                flags = classify16_exact_cold(w8, dbits, reinterpret_cast<uint32_t*>(S) + 8 * lane);  // (the tile's symbols are not written yet: 32 bytes of LDS per lane)
            }
            // The reference ends a document at a word of more than 262144 bytes (core.c:402-407).  With seams such a word can
            // be a run of short ones here, so k_cut looks for what every such word leaves behind: at least 272 tiles in
            // a row without a start of the reference's own (taken here, before the seams come in).  Normal text never
            // sets a bit.
            // (k_cut, below: the lanes with a word start of the reference's own, and the starts of the last such lane of the tile)
            with_start = __ballot(flags != 0);
            last_real16 = (uint32_t)__builtin_amdgcn_readlane((int)flags, 63 - __builtin_clzll((with_start & 0x0FFFFFFFFFFFFFFFull) | 1ull));
            if (T.seam_on && !A.word_bits) {
                // Seams (hutk_internal.h, Tables::seam_hi): where no merge can join the input byte x to the lead byte y of the
                // three- or four-byte character behind it, the word's encoding is the concatenation of the encodings of
                // its two sides (core.c:66-209 merges only what the pair table holds), so y starts a word of its own:
                // runs of CJK characters become table words instead of the merge loop's longest ones.
                const uint64_t K8 = 0x8080808080808080ull;
                uint64_t m0 = w.b & (w.b << 1) & (w.b << 2) & K8;  // bytes >= 0xE0 among my positions 0..7
                uint64_t m1 = w.c & (w.c << 1) & (w.c << 2) & K8;  // ... 8..15
                if (m0 | m1) {
                    const uint64_t p0 = (w.b << 8) | (w.a >> 56), p1 = (w.c << 8) | (w.b >> 56);  // the byte in front of each
                    for (; m0; m0 &= m0 - 1) {
                        const int sh = __builtin_ctzll(m0) - 7;
                        const uint32_t y = (uint32_t)(w.b >> sh) & 0xFFu, x = (uint32_t)(p0 >> sh) & 0xFFu;
                        const uint32_t sm = *reinterpret_cast<const uint32_t*>(s_m.c.dfa + dfa::seam_offset(x));
                        bool join = (sm >> (y & 31u)) & 1u;
#if HUTK_LAB_SEAM2_TILES
                        if (join && T.seam2_on) join = !seam2_cuts(T, win24(w.a, w.b, 40 + sh), win24(w.b, w.c, sh));  // whole characters (Tables::seam2_*)
#endif
                        if (!join) flags |= 1u << (sh >> 3);
                    }
                    for (; m1; m1 &= m1 - 1) {
                        const int sh = __builtin_ctzll(m1) - 7;
                        const uint32_t y = (uint32_t)(w.c >> sh) & 0xFFu, x = (uint32_t)(p1 >> sh) & 0xFFu;
                        const uint32_t sm = *reinterpret_cast<const uint32_t*>(s_m.c.dfa + dfa::seam_offset(x));
                        bool join = (sm >> (y & 31u)) & 1u;
#if HUTK_LAB_SEAM2_TILES
                        if (join && T.seam2_on) join = !seam2_cuts(T, win24(w.b, w.c, 40 + sh), win24(w.c, w.d, sh));
#endif
                        if (!join) flags |= 1u << (8 + (sh >> 3));
                    }
                }
            }
        }
        {  // a 0x00 byte inside the data is an error (the reference's strings end there): any zero among my 16 bytes?
            const int64_t valid = tile_end - (t0 + 16 * lane);  // my positions that are data of this tile
            const uint64_t lo = w.b, hi = w.c;                  // window bytes 8..15 and 16..23
            const uint64_t K1 = 0x0101010101010101ull, K8 = 0x8080808080808080ull;
            uint64_t zlo = (lo - K1) & ~lo & K8, zhi = (hi - K1) & ~hi & K8;  // lowest set flag is exact
            if (valid < 16) {
                const int v = valid < 0 ? 0 : (int)valid;
                zlo &= v >= 8 ? ~0ull : ((1ull << (8 * v)) - 1ull);
                zhi &= v <= 8 ? 0ull : ((1ull << (8 * (v - 8))) - 1ull);
            }
            if (zlo | zhi) raise(A.err, HUTK_E_NUL_BYTE);
        }
        {   // the tile's first word start, for the words of EARLIER tiles that end here (d_exc_ends)
            const unsigned long long sb_ = __ballot((flags & 0xFFFFu) != 0);
            const int l0 = sb_ ? __builtin_ctzll(sb_) : 0;
            const uint32_t f0 = (uint32_t)__builtin_amdgcn_readlane((int)flags, l0) & 0xFFFFu;
            if (lane == 0 && tile_ok) W.tile_first_start[tile] = sb_ ? (uint32_t)(16 * l0 + __builtin_ctz(f0)) : 0xFFFFu;
        }
        wmask16[lane] = (uint16_t)flags;
#if HUTK_PERTURB_VALU
        {   // MEASUREMENT ONLY: extra VALU instructions (a dependent chain, every lane) -- is the kernel bound by VALU issue?
            uint32_t x = flags | 1u;
            for (int i = 0; i < HUTK_PERTURB_VALU / 4; i++) {
                x ^= x << 13; x ^= x >> 17;
                asm volatile("" : "+v"(x));
            }
            if (x == 0x9E3779B9u) raise(A.err, HUTK_E_MEMORY);
        }
#endif
        HUTK_STAMP(2);

        wave_sync();
        HUTK_STAMP(3);

        // ---- 5. words, spread evenly over the lanes: whole-word table; mark what needs merging ----
        // Word j of the tile goes to lane j % 64: every round the owning lanes put the starts of words
        // [r0, r0 + 64) into a 64-entry staging buffer (no per-tile word list is kept in LDS: its
        // footprint would cost resident wavefronts, and residency is what hides the gather latency).
        //
        // State handed to the epilogue, all of it bitmaps over tile positions:
        //   livem  units that survive: starts as "every word start" (the first unit of a word always
        //          survives); the merge loop adds the other survivors of its words (unit i of the word at ws
        //          is bit ws + i, and S[ws + i] its symbol); exception and arena words are taken out
        //   excm   starts of exception words
        const int limit = (int)(tile_end - t0);  // words are starts at tile offsets < limit
        own = flags;                             // starts that are words of this tile
        {
            const int lo = 16 * lane;
            if (lo >= limit) own = 0;
            else if (lo + 16 > limit) own &= (1u << (limit - lo)) - 1u;
        }
        if (!A.word_bits && HUTK_LAB_NO_CUTFLAGS != 1) {
            // The reference ends a document at a word of more than 262144 bytes (core.c:402-407).  With seams such a word can
            // be a run of short ones here, so k_cut looks for what every such word leaves behind: at least 272 tiles in
            // a row without a start of the reference's own (with_start: taken before the seams came in).  The lanes
            // whose positions are data of this tile: a last lane that is only partly data counts whole, which can only leave
            // the batch's last tile unmarked, and k_cut counts whole tiles.  Both cases are rare and live out of line.
            const unsigned long long mine = with_start & ((1ull << ((limit + 15) >> 4)) - 1ull);
            if (!HUTK_LAB_NO_COLD && (mine == 0 || (with_start >> 60) == 0))
                cut_note_cold(W.noreal_bits, W.counters, (uint32_t)tile, mine, last_real16, &me.cutpos);
        }
        reinterpret_cast<uint16_t*>(livem)[lane] = (uint16_t)own;
        uint32_t rest = own;  // my word starts that the general rounds below take
        if constexpr (BYTE_MODE && !MULTI) {
            if (!A.word_bits && (HUTK_LAB_SLIM_ROUNDS || ONE)) {  // (uniform; in the one-launch kernel, where a round's length IS the call's latency: 1.5 k cycles a round less)
                // SHORT words first, in rounds that do nothing else.  The starts of my 16 positions by the length of their word,
                // all sixteen at once: f32 = my starts and the next lane's; x: bit j = a start among positions j + 1 .. j +
                // WORD_KEY, i.e. the word at j is at most WORD_KEY bytes (the key of the whole-word table) and ends inside the
                // classified positions.  With a prefix the first word of a document is an exception word whatever its length
                // (core.c:364-366, 421-451): left, like the long words, to the general rounds.
                constexpr int WORD_KEY = sizeof(SymT) == 2 ? WORD_KEY_BYTES_16 : WORD_KEY_BYTES_32;
                constexpr int KSH = sizeof(SymT) == 2 ? 16 : 32;  // bits of k[3] that are symbol, not key
                const uint32_t f32 = flags | ((uint32_t)wmask16[lane + 1] << 16);
                uint32_t x = f32 >> 1;
                x |= x >> 1;
                x |= x >> 2;
                x |= x >> 4;
                x |= x >> (WORD_KEY - 8);  // 1 .. 8 and WORD_KEY - 7 .. WORD_KEY
                uint32_t shorts = own & x;
                if (T.has_prefix) shorts &= ~(uint32_t)bits64(docm, 16 * lane + LOOKBACK);
                rest = own & ~shorts;
                uint32_t nS;
                uint32_t sidx = wave_excl_scan(__popc(shorts), lane, &nS);
                for (uint32_t r0 = 0; r0 < nS; r0 += 64) {
                    while (shorts && sidx - r0 < 64u) {  // (each start is visited once, in the round it belongs to)
                        stage[sidx - r0] = (uint16_t)(16 * lane + __builtin_ctz(shorts));
                        shorts &= shorts - 1;
                        sidx++;
                    }
                    wave_sync();
                    if (r0 + lane < nS) {
                        // One unaligned LDS read each for the start bits behind the word (at least 24 of them: the next start
                        // is within WORD_KEY) and for its bytes; both table slots and the first byte's symbol loaded together.
                        const uint32_t ws = stage[lane];
                        uint32_t wb;
                        __builtin_memcpy(&wb, reinterpret_cast<const uint8_t*>(wmask16) + ((ws + 1u) >> 3), 4);
                        const uint32_t nb = 1u + (uint32_t)__builtin_ctz(wb >> ((ws + 1u) & 7u));
                        uint4 key;
                        __builtin_memcpy(&key, sb + LOOKBACK + ws, 16);
                        const uint32_t isym = T.item_sym[key.x & 0xFFu];
                        const uint64_t mlo = nb >= 8 ? ~0ull : ((1ull << (8 * nb)) - 1ull);
                        const uint64_t mhi = nb <= 8 ? 0ull : ((1ull << (8 * (nb - 8))) - 1ull);  // (nb <= 14)
                        const uint32_t k0 = key.x & (uint32_t)mlo, k1 = key.y & (uint32_t)(mlo >> 32);
                        const uint32_t k2 = key.z & (uint32_t)mhi, k3 = key.w & (uint32_t)(mhi >> 32);
                        bool done = nb == 1;
                        uint32_t sym = isym;
                        if (T.word_mask) {  // (uniform)
                            const uint32_t wh = word_hash(k0, k1, k2, k3);
                            uint4 s1 = reinterpret_cast<const uint4*>(T.word_tab)[wh & T.word_mask];
                            uint4 s2 = reinterpret_cast<const uint4*>(T.word_tab)[word_slot2(wh, T.word_mask)];
                            // (both loads in flight together: short of registers, the compiler otherwise compares the first slot
                            // before it asks for the second -- two memory round trips per round, seen in the ISA)
                            asm volatile("" : "+v"(s1.x), "+v"(s2.x));
                            // bitwise on purpose: with && the compiler fetches one word first and the rest only on a match
                            // (a one-byte word matches no slot: keys have at least two bytes, an empty slot is all zero)
                            const uint32_t d1 = KSH == 32 ? 0u : (s1.w ^ k3) << (KSH & 31);
                            const uint32_t d2 = KSH == 32 ? 0u : (s2.w ^ k3) << (KSH & 31);
                            const bool hit1 = ((s1.x ^ k0) | (s1.y ^ k1) | (s1.z ^ k2) | d1) == 0;
                            const bool hit2 = ((s2.x ^ k0) | (s2.y ^ k1) | (s2.z ^ k2) | d2) == 0;
                            if (nb != 1 && (hit1 || hit2)) {
                                const uint32_t swd = hit1 ? s1.w : s2.w;
                                sym = KSH == 32 ? swd : swd >> (KSH & 31);
                                done = true;
                            }
                        }
                        if (done) {
                            S[ws] = Sym<SymT>::narrow(sym);
                        } else if (HUTK_LAB_POOL_UNITS < WORD_KEY && nb > (uint32_t)HUTK_LAB_POOL_UNITS) {  // (measurement builds only)
                            atomicOr(&excm[ws >> 5], 1u << (ws & 31));
                            atomicAnd(&livem[ws >> 5], ~(1u << (ws & 31)));
                        } else {
                            atomicOr(&mergem[ws >> 5], 1u << (ws & 31));  // needs the merge loop
                        }
                    }
                    wave_sync();
                }
            }
        }
        uint32_t nW;
        const uint32_t wbase = wave_excl_scan(__popc(rest), lane, &nW);  // index of this lane's first word
        uint32_t widx = wbase;  // the index of the first of my word starts not yet handed out
#if HUTK_LAB_ALIGN
        asm volatile(".p2align " HUTK_STR(HUTK_LAB_ALIGN));
#endif
        for (uint32_t r0 = 0; r0 < nW; r0 += 64) {
            while (rest && widx - r0 < 64u) {  // (each start is visited once, in the round it belongs to)
                stage[widx - r0] = (uint16_t)(16 * lane + __builtin_ctz(rest));
                rest &= rest - 1;
                widx++;
            }
            wave_sync();
            if (r0 + lane < nW) {
                const int ws = stage[lane];
                // end of the word: the next start bit within 63 positions (bits beyond the
                // window are ones, which is only true when the data ends there)
                const uint64_t nxt = bits64(wmask32, ws + 1) & 0x7FFFFFFFFFFFFFFFull;
                const int nb = nxt ? 1 + __builtin_ctzll(nxt) : 64;
                const bool known_end = nxt != 0 && (ws + nb < NPOS || t0 + ws + nb >= A.n_bytes);
                const bool docfirst = bit_at(docm, ws + LOOKBACK);
                const uint32_t b0 = sb[ws + LOOKBACK];
                // first word of a document with a prefix configured (core.c:364-366, 421-451): a leading space
                // means "prefix ids as a word of their own, then the word as it is" (handled in the epilogue);
                // otherwise the prefix units go in front of the word's own units (arena)
                // regex pre-token path: text that no match covers is dropped (core.c:372-378 takes the LEFTMOST match at or
                // after the cursor): such a stretch is a "word" without ids
                const bool gap = A.gap_bits && ((A.gap_bits[(t0 + ws) >> 5] >> ((t0 + ws) & 31)) & 1u);
                const bool sp0 = A.alone_bits ? gbit(A.alone_bits, t0 + ws) : b0 == ' ';  // the document begins with a space
                const bool pfx = !gap && T.has_prefix && docfirst && !sp0;
                bool exc = !gap && (!known_end || nb > LANE_MAX_BYTES || (BYTE_MODE && T.has_prefix && docfirst));
                if (MULTI && !gap && !exc) {
                    // an item whose replacement has several units, or none (pretokenizer.c:102-168 emits any string): the
                    // word's units are not one per item; the exception kernels expand it (uniform branch, rare vocabularies)
                    bool m = false;
                    for (int i = 0; i < nb; i++) {
                        const uint32_t b = sb[ws + LOOKBACK + i];
                        if (BYTE_MODE || !is_cont(b)) m = m || ((T.multi_bits[b >> 5] >> (b & 31u)) & 1u);
                    }
                    exc = m;
                }
                if (!BYTE_MODE && !exc && T.has_prefix && docfirst && !pfx)  // prefix-alone ids go in front
                    exc = atomicAdd(&s_extra, (uint32_t)T.n_prefix_alone) + T.n_prefix_alone > (uint32_t)RUN_EXTRA;
                // Every global load of this round is issued here, unconditionally and together, so that the round
                // costs one memory round trip whatever mix of words the lanes hold:
                //   byte mode: symbol of the first byte (all a one-byte word needs)
                //   whole-word table: the word's raw bytes (zero padded to 16) -> symbol of the single token it
                //   encodes to; two-choice cuckoo tables, entries verified by this pipeline at context creation;
                //   one table for words of 2..14 bytes (12 with 32-bit symbols), 16-byte slots {key bytes, symbol}: one load each
                constexpr int WORD_KEY = sizeof(SymT) == 2 ? WORD_KEY_BYTES_16 : WORD_KEY_BYTES_32;
                const bool probe = !gap && !exc && !pfx && T.word_mask && nb >= 2 && nb <= WORD_KEY;
                uint32_t k0, k1, k2, k3;
                {
                    const int a = (ws + LOOKBACK) & ~3, o8 = 8 * ((ws + LOOKBACK) & 3);
                    const uint32_t* sw = reinterpret_cast<const uint32_t*>(sb + a);
                    const uint32_t q0 = sw[0], q1 = sw[1], q2 = sw[2], q3 = sw[3], q4 = sw[4];
                    k0 = funnel_r(q1, q0, o8);  // (a shift of 0 returns the low word)
                    k1 = funnel_r(q2, q1, o8);
                    k2 = funnel_r(q3, q2, o8);
                    k3 = funnel_r(q4, q3, o8);
                    // zero the bytes at and beyond nb (branch-free: two 64-bit masks; nb > 16 is never probed)
                    const uint64_t mlo = nb >= 8 ? ~0ull : ((1ull << (8 * nb)) - 1ull);
                    const uint64_t mhi = nb <= 8 ? 0ull : nb >= 16 ? ~0ull : ((1ull << (8 * (nb - 8))) - 1ull);
                    k0 &= (uint32_t)mlo;
                    k1 &= (uint32_t)(mlo >> 32);
                    k2 &= (uint32_t)mhi;
                    k3 &= (uint32_t)(mhi >> 32);
                }
                // A word of 15..28 bytes (13.. with 32-bit symbols) has the companion table: 28 key bytes and the symbol in two
                // consecutive 16-byte slots behind the main table.  Its lane loads those two INSTEAD of the main table's two
                // candidate slots -- same registers, same load instructions, other offsets --, so the round stays one
                // memory round trip and the common lanes pay two selects.  (One choice: see build_word_table.)
                const bool probe_long = !gap && !exc && !pfx && T.wordl_mask && nb > WORD_KEY && nb <= WORDL_KEY_BYTES;
                uint32_t k4 = 0, k5 = 0, k6 = 0;
                const uint32_t wh = word_hash(k0, k1, k2, k3);
                uint32_t o1 = probe ? wh & T.word_mask : 0u;  // slots to load (the table's first slot for a lane without a probe)
                uint32_t o2 = probe ? word_slot2(wh, T.word_mask) : 0u;
                if (__any(probe_long)) {  // (uniform)
                    if (probe_long) {
                        const int a = (ws + LOOKBACK) & ~3, o8 = 8 * ((ws + LOOKBACK) & 3);
                        const uint32_t* sw = reinterpret_cast<const uint32_t*>(sb + a);
                        const uint32_t q4 = sw[4], q5 = sw[5], q6 = sw[6], q7 = sw[7];
                        const int nb2 = nb - 16;  // -3 .. 12 bytes beyond the first sixteen
                        const uint64_t mlo = nb2 >= 8 ? ~0ull : nb2 <= 0 ? 0ull : ((1ull << (8 * nb2)) - 1ull);
                        const uint32_t mhi = nb2 <= 8 ? 0u : nb2 >= 12 ? ~0u : ((1u << (8 * (nb2 - 8))) - 1u);
                        k4 = funnel_r(q5, q4, o8) & (uint32_t)mlo;
                        k5 = funnel_r(q6, q5, o8) & (uint32_t)(mlo >> 32);
                        k6 = funnel_r(q7, q6, o8) & mhi;
                        o1 = T.wordl_off + 2u * (word_hash_long(k0, k1, k2, k3, k4, k5, k6) & T.wordl_mask);
                        o2 = o1 + 1u;
                    }
                }
                uint4 s1 = make_uint4(0, 0, 0, 0), s2 = s1;
                uint32_t isym = 0;
                if (T.word_mask) {  // uniform
                    s1 = reinterpret_cast<const uint4*>(T.word_tab)[o1];
                    s2 = reinterpret_cast<const uint4*>(T.word_tab)[o2];
                }
                if (BYTE_MODE) isym = T.item_sym[b0];
#if HUTK_PERTURB_MEM
                {   // MEASUREMENT ONLY: extra 16-byte gathers per word -- is the kernel bound by the L1 / L2 request rate?
                    uint32_t acc = 0;
#pragma unroll
                    for (int j = 0; j < HUTK_PERTURB_MEM; j++) {
                        const uint4 v = T.pair_buckets[(wh * (2u * j + 3u) * 0x9E3779B1u) >> T.pair_shift];
                        acc ^= v.x ^ v.w;
                    }
                    if (acc == 0x12345678u && wh == 0x9E3779B9u) raise(A.err, HUTK_E_MEMORY);
                }
#endif
                bool done = false;
                if (probe) {
                    // bitwise on purpose: with && the compiler fetches one word first and the rest only on a match
                    // (the probed word's bytes beyond WORD_KEY are zero: k3 has nothing in the symbol's place)
                    constexpr int KSH = sizeof(SymT) == 2 ? 16 : 32;  // bits of k[3] that are symbol, not key
                    const uint32_t d1 = KSH == 32 ? 0u : (s1.w ^ k3) << (KSH & 31);
                    const uint32_t d2 = KSH == 32 ? 0u : (s2.w ^ k3) << (KSH & 31);
                    const bool hit1 = ((s1.x ^ k0) | (s1.y ^ k1) | (s1.z ^ k2) | d1) == 0;
                    const bool hit2 = ((s2.x ^ k0) | (s2.y ^ k1) | (s2.z ^ k2) | d2) == 0;
                    done = hit1 || hit2;
                    const uint32_t sw = hit1 ? s1.w : s2.w;
                    if (done) S[ws] = Sym<SymT>::narrow(KSH == 32 ? sw : sw >> (KSH & 31));
                }
                if (probe_long) {
                    done = ((s1.x ^ k0) | (s1.y ^ k1) | (s1.z ^ k2) | (s1.w ^ k3) | (s2.x ^ k4) | (s2.y ^ k5) | (s2.z ^ k6)) == 0;
                    if (done) S[ws] = Sym<SymT>::narrow(s2.w);
                }
                int n = 0;
                SymT* Sdst = S + ws;
                int slot = -1;
                if (!BYTE_MODE && !exc && pfx) {
                    slot = (int)atomicAdd(&s_arena_used, 1u);
                    if (slot >= ARENA_WORDS ||
                        atomicAdd(&s_extra, (uint32_t)T.n_prefix) + T.n_prefix > (uint32_t)RUN_EXTRA) exc = true;
                    else {
                        Sdst = arenaS + slot * ARENA_W;
                        for (int i = 0; i < T.n_prefix && i < ARENA_W; i++) Sdst[i] = Sym<SymT>::narrow(T.prefix_syms[i]);
                        n = T.n_prefix;
                    }
                }
                const int n_cap = pfx ? ARENA_W : LANE_MAX_UNITS;
                if (!gap && !exc && !done) {
                    if (BYTE_MODE) {
                        n = nb;
                    } else {
                        const int lw = ws + LOOKBACK;
                        int i = 0;
                        while (i < nb) {
                            const uint32_t b = sb[lw + i];
                            int L = 1;
                            if (b >= 0x80u) {
                                L = (b >= 0xF0u) ? 4 : (b >= 0xE0u) ? 3 : (b >= 0xC0u) ? 2 : 1;
                                if (L == 1 || i + L > nb) { raise(A.err, HUTK_E_INVALID_UTF8); L = 1; }
                            }
                            uint32_t sym;
                            if ((s_item_direct[b >> 5] >> (b & 31u)) & 1u) sym = Sym<SymT>::widen(s_item_sym[b]);
                            else if (L == 1) sym = SYM_UNK;
                            else {
                                uint32_t packed = b | ((uint32_t)sb[lw + i + 1] << 8);
                                if (L > 2) packed |= (uint32_t)sb[lw + i + 2] << 16;
                                if (L > 3) packed |= (uint32_t)sb[lw + i + 3] << 24;
                                sym = char_lookup(T, packed);
                            }
                            if (n < n_cap) Sdst[n] = Sym<SymT>::narrow(sym);
                            n++;
                            i += L;
                        }
                    }
                    if (n > HUTK_LAB_POOL_UNITS) exc = true;  // (LANE_MAX_UNITS; a lower limit sends the longest pooled words to the exception kernels: measured, hutk_lab.h)
                }
                if (gap) {
                    atomicAnd(&livem[ws >> 5], ~(1u << (ws & 31)));  // no ids
                    done = true;
                } else if (done) {
                } else if (!BYTE_MODE && !exc && pfx) {  // arena word: at least two units, always through the merge loop
                    arena_ws[slot] = (uint16_t)ws;
                    arena_n[slot] = (uint16_t)n;
                    atomicAnd(&livem[ws >> 5], ~(1u << (ws & 31)));  // its ids are counted from arena_live[]
                } else if (exc) {
                    atomicOr(&excm[ws >> 5], 1u << (ws & 31));
                    atomicAnd(&livem[ws >> 5], ~(1u << (ws & 31)));
                    done = true;
                } else if (n == 1) {  // a single unit: nothing to merge
                    if (BYTE_MODE) S[ws] = Sym<SymT>::narrow(isym);
                    done = true;
                }
                if (!done) atomicOr(&mergem[ws >> 5], 1u << (ws & 31));  // needs the merge loop
            }
            wave_sync();
        }
        HUTK_STAMP(4);
        {   // exception words of my tile (excm is final here): my share of the workgroup's claim
            const uint32_t e16 = reinterpret_cast<const uint16_t*>(excm)[lane];
            if (__any(e16 != 0)) {  // (rare)
                uint32_t tot;
                (void)wave_excl_scan((uint32_t)__popc(e16), lane, &tot);
                uint32_t off = 0;
                if (lane == 0) off = atomicAdd(&s_exc_cnt, tot | (1u << 16));
                my_exc_off = (uint32_t)__builtin_amdgcn_readfirstlane((int)off);
            }
        }

    }

    // ---- 6. merge: the workgroup POOLS the merge-loop words of its WAVES tiles ------------------------
    // One lane per word, one merge step per trip, and a wavefront runs as many trips as its longest word
    // needs.  A tile has ~10 such words, a few of them long: merged by its own wavefront, most lanes would idle and
    // every tile would run the long trip count (measured, -16 %: the kernel is short of instruction issue slots, not
    // only of latency).  Pooled, the words fill one wavefront, long ones first.
    //   pool[0 .. n_long)              words with more than POOL_LONG units
    //   pool[POOL_CAP-1 downto ...]    the others; an entry is arena offset << 16 | tile-in-workgroup << 10 | word start
    // Words that do not fit stay in their tile's mergem and go into the next epoch (rare).
#if HUTK_ABLATE_MERGE
    if (tile_ok) reinterpret_cast<uint16_t*>(mergem)[lane] = 0;  // MEASUREMENT ONLY: no word is merged (wrong ids)
#endif
#if HUTK_PERTURB_SLEEP
    for (int i = 0; i < HUTK_PERTURB_SLEEP; i++) __builtin_amdgcn_s_sleep(127);  // MEASUREMENT ONLY: ~8 k cycles each, no VALU
#endif
    [[maybe_unused]] bool first_epoch = true;
    for (;;) {
#if HUTK_MERGE_STAMPS
        if (first_epoch && tile_ok && W.prof && lane == 0)  // the SIMD this wavefront runs on (HW_ID bits 5:4) in the stamp's low bits
            W.prof[tile * N_PHASE + 0] = (clock64() & ~3ll) | (long long)__builtin_amdgcn_s_getreg((1 << 11) | (4 << 6) | 4);
#endif
        if (threadIdx.x == 0) { pool_cnt[0] = 0; pool_cnt[1] = 0; pool_cnt[2] = FAST_TRIPS ? 2u : 0u; }  // (see scan_key: m[0..1] stay free)
        __syncthreads();  // (also: every wavefront is done with the automaton, whose LDS the merge arrays reuse)
        if (first_epoch && threadIdx.x == 0 && s_exc_cnt != 0) {  // (every wavefront has added its tile's count; read after the next barrier)
            const uint32_t cnt = s_exc_cnt;
            s_exc_base = atomicAdd(reinterpret_cast<unsigned long long*>(W.counters),
                                   ((unsigned long long)(cnt >> 16) << 32) | (unsigned long long)(cnt & 0xFFFFu));
        }
        if (first_epoch) HUTK_MSTAMP(1);
        uint32_t pending = 0;
        if (tile_ok) {
            pending = reinterpret_cast<const uint16_t*>(mergem)[lane];
            for (uint32_t m = pending; m; m &= m - 1) {
                const int j = __builtin_ctz(m);
                const int ws = 16 * lane + j;
                const int n = word_units(me, ws);
                const bool is_long = n > POOL_LONG;
                const uint32_t moff = atomicAdd(&pool_cnt[2], (uint32_t)n);
                if (moff + n > (uint32_t)M_ARENA) continue;  // no room in m this epoch
                const uint32_t idx = atomicAdd(&pool_cnt[is_long ? 0 : 1], 1u);
                if (idx < (uint32_t)(is_long ? POOL_LONG_CAP : POOL_CAP - POOL_LONG_CAP)) {
                    pool[is_long ? idx : POOL_CAP - 1 - idx] = (moff << 16) | (uint32_t)((wv << 10) | ws);
                    pending &= ~(1u << j);
                }
            }
            reinterpret_cast<uint16_t*>(mergem)[lane] = (uint16_t)pending;
        }
        if (first_epoch) HUTK_MSTAMP(2);
        __syncthreads();
        if (first_epoch) { HUTK_MSTAMP(3); HUTK_MSTAMP(4); }
        const uint32_t n_long = min(pool_cnt[0], (uint32_t)POOL_LONG_CAP);
        const uint32_t n_pool = n_long + min(pool_cnt[1], (uint32_t)(POOL_CAP - POOL_LONG_CAP));
        for (uint32_t base = 64u * wv; base < n_pool; base += 64u * WAVES) {
            const uint32_t wi = base + lane;
            bool have = wi < n_pool;
            const uint32_t entry = have ? (wi < n_long ? pool[wi] : pool[POOL_CAP - 1 - (wi - n_long)]) : 0u;
            Tile& X = L[(entry >> 10) & 63u];  // the word's tile
            const int ws = entry & 1023;
            SymT* Sw = X.S + ws;
            SymT* Mw = s_m.g.m + (entry >> 16);
            int arena_slot = -1;
            int n = 0;
            if (have) {
                n = word_units(X, ws);
                if (!BYTE_MODE)
                    for (int a = 0; a < ARENA_WORDS; a++)
                        if (X.arena_ws[a] == ws) {
                            Sw = X.arenaS + a * ARENA_W;
                            Mw = X.arenaM + a * ARENA_W;
                            arena_slot = a;
                        }
            }
            if constexpr (FAST_TRIPS) {
                // 16-bit symbols, rank == symbol order.  A pair is the 32-bit KEY
                // merged symbol << 5 | position: the smallest key is the pair of minimal rank, leftmost on
                // ties (queue.c:162-164), so the best pair is one register and every comparison a v_min.
                constexpr uint32_t NOKEY = 0xFFFFFFFFu;
                uint32_t live = 0, cand = 0, best = NOKEY;
                const uint32_t mw_lds = lds_addr(Mw);
                auto scan_key = [&](uint32_t c) -> uint32_t {  // four candidates per step, their LDS reads in flight together
                    uint32_t b = NOKEY;
                    while (c) {
                        // fewer than four left: the index of "no bit" is -1, whose key is all ones whatever m[offset - 1]
                        // holds (a word's stretch never starts at m[0], so that is a slot of this array)
                        const uint32_t c1 = c & (c - 1), c2 = c1 & (c1 - 1), c3 = c2 & (c2 - 1);
                        const int i0 = ffbl_raw(c), i1 = ffbl_raw(c1), i2 = ffbl_raw(c2), i3 = ffbl_raw(c3);
                        uint32_t m0, m1, m2, m3;
                        lds_read4_u16(mw_lds + 2u * i0, mw_lds + 2u * i1, mw_lds + 2u * i2, mw_lds + 2u * i3, m0, m1, m2, m3);
                        const uint32_t k0 = (m0 << 5) | (uint32_t)i0, k1 = (m1 << 5) | (uint32_t)i1,
                                       k2 = (m2 << 5) | (uint32_t)i2, k3 = (m3 << 5) | (uint32_t)i3;
                        b = min(min(b, k0), min(min(k1, k2), k3));
                        c = c3 & (c3 - 1);
                    }
                    return b;
                };
                auto publish = [&]() {  // the surviving units (unit 0 is in livem already); arena words keep theirs apart
                    if (!BYTE_MODE && arena_slot >= 0) {
                        X.arena_live[arena_slot] = live;
                        return;
                    }
                    const uint64_t lm = (uint64_t)(live & ~1u) << (ws & 31);
                    if ((uint32_t)lm) atomicOr(&X.livem[ws >> 5], (uint32_t)lm);
                    if ((uint32_t)(lm >> 32)) atomicOr(&X.livem[(ws >> 5) + 1], (uint32_t)(lm >> 32));
                };
                if (have && !BYTE_MODE) {
                    // the symbols are in place (phase 5); the pair results of neighbours: four lookups (eight loads) in flight
                    live = (n >= 32) ? 0xFFFFFFFFu : ((1u << n) - 1u);
                    for (int i0 = 0; i0 + 1 < n; i0 += 4) {
                        PairProbe pr[4];
                        uint32_t sy[5];
#pragma unroll
                        for (int j = 0; j < 5; j++) sy[j] = (i0 + j < n) ? (uint32_t)Sw[i0 + j] : 0u;
#pragma unroll
                        for (int j = 0; j < 4; j++) pr[j] = pair_issue(T, sy[j], sy[j + 1]);
#pragma unroll
                        for (int j = 0; j < 4; j++) {
                            const int i = i0 + j;
                            if (i + 1 < n) {
                                const uint32_t m = pair_resolve(T, pr[j], sy[j], sy[j + 1]);
                                Mw[i] = (SymT)m;
                                if (m != SYM_NONE) cand |= 1u << i;
                            }
                        }
                    }
                    best = scan_key(cand);
                }
                if (have && BYTE_MODE) {
                    live = (n >= 32) ? 0xFFFFFFFFu : ((1u << n) - 1u);
                    // Set-up, eight units per step and no branch per unit: the word's bytes come out of LDS as three
                    // aligned dwords; two consecutive bytes are the index of the (byte, next byte) table, whose
                    // entry is {symbol of the byte, merged symbol of the pair}.  Units beyond the word are
                    // looked up all the same and stored to a dummy slot.
                    SymT* const dummy = reinterpret_cast<SymT*>(me.stage) + lane;  // the staging buffer is idle in this phase
                    const int li = ws + LOOKBACK;
                    const uint32_t* bp = reinterpret_cast<const uint32_t*>(T.bytepair);
                    for (int i0 = 0; i0 < n; i0 += 8) {
                        const int a = (li + i0) & ~3, o8 = 8 * ((li + i0) & 3);
                        const uint32_t* sw = reinterpret_cast<const uint32_t*>(X.sb + a);
                        const uint32_t q0 = sw[0], q1 = sw[1], q2 = sw[2];
                        const uint64_t lo = (uint64_t)funnel_r(q1, q0, o8) | ((uint64_t)funnel_r(q2, q1, o8) << 32);
                        const uint32_t k2 = q2 >> o8;  // its low byte is byte 8 of the stretch
                        uint32_t e[8];
#pragma unroll
                        for (int j = 0; j < 8; j++) {
                            const uint32_t idx = j < 7 ? (uint32_t)(lo >> (8 * j)) & 0xFFFFu
                                                       : ((uint32_t)(lo >> 56) | ((k2 & 0xFFu) << 8));
                            e[j] = bp[idx];
                        }
                        uint32_t cb = 0;
#pragma unroll
                        for (int j = 0; j < 8; j++) {
                            const bool in = i0 + j < n;
                            SymT* ds = in ? Sw + i0 + j : dummy;
                            SymT* dm = in ? Mw + i0 + j : dummy;
                            *ds = (SymT)e[j];
                            *dm = (SymT)(e[j] >> 16);
                            cb |= (e[j] < 0xFFFF0000u ? 1u : 0u) << j;
                        }
                        cand |= cb << i0;
                    }
                    cand &= (1u << (n - 1)) - 1u;  // the last unit has no next one (n >= 2)
                    best = scan_key(cand);
                }
                if (first_epoch && base == 64u * wv) HUTK_MSTAMP(4);
                // One merge per trip: apply the best pair, ISSUE the lookups of the two new neighbour pairs, rescan the
                // untouched candidates while those loads fly, then fold the two new keys in.
                const bool mine = have;
#if HUTK_MERGE_STAMPS
                long long ts_issue = 0, ts_scan = 0, ts_resolve = 0, ts_trips = 0;
#endif
                // A lookup that must go on in the pair's SECOND bucket (a filter bit of the first one says so; under 1 % of
                // the lookups) is not followed up inside the trip: with 64 lanes and two lookups each, nearly every trip
                // had some lane in that case, and every lane paid its extra round trip(s).  The lane remembers which of its
                // two lookups it was (`again`) and REPEATS both in the next trip, from the buckets they need, beside the
                // other lanes' ordinary ones: its merge is applied already, so the same code finds the same neighbours.
                uint32_t again = 0;  // bit 0 / 1: the right / left lookup of the merge at p goes to its second bucket this trip
                int p = 0;
#if HUTK_LAB_ALIGN
                asm volatile(".p2align " HUTK_STR(HUTK_LAB_ALIGN));
#endif
                for (;;) {
#if HUTK_MERGE_STAMPS
                    const long long tt0 = clock64();
                    long long tt1 = tt0, tt2 = tt0;
#endif
                    have = have && (best != NOKEY || again != 0);
                    if (!__any(have)) break;
                    if (have) {
                        uint32_t merged;
                        if (again == 0) {
                            p = (int)(best & 31u);
                            merged = best >> 5;
                            const uint32_t above = live & ~((2u << p) - 1u);  // not empty: bit p of cand was set
                            const int q = __builtin_ctz(above);               // the unit the merge consumes
                            Sw[p] = (SymT)merged;
                            live &= ~(1u << q);
                            cand &= ~((1u << q) | (1u << p));
                        } else {
                            merged = Sw[p];  // the merge is applied: the same code finds the same neighbours
                        }
                        const uint32_t rmask = live & ~((2u << p) - 1u);  // live units after p (the consumed one is gone)
                        const uint32_t lmask = live & ((1u << p) - 1u);   // live units before p: none iff p == 0
                        const int q2 = __builtin_ctz(rmask | 0x80000000u);
                        const int p0 = 31 - __builtin_clz(lmask | 1u);    // == p when there is none
                        const uint32_t sr = Sw[q2], sl = Sw[p0];          // (read and looked up even when absent)
                        const uint32_t t1 = pair_mix(merged, sr), t2 = pair_mix(sl, merged);
                        uint32_t b1 = pair_bucket1(t1, T.pair_shift), b2 = pair_bucket1(t2, T.pair_shift);
                        if (again != 0) {  // (rare, and only the repeating lanes)
                            if (again & 1u) b1 = pair_bucket2(t1, T.pair_shift);
                            if (again & 2u) b2 = pair_bucket2(t2, T.pair_shift);
                        }
                        const uint4 e1 = T.pair_buckets[b1], e2 = T.pair_buckets[b2];
#if HUTK_MERGE_STAMPS
                        tt1 = clock64();
#endif
                        if (again == 0) {  // (a repeating lane's rescan is done)
                            cand &= ~(1u << p0);
                            best = scan_key(cand);
                        }
#if HUTK_MERGE_STAMPS
                        tt2 = clock64();
#endif
                        const uint32_t y1 = pair_match(e1, merged | ((sr & 0xFFFu) << 20), sr >> 12);
                        const uint32_t y2 = pair_match(e2, sl | ((merged & 0xFFFu) << 20), merged >> 12);
                        const uint32_t f1 = (e1.y >> 28) | ((e1.w >> 28) << 4), f2 = (e2.y >> 28) | ((e2.w >> 28) << 4);
                        const bool need1 = rmask != 0 && y1 == 0xFFFFFFFFu && !(again & 1u) && ((f1 >> (t1 & 7u)) & 1u);
                        const bool need2 = lmask != 0 && y2 == 0xFFFFFFFFu && !(again & 2u) && ((f2 >> (t2 & 7u)) & 1u);
                        if (need1 || need2) {
                            again |= (need1 ? 1u : 0u) | (need2 ? 2u : 0u);
                        } else {
                            again = 0;
                            uint32_t mr = (y1 >> 8) & 0xFFFFFu, ml = (y2 >> 8) & 0xFFFFFu;
                            mr = (rmask != 0 && mr != PAIR_ABSENT) ? mr : SYM_NONE;
                            ml = (lmask != 0 && ml != PAIR_ABSENT) ? ml : SYM_NONE;
                            Mw[p0] = (SymT)ml;  // first: without a left neighbour p0 == p
                            Mw[p] = (SymT)mr;
                            const bool hr = mr != SYM_NONE, hl = ml != SYM_NONE;
                            cand |= ((hr ? 1u : 0u) << p) | ((hl ? 1u : 0u) << p0);
                            const uint32_t kr = hr ? ((mr << 5) | (uint32_t)p) : NOKEY;
                            const uint32_t kl = hl ? ((ml << 5) | (uint32_t)p0) : NOKEY;
                            best = min(best, min(kr, kl));
                        }
                    }
#if HUTK_MERGE_STAMPS
                    {   // lane 0's view of the trip (it holds the pool's longest word)
                        const long long tt3 = clock64();
                        ts_issue += tt1 - tt0; ts_scan += tt2 - tt1; ts_resolve += tt3 - tt2; ts_trips++;
                    }
#endif
                }
#if HUTK_MERGE_STAMPS
                if (first_epoch && base == 64u * wv && tile_ok && W.prof && lane == 0) {
                    W.prof[tile * N_PHASE + 7] = ts_issue; W.prof[tile * N_PHASE + 8] = ts_scan;
                    W.prof[tile * N_PHASE + 9] = ts_resolve | (ts_trips << 40);
                }
#endif
                if (mine) {  // publish the surviving units (unit 0 is in livem already)
                    publish();
                }
                wave_sync();
            } else {
                uint32_t live = 0, cand = 0;  // lane words have at most 32 units
                uint32_t br = 0xFFFFFFFFu;
                int bp = 0;
                SymT bm = 0;
                if (have) {
                    live = (n >= 32) ? 0xFFFFFFFFu : ((1u << n) - 1u);
                    if (BYTE_MODE) {
                        // symbols and initial pair results, one load per unit from the 65536-entry
                        // (byte, next byte) table
                        const uint8_t* wb = X.sb + ws + LOOKBACK;
                        for (int i0 = 0; i0 < n; i0 += 16) {
                            typename Sym<SymT>::Pair e[16];
#pragma unroll
                            for (int j = 0; j < 16; j++) {  // 16 independent loads in flight
                                const int i = i0 + j;
                                const uint32_t b = (i < n) ? wb[i] : 0u, b2 = (i + 1 < n) ? wb[i + 1] : 0u;
                                e[j] = reinterpret_cast<const typename Sym<SymT>::Pair*>(T.bytepair)[b | (b2 << 8)];
                            }
#pragma unroll
                            for (int j = 0; j < 16; j++) {
                                const int i = i0 + j;
                                if (i < n) {
                                    const SymT mv = Sym<SymT>::pair_merged(e[j]);
                                    Sw[i] = Sym<SymT>::pair_sym(e[j]);
                                    Mw[i] = mv;
                                    if (i + 1 < n && mv != Sym<SymT>::NONE) cand |= 1u << i;
                                }
                            }
                        }
                    } else {
                        for (int i0 = 0; i0 + 1 < n; i0 += 4) {  // four lookups (eight loads) in flight
                            PairProbe pr[4];
                            uint32_t sl[5];
#pragma unroll
                            for (int j = 0; j < 5; j++) sl[j] = (i0 + j < n) ? Sym<SymT>::widen(Sw[i0 + j]) : 0u;
#pragma unroll
                            for (int j = 0; j < 4; j++) pr[j] = pair_issue(T, sl[j], sl[j + 1]);
#pragma unroll
                            for (int j = 0; j < 4; j++) {
                                const int i = i0 + j;
                                if (i + 1 < n) {
                                    const uint32_t m = pair_resolve(T, pr[j], sl[j], sl[j + 1]);
                                    Mw[i] = Sym<SymT>::narrow(m);
                                    if (m != SYM_NONE) cand |= 1u << i;
                                }
                            }
                        }
                    }
                    scan_best(cand, Mw, br, bp, bm);
                }
                // Each lane keeps (br, bp, bm) = rank, position and merged symbol of its word's best pair.
                // A trip applies that merge, ISSUES the pair-table loads for the two new neighbour pairs,
                // rescans the untouched candidates in LDS while those loads are in flight, and then picks
                // the next best among {rescan, new left pair, new right pair}.
                for (;;) {
                    if (have && cand == 0) {
                        // done: publish the surviving units (unit 0 always survives and is already in livem)
                        if (!BYTE_MODE && arena_slot >= 0) {
                            X.arena_live[arena_slot] = live;
                        } else {
                            const uint64_t lm = (uint64_t)(live & ~1u) << (ws & 31);
                            if ((uint32_t)lm) atomicOr(&X.livem[ws >> 5], (uint32_t)lm);
                            if ((uint32_t)(lm >> 32)) atomicOr(&X.livem[(ws >> 5) + 1], (uint32_t)(lm >> 32));
                        }
                        have = false;
                    }
                    if (!__any(have)) break;
                    if (have) {
                        const int p = bp;
                        const uint32_t merged = Sym<SymT>::widen(bm);
                        // q: next live unit after p (exists: bit p of cand was set)
                        const uint32_t above = live & ~((2u << p) - 1u);
                        const int q = __builtin_ctz(above);
                        Sw[p] = bm;
                        live &= ~(1u << q);
                        cand &= ~((1u << q) | (1u << p));
                        const uint32_t right = above & (above - 1u);    // live units after q
                        const uint32_t left = live & ((1u << p) - 1u);  // live units before p
                        const int q2 = right ? __builtin_ctz(right) : 0;
                        const int p0 = left ? 31 - __builtin_clz(left) : 0;
                        const uint32_t sr = right ? Sym<SymT>::widen(Sw[q2]) : 0u;
                        const uint32_t sl = left ? Sym<SymT>::widen(Sw[p0]) : 0u;
                        // issue the lookups of both new pairs: four independent loads (unconditional: a
                        // missing neighbour reads as symbol 0, and the result is discarded)
                        const PairProbe s1 = pair_issue(T, merged, sr);
                        const PairProbe s2 = pair_issue(T, sl, merged);
                        // rescan what the merge did not touch
                        if (left) cand &= ~(1u << p0);
                        br = 0xFFFFFFFFu;
                        scan_best(cand, Mw, br, bp, bm);
                        // the two new pairs
                        const uint32_t mr = right ? pair_resolve(T, s1, merged, sr) : SYM_NONE;
                        const uint32_t ml = left ? pair_resolve(T, s2, sl, merged) : SYM_NONE;
                        if (right) {
                            const SymT mn = Sym<SymT>::narrow(mr);
                            Mw[p] = mn;
                            if (mr != SYM_NONE) {
                                cand |= 1u << p;
                                const uint32_t r = RK(mr);
                                if (r < br || (r == br && p < bp)) {
                                    br = r;
                                    bp = p;
                                    bm = mn;
                                }
                            }
                        }
                        if (left) {
                            const SymT mn = Sym<SymT>::narrow(ml);
                            Mw[p0] = mn;
                            if (ml != SYM_NONE) {
                                cand |= 1u << p0;
                                const uint32_t r = RK(ml);
                                if (r < br || (r == br && p0 < bp)) {
                                    br = r;
                                    bp = p0;
                                    bm = mn;
                                }
                            }
                        }
                    }
                }
                wave_sync();
            }
        }
        if (first_epoch) HUTK_MSTAMP(5);
        const bool again = __syncthreads_or(pending != 0);
        if (first_epoch) HUTK_MSTAMP(6);
        first_epoch = false;
        if (!again) break;
    }
    if (tile_ok) HUTK_STAMP(5);

    if (!ONE && !tile_ok) return;
    if (tile_ok) {
    // ---- 7. per-position epilogue: counts -> scan -> symbols out, exception records ----
    // A lane's ids are the surviving units at ITS 16 positions (whichever word they belong to: units sit
    // inside their word's byte span, so position order is id order), plus -- non-byte mode with a prefix
    // only -- the ids of arena words and the prefix-alone ids, both counted at their word's start.
    uint16_t* lanepref = stage;  // the staging buffer is free now: ids before lane l's positions
    const uint32_t live16 = reinterpret_cast<const uint16_t*>(livem)[lane];
    const uint32_t exc16 = reinterpret_cast<const uint16_t*>(excm)[lane];
    constexpr bool PREFIXED = !BYTE_MODE;  // arena words and prefix-alone ids exist in this mode only
    auto alone_ids = [&](int ws) -> uint32_t {  // prefix encoded as a word of its own (core.c:421-446)
        if (!PREFIXED || !T.has_prefix) return 0u;
        if (!bit_at(docm, ws + LOOKBACK)) return 0u;
        return (A.alone_bits ? gbit(A.alone_bits, t0 + ws) : sb[ws + LOOKBACK] == ' ') ? (uint32_t)T.n_prefix_alone : 0u;
    };
    auto arena_at = [&](int ws) -> int {
        if (PREFIXED)
            for (int a = 0; a < ARENA_WORDS; a++)
                if (arena_ws[a] == ws) return a;
        return -1;
    };
    // extra ids (arena words, prefix-alone) of word starts in `starts` (bits of lane lr's 16 positions)
    auto extra_ids = [&](int lr, uint32_t starts) -> uint32_t {
        uint32_t x = 0;
        if (PREFIXED && T.has_prefix)
            // (only the words that begin a document: one 16-bit slice of the document-start bitmap instead of a test per word)
            for (uint32_t m = starts & (uint32_t)bits64(docm, 16 * lr + LOOKBACK) & 0xFFFFu; m; m &= m - 1) {
                const int ws = 16 * lr + __builtin_ctz(m);
                if (bit_at(excm, ws)) continue;
                const int a = arena_at(ws);
                x += a >= 0 ? (uint32_t)__popc(arena_live[a]) : alone_ids(ws);
            }
        return x;
    };
    // one scan for two counts: ids (bits 0-10, at most RUN_STRIDE), exception words (11-20)
    uint32_t mine = (uint32_t)__popc(live16) + extra_ids(lane, own) + ((uint32_t)__popc(exc16) << 11);
    uint32_t total;
    uint32_t run = wave_excl_scan(mine, lane, &total);
    lanepref[lane] = (uint16_t)(run & 0x7FFu);
    const uint32_t n_dense = total & 0x7FFu, n_exc = (total >> 11) & 0x3FFu;
    static_assert(RUN_STRIDE < 2048 && TILE_BYTES < 1024, "count fields");
    uint32_t exc_first = 0;
    if (lane == 0) {
        if (n_exc) {
            // the tile's exception records (counters[0]) and its place on k_gather_exc's work list (counters[1]): my share
            // of what the workgroup claimed with one 64-bit atomic
            const unsigned long long old = s_exc_base;
            exc_first = (uint32_t)old + (my_exc_off & 0xFFFFu);
            W.exc_tiles[(uint32_t)(old >> 32) + (my_exc_off >> 16)] = (uint32_t)tile;
        }
        W.tile_count[tile] = n_dense;
        W.tile_dense[tile] = n_dense;
        W.tile_run_start[tile] = 0;
        W.tile_exc_first[tile] = exc_first;
        W.tile_nexc[tile] = n_exc;
    }
    exc_first = __shfl(exc_first, 0, 64);
    HUTK_STAMP(6);
    // symbols in the width the LDS arrays use (k_finish widens them and turns them into ids), stored lane by lane
    // (collected in LDS first and stored as 16-byte vectors: measured no faster)
    SymT* run_out = reinterpret_cast<SymT*>(W.run) + tile * RUN_STRIDE;
    auto emit_run = [&](SymT* dst) {
        uint32_t pos = run & 0x7FFu, eidx = (run >> 11) & 0x3FFu;
        uint32_t ev = live16 | exc16;
        if (PREFIXED && T.has_prefix) ev |= own;  // arena words have no live bit of their own
        for (; ev; ev &= ev - 1) {
            const int j = __builtin_ctz(ev);
            const int ws = 16 * lane + j;
            if ((exc16 >> j) & 1u) {
                const uint64_t slot = (uint64_t)exc_first + eidx;
                if ((int64_t)slot < W.cap_exc) {
                    const uint64_t nxt = bits64(wmask32, ws + 1) & 0x7FFFFFFFFFFFFFFFull;
                    int nb = nxt ? 1 + __builtin_ctzll(nxt) : 64;
                    bool known_end = nxt != 0 && (ws + nb < NPOS || t0 + ws + nb >= A.n_bytes);
                    if (nxt == 0) {
                        // more than 63 bytes: the word may still end inside this tile's 1024 classified positions (a word
                        // of a hundred letters mostly does): then its length is known here and d_exc_ends, which would
                        // stage and classify the text again, has nothing to do for it
                        int e = -1;
                        for (int k = (ws + 64) >> 5; k < NPOS / 32 && e < 0; k++) {
                            uint32_t m = wmask32[k];
                            if (k == (ws + 64) >> 5) m &= ~0u << ((ws + 64) & 31);
                            if (m) e = 32 * k + __builtin_ctz(m);
                        }
                        if (e >= 0) { nb = e - ws; known_end = true; }
                    }
                    ExcRec rec;
                    rec.ws = t0 + ws;
                    rec.tok_base = 0;
                    rec.out_pos = 0;
                    rec.len = known_end ? nb : -1;
                    rec.wpos = pos;
                    rec.cnt = 0;
                    rec.tile = (uint32_t)tile;
                    W.exc[slot] = rec;
                } else {
                    raise(A.err, HUTK_E_MEMORY);
                }
                eidx++;
                continue;
            }
            if (PREFIXED && T.has_prefix && ((own >> j) & 1u) && bit_at(docm, ws + LOOKBACK)) {
                const int a = arena_at(ws);
                if (a >= 0) {
                    for (uint32_t sv = arena_live[a]; sv; sv &= sv - 1)
                        dst[pos++] = arenaS[a * ARENA_W + __builtin_ctz(sv)];
                    continue;
                }
                const uint32_t na = alone_ids(ws);
                for (uint32_t i = 0; i < na; i++) dst[pos++] = Sym<SymT>::narrow(T.prefix_alone_syms[i]);
            }
            if ((live16 >> j) & 1u) dst[pos++] = S[ws];  // stores only: nothing here waits
        }
    };
    emit_run(run_out);
    wave_sync();
    HUTK_STAMP(7);

    // ... and before the tile's last start of the reference's own, when none follows in the halo: that is where k_cut
    // cuts a document if the word turns out to be over-long (rare: a word of 64 bytes and more)
    if (const uint32_t cp = me.cutpos; cp != 0) {  // (rare)
        if (lane == 0) {
            const int pos = (int)cp - 1;
            const int lr = pos >> 4;
            const uint32_t below = (1u << (pos & 15)) - 1u;
            uint32_t before = lanepref[lr];
            before += (uint32_t)__popc(reinterpret_cast<const uint16_t*>(livem)[lr] & below);
            before += extra_ids(lr, wmask16[lr] & below);
            W.tile_lastreal[tile] = (uint32_t)pos | (before << 16);
        }
    }
    // ---- 8. ids emitted before each document that starts in this tile ----------
    for (int64_t d = dfirst + lane; d <= A.n_docs; d += 64) {
        const int64_t o = A.offsets[d];  // (keeping the first ones from the start in registers: measured, no faster)
        if (o >= tile_end) break;
        if (o < t0) continue;
        const int r = (int)(o - t0);
        const int lr = r >> 4;
        const uint32_t below = (1u << (r & 15)) - 1u;
        uint32_t before = lanepref[lr];
        before += (uint32_t)__popc(reinterpret_cast<const uint16_t*>(livem)[lr] & below);
        before += extra_ids(lr, wmask16[lr] & below);
        W.doc_tile_pos[d] = before;
    }
    HUTK_STAMP(8);
    HUTK_STAMP(9);
    }
    if constexpr (ONE) {
        // ---- 9. the rest of the pipeline (k_scan, k_finish), by this workgroup ----
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");  // (the tiles' runs and counts, written by the four wavefronts, read below by other lanes)
        __syncthreads();
        if (__hip_atomic_load(&W.counters[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
            // exception words: their kernels are not in this launch
            if (threadIdx.x == 0) __hip_atomic_store(W.one_flag, 2, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            return;
        }
        if (threadIdx.x < 64) {  // exclusive scan of the tiles' id counts (at most WAVES <= 64)
            uint32_t total;
            const uint32_t mine = threadIdx.x < A.n_tiles ? W.tile_count[threadIdx.x] : 0u;
            const uint32_t before = wave_excl_scan(mine, (int)threadIdx.x, &total);
            if (threadIdx.x < A.n_tiles) W.tile_base[threadIdx.x] = before;
            if (threadIdx.x == 0) W.tile_base[A.n_tiles] = total;
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
        __syncthreads();
        static_assert(WAVES <= 4, "one wavefront copies the runs of four tiles");
        if (wv == 0) d_gather<SymT>(T, A, W, 0);
        for (int64_t vb = 0; vb * (64 * WAVES) <= A.n_docs; vb++) d_doc_off(A, W, vb);
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "");  // system scope: ids, offsets, status and error word are in the caller's host memory
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(W.one_flag, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
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

// minimum over the wavefront, in every lane: DPP row shifts and broadcasts (the scan's pattern; a lane without a source
// reads all ones), then lane 63's value.  (Twelve ds_bpermute round trips through the LDS pipe before: a merge of a long
// exception word does four of these reductions.)
__device__ __forceinline__ uint64_t wave_min_u64(uint64_t v) {
    auto step = [&](auto ctrl, auto rows) {
        const uint32_t oh = (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)(uint32_t)(v >> 32), decltype(ctrl)::value, decltype(rows)::value, 0xf, false);
        const uint32_t ol = (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)(uint32_t)v, decltype(ctrl)::value, decltype(rows)::value, 0xf, false);
        const uint64_t o = ((uint64_t)oh << 32) | ol;
        v = o < v ? o : v;
    };
    step(std::integral_constant<int, 0x111>{}, std::integral_constant<int, 0xf>{});
    step(std::integral_constant<int, 0x112>{}, std::integral_constant<int, 0xf>{});
    step(std::integral_constant<int, 0x114>{}, std::integral_constant<int, 0xf>{});
    step(std::integral_constant<int, 0x118>{}, std::integral_constant<int, 0xf>{});
    step(std::integral_constant<int, 0x142>{}, std::integral_constant<int, 0xa>{});
    step(std::integral_constant<int, 0x143>{}, std::integral_constant<int, 0xc>{});
    const uint32_t h = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), 63);
    const uint32_t l = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, 63);
    return ((uint64_t)h << 32) | l;
}

// Cooperative merge of n symbols held in Sa (pairs in Ma) by one wavefront.
// Dense arrays: a merge removes element p+1 by shifting the tail left.
template <class Arr>
__device__ __forceinline__ int64_t bpe_wave(const DevTables& T, Arr Sa, Arr Ma, int64_t n, int lane) {
    for (int64_t i = lane; i < n; i += 64)
        Ma.set(i, (i + 1 < n) ? pair_lookup(T, Sa.get(i), Sa.get(i + 1)) : SYM_NONE);
    wave_wg_sync();
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
        wave_wg_sync();
        for (int64_t base = p + 1; base + 1 < n; base += 64) {
            const int64_t i = base + lane;
            uint32_t s = 0, m = 0;
            const bool on = i + 1 < n;
            if (on) {
                s = Sa.get(i + 1);
                m = Ma.get(i + 1);
            }
            wave_wg_sync();
            if (on) {
                Sa.set(i, s);
                Ma.set(i, m);
            }
            wave_wg_sync();
        }
        n -= 1;
        if (lane == 0) {
            Sa.set(p, merged);
            Ma.set(p, has_right ? pair_lookup(T, merged, sr) : SYM_NONE);
        }
        if (lane == 1 && has_left) Ma.set(p - 1, pair_lookup(T, sl, merged));
        wave_wg_sync();
    }
    return n;
}

#ifndef HUTK_EXC_CHUNK
#define HUTK_EXC_CHUNK 256
#endif
constexpr int EXC_CHUNK = HUTK_EXC_CHUNK;                    // positions examined per step when a word end is unknown
constexpr int EXC_WIN = 16 + EXC_CHUNK + 16;      // staged bytes per step

// The same merge rule for words too long for the LDS arrays (up to MAX_WORD_BYTES units), in time
// O(merges x chunk) instead of O(merges x n): units stay where they are (a consumed unit is marked
// dead), and the minimum over all pairs comes from a two-level structure -- per chunk of CH units the
// best (rank, position) key in LDS (L1r/L1p, at most 1024 chunks), the global best by a wave reduction
// over those.  A merge touches three pair results, so three chunks are rescanned.  The survivors are
// compacted to the front at the end.  Sg/Mg are this word's regions of the exception arrays in HBM.
constexpr uint32_t UNIT_DEAD = 0xFFFFFFFEu;
#ifndef HUTK_EXC_SHIFT_MAX
#define HUTK_EXC_SHIFT_MAX 128
#endif
constexpr int64_t EXC_SHIFT_MAX = HUTK_EXC_SHIFT_MAX;  // longest word in LDS that d_exc merges by shifting (bpe_wave)
template <class Arr>
__device__ __forceinline__ int64_t bpe_wave_big(const DevTables& T, Arr Sg, Arr Mg, uint32_t* L1r, uint32_t* L1p, int64_t n,
                                int lane) {
    const int64_t CH = (((n + EXC_LDS_UNITS - 1) / EXC_LDS_UNITS) + 63) & ~(int64_t)63;  // units per chunk
    const int NC = (int)((n + CH - 1) / CH);                                               // <= 1024
    for (int64_t i = lane; i < n; i += 64)
        Mg.set(i, (i + 1 < n) ? pair_lookup(T, Sg.get(i), Sg.get(i + 1)) : SYM_NONE);
    wave_wg_sync();
    auto rescan = [&](int64_t c) {  // whole wavefront: best key of chunk c -> L1
        const int64_t lo = c * CH, hi = (lo + CH < n) ? lo + CH : n;
        uint64_t best = ~0ull;
        for (int64_t i = lo + lane; i < hi; i += 64) {
            const uint32_t m = Mg.get(i);
            if (m != SYM_NONE) {
                const uint64_t k = ((uint64_t)rank_of(T, m) << 32) | (uint64_t)i;
                best = k < best ? k : best;
            }
        }
        best = wave_min_u64(best);
        if (lane == 0) {
            L1r[c] = (uint32_t)(best >> 32);
            L1p[c] = (uint32_t)best;
        }
    };
    for (int c = 0; c < NC; c++) rescan(c);
    wave_wg_sync();
    // first live unit at or after `from` (-1: none); last live unit at or before `from` (-1: none)
    auto next_live = [&](int64_t from) -> int64_t {
        for (int64_t base = from; base < n; base += 64) {
            const int64_t i = base + lane;
            const unsigned long long bal = __ballot(i < n && Sg.get(i) != UNIT_DEAD);
            if (bal) return base + __builtin_ctzll(bal);
        }
        return -1;
    };
    auto prev_live = [&](int64_t from) -> int64_t {
        for (int64_t base = from; base >= 0; base -= 64) {
            const int64_t i = base - lane;
            const unsigned long long bal = __ballot(i >= 0 && Sg.get(i) != UNIT_DEAD);
            if (bal) return base - __builtin_ctzll(bal);
        }
        return -1;
    };
    for (;;) {
        uint64_t best = ~0ull;
        for (int c = lane; c < NC; c += 64) {
            const uint64_t k = ((uint64_t)L1r[c] << 32) | (uint64_t)L1p[c];
            best = k < best ? k : best;
        }
        best = wave_min_u64(best);
        if (best == ~0ull) break;
        const int64_t p = (int64_t)(best & 0xFFFFFFFFull);
        const uint32_t merged = Mg.get(p);
        const int64_t q = next_live(p + 1);  // the unit the merge consumes (exists: the pair was a candidate)
        const int64_t q2 = next_live(q + 1), p0 = prev_live(p - 1);
        const uint32_t sr = q2 >= 0 ? Sg.get(q2) : 0u, sl = p0 >= 0 ? Sg.get(p0) : 0u;
        wave_wg_sync();
        if (lane == 0) {
            Sg.set(p, merged);
            Sg.set(q, UNIT_DEAD);
            Mg.set(q, SYM_NONE);
            Mg.set(p, q2 >= 0 ? pair_lookup(T, merged, sr) : SYM_NONE);
        }
        if (lane == 1 && p0 >= 0) Mg.set(p0, pair_lookup(T, sl, merged));
        wave_wg_sync();
        const int64_t cp = p / CH, cq = q / CH, c0 = p0 >= 0 ? p0 / CH : cp;
        rescan(cp);
        if (cq != cp) rescan(cq);
        if (c0 != cp) rescan(c0);
        wave_wg_sync();
    }
    // compaction of the survivors to the front, 64 units at a time (writes never pass the reads)
    int64_t out = 0;
    for (int64_t base = 0; base < n; base += 64) {
        const int64_t i = base + lane;
        const uint32_t sym = i < n ? Sg.get(i) : UNIT_DEAD;
        const bool live = sym != UNIT_DEAD;
        const unsigned long long bal = __ballot(live);
        wave_wg_sync();
        if (live) Sg.set(out + __popcll(bal & ((1ull << lane) - 1ull)), sym);
        out += __popcll(bal);
        wave_wg_sync();
    }
    return out;
}


// minimum over the wavefront of a 32-bit key, in every lane (the scan's DPP pattern; lane 63's value read back)
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)v, 0x111, 0xf, 0xf, false));
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)v, 0x112, 0xf, 0xf, false));
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)v, 0x114, 0xf, 0xf, false));
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)v, 0x118, 0xf, 0xf, false));
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)v, 0x142, 0xa, 0xf, false));
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)v, 0x143, 0xc, 0xf, false));
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

// The merge rule for ONE word of 2..EXC_LDS_UNITS units held in LDS, by one wavefront -- the form for vocabularies whose
// rank is the symbol order (T.rank_is_sym); round 4: about half the time per merge of bpe_wave_big, which it replaces there.
//   Sl[i] = symbol (20 bits) | index of the NEXT live unit << 20 (11 bits, FL_NONE: none) | dead << 31
//   Ml[i] = merged symbol of (unit i, next live unit) (20 bits, PAIR_ABSENT: no rank) | index of the PREVIOUS live unit << 20
//   l1[c] = smallest key  merged symbol << 10 | index  among the 64 units of chunk c (all ones: none)
// A merge: the smallest key over the chunks (one read, one DPP reduction); its neighbours by following the links -- three
// dependent LDS reads that every lane makes at the same address, where bpe_wave_big scanned for live units with ballots;
// lanes 0 and 1 ask the pair table for the two new pairs; while those loads fly the (at most three) chunks the merge touched
// are searched again without the entries that are about to change, which are folded in when the loads are back.  No workgroup
// barrier: one wavefront, LDS in program order.
constexpr uint32_t FL_NONE = 0x7FFu, FL_SYM = 0xFFFFFu, FL_DEAD = 0x80000000u;
constexpr int FAST_LDS_UNITS = 2046;  // eleven bits of position in the key and in the links, 0x7FF means "none" (d_exc<2048>)
__device__ __forceinline__ int64_t bpe_wave_fast(const DevTables& T, uint32_t* Sl, uint32_t* Ml, uint32_t* l1, int n, int lane) {
    constexpr uint32_t NOKEY = 0xFFFFFFFFu;
    const int NC = (n + 63) >> 6;
    // links and the pair results of neighbours, every unit at once
    for (int i0 = 0; i0 < n; i0 += 64) {
        const int i = i0 + lane;
        uint32_t s0 = 0, s1 = 0;
        if (i < n) s0 = Sl[i] & FL_SYM;
        if (i + 1 < n) s1 = Sl[i + 1] & FL_SYM;
        wave_sync();  // (every lane has read its neighbour's plain symbol before anybody adds the link bits)
        if (i < n) {
            const uint32_t m = (i + 1 < n) ? pair_lookup(T, s0, s1) : SYM_NONE;
            Sl[i] = s0 | ((i + 1 < n ? (uint32_t)(i + 1) : FL_NONE) << 20);
            Ml[i] = (m == SYM_NONE ? PAIR_ABSENT : m) | ((i > 0 ? (uint32_t)(i - 1) : FL_NONE) << 20);
        }
    }
    wave_sync();
    auto chunk_key = [&](int c, uint32_t skip_a, uint32_t skip_b) -> uint32_t {  // smallest key of chunk c, two positions left out
        const uint32_t i = (uint32_t)(64 * c + lane);
        uint32_t k = NOKEY;
        if ((int)i < n && i != skip_a && i != skip_b) {
            const uint32_t m = Ml[i] & FL_SYM;
            if (m != PAIR_ABSENT) k = (m << 11) | i;
        }
        return wave_min_u32(k);
    };
    for (int c = 0; c < NC; c++) {
        const uint32_t k = chunk_key(c, NOKEY, NOKEY);
        if (lane == 0) l1[c] = k;
    }
    wave_sync();
    int left = n;
    for (;;) {
        const uint32_t best = wave_min_u32(lane < NC ? l1[lane] : NOKEY);
        if (best == NOKEY) break;
        const uint32_t p = best & 2047u, merged = best >> 11;
        const uint32_t sp = Sl[p], mp = Ml[p];
        const uint32_t q = (sp >> 20) & FL_NONE, p0 = (mp >> 20) & FL_NONE;  // the unit the merge consumes (there is one), the unit in front (or none)
        const uint32_t sq = Sl[q];
        const uint32_t sl0 = p0 != FL_NONE ? Sl[p0] : 0u;
        const uint32_t q2 = (sq >> 20) & FL_NONE;  // the unit behind the consumed one (or none)
        const uint32_t sr0 = q2 != FL_NONE ? Sl[q2] : 0u;
        const uint32_t mp0 = p0 != FL_NONE ? Ml[p0] : 0u, mq2 = q2 != FL_NONE ? Ml[q2] : 0u;
        // the two new pairs: lane 0 asks for (merged, right neighbour), lane 1 for (left neighbour, merged)
        uint32_t lk = SYM_NONE;
        const bool ask = (lane == 0 && q2 != FL_NONE) || (lane == 1 && p0 != FL_NONE);
        PairProbe pr{};
        const uint32_t pl = lane == 0 ? merged : (sl0 & FL_SYM), prr = lane == 0 ? (sr0 & FL_SYM) : merged;
        if (ask) pr = pair_issue(T, pl, prr);
        // the merge itself
        wave_sync();
        if (lane == 0) {
            Sl[p] = merged | (q2 << 20);
            Sl[q] = FL_DEAD;
            Ml[q] = PAIR_ABSENT | (FL_NONE << 20);
            if (q2 != FL_NONE) Ml[q2] = (mq2 & FL_SYM) | (p << 20);
        }
        left--;
        wave_sync();
        // the chunks the merge touched, without p and p0 (their pairs are being looked up) -- q is dead: its entry reads "no rank"
        const int cp = (int)(p >> 6), cq = (int)(q >> 6), c0 = p0 != FL_NONE ? (int)(p0 >> 6) : cp;
        uint32_t kp = chunk_key(cp, p, p0);
        uint32_t kq = cq != cp ? chunk_key(cq, p, p0) : NOKEY;
        uint32_t k0 = (c0 != cp && c0 != cq) ? chunk_key(c0, p, p0) : NOKEY;
        if (ask) lk = pair_resolve(T, pr, pl, prr);
        const uint32_t mr = (uint32_t)__builtin_amdgcn_readlane((int)lk, 0), ml = (uint32_t)__builtin_amdgcn_readlane((int)lk, 1);
        const uint32_t mrf = (q2 != FL_NONE && mr != SYM_NONE) ? mr : PAIR_ABSENT;
        const uint32_t mlf = (p0 != FL_NONE && ml != SYM_NONE) ? ml : PAIR_ABSENT;
        const uint32_t key_r = mrf != PAIR_ABSENT ? ((mrf << 11) | p) : NOKEY;
        const uint32_t key_l = mlf != PAIR_ABSENT ? ((mlf << 11) | p0) : NOKEY;
        kp = min(kp, key_r);
        if (c0 == cp) kp = min(kp, key_l);
        else if (c0 == cq) kq = min(kq, key_l);
        else k0 = min(k0, key_l);
        if (lane == 0) {
            Ml[p] = mrf | (p0 << 20);
            if (p0 != FL_NONE) Ml[p0] = mlf | (mp0 & ~FL_SYM);
            l1[cp] = kp;
            if (cq != cp) l1[cq] = kq;
            if (c0 != cp && c0 != cq) l1[c0] = k0;
        }
        wave_sync();
    }
    // the survivors to the front as plain symbols, 64 units at a time (writes never pass the reads)
    int out = 0;
    for (int base = 0; base < n; base += 64) {
        const int i = base + lane;
        const uint32_t e = i < n ? Sl[i] : FL_DEAD;
        const bool live = !(e & FL_DEAD);
        const unsigned long long bal = __ballot(live);
        wave_sync();
        if (live) Sl[out + __popcll(bal & ((1ull << lane) - 1ull))] = e & FL_SYM;
        out += __popcll(bal);
        wave_sync();
    }
    (void)left;
    return out;
}

// document holding byte ws of tile `tile`: last d with offsets[d] <= ws.  The tile metadata brackets it
// (tile_first_doc = first document at or after the tile start - LOOKBACK), so the search is two or three
// probes instead of log2(n_docs).
__device__ __forceinline__ int64_t doc_of(const BatchArgs& A, const Workspace& W, int64_t ws, uint32_t tile) {
    const int64_t f = W.tile_first_doc[tile];
    int64_t lo = f > 0 ? f - 1 : 0;                                                     // offsets[lo] <= ws
    int64_t hi = ((int64_t)tile + 2 < A.n_tiles) ? W.tile_first_doc[tile + 2] : A.n_docs;  // offsets[hi] > ws
    if (hi > A.n_docs) hi = A.n_docs;
    if (hi <= lo) hi = lo + 1;
    while (hi - lo > 1) {
        const int64_t mid = (lo + hi) >> 1;
        if (A.offsets[mid] <= ws) lo = mid; else hi = mid;
    }
    return lo;
}

// ------------------------------------------------------------------------
// k_exc_medium: exception words whose end the tile could see (at most 63 bytes, so at most 63 units) and
// that were refused only for having more than 32 units (or, non-byte mode, for the prefix budget): ONE LANE
// PER WORD, 64 words per wavefront, the merge loop of k_tiles with 64-bit unit masks.  Whatever it does not
// take (unknown end, more than 64 units with a prefix) is left for k_exc, one wavefront per word.
// ------------------------------------------------------------------------
constexpr int MEDIUM_UNITS = 64;
constexpr int QUAD_UNITS = 256;  // longest word of k_exc_b's quad list (d_exc_quad, d_exc_lane_fast<4>)
// A word of known length that d_exc_medium does not take (prefix units make it longer than MEDIUM_UNITS) goes straight on
// k_exc_quad's or k_exc's list, one atomic per wavefront and list; words of unknown length are d_exc_ends' business.
// (The quad list is TWO lists in one array: words of up to 128 units -- prefix included, whether or not the word gets it --
// from the front, counters[4] of them, the longer ones from the back, counters[11]: d_exc_group_fast<2> and <4> each walk
// their own.  As one list, 800 k words of 70-120 letters were walked a second time, 50 k lots of a cursor atomic and
// two dependent loads each, to find nothing.)
__device__ __forceinline__ uint32_t quad_list_len(const Workspace& W) { return W.counters[4] + W.counters[11]; }
__device__ __forceinline__ uint32_t quad_list_at(const Workspace& W, uint64_t li) {
    const uint32_t ns = W.counters[4];
    return li < ns ? W.exc_quad[li] : W.exc_quad[W.cap_exc - 1 - (int64_t)(li - ns)];
}
constexpr int QUAD_SHORT_UNITS = 128;
constexpr int GROUP_UNITS = 1024;  // longest word of d_exc_group_fast (16 lanes per word)
// the list of a word of `units` units (the prefix counted in, whether or not the word gets it): 0 / 1 the quad list's halves,
// 2 / 3 exc_mid's (d_exc_group_fast<8>, <16>: 16-bit symbols with rank == symbol order only), 4 d_exc's
__device__ __forceinline__ int exc_list_of(const DevTables& T, int64_t units) {
    const bool quad_ok = (T.is_byte_encoder || T.sym16) && T.rank_is_sym && !T.has_multi;
    const bool group_ok = HUTK_LAB_EXC_GROUP && T.sym16 && T.rank_is_sym && !T.has_multi;
    return quad_ok && units <= QUAD_SHORT_UNITS ? 0 : quad_ok && units <= QUAD_UNITS ? 1
         : group_ok && units <= GROUP_UNITS / 2 ? 2 : group_ok && units <= GROUP_UNITS ? 3 : 4;
}
__device__ __forceinline__ uint32_t* exc_list_slot(const Workspace& W, int list, uint32_t k) {  // entry k of list 0 .. 4
    return list == 0 ? W.exc_quad + k : list == 1 ? W.exc_quad + (W.cap_exc - 1 - (int64_t)k)
         : list == 2 ? W.exc_mid + k : list == 3 ? W.exc_mid + (W.cap_exc - 1 - (int64_t)k) : W.exc_wave + k;
}
__device__ __forceinline__ uint32_t* exc_list_count(const Workspace& W, int list) {
    return W.counters + (list == 0 ? 4 : list == 1 ? 11 : list == 2 ? 12 : list == 3 ? 13 : 5);
}
__device__ __forceinline__ void medium_leave(const DevTables& T, const Workspace& W, bool leave, uint64_t idx, int lane, int32_t len) {
    // (a length from k_tiles can be anything up to a tile's window; outside byte-encoder mode the quad list is d_exc_lane_fast's only)
    const int list = leave ? exc_list_of(T, (int64_t)len + T.n_prefix) : -1;
    const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll
    for (int l = 0; l < 5; l++) {  // one atomic per wavefront and list
        const unsigned long long b = __ballot(list == l);
        if (b) {
            uint32_t at = 0;
            if (lane == 0) at = atomicAdd(exc_list_count(W, l), (uint32_t)__popcll(b));
            at = __shfl(at, 0, 64);
            if (list == l) *exc_list_slot(W, l, at + __popcll(b & below)) = (uint32_t)idx;
        }
    }
}

// The same for 16-bit symbols with rank == symbol order (GPT-2-shaped files, the id-keyed path): ONE DWORD PER UNIT,
// merged symbol of (unit, next live unit) << 16 | symbol of the unit, in a row of the lane's own (MEDIUM_ROW dwords), so
// that the search for the best pair reads FOUR units per LDS instruction and prices each with one v_and_or:
// key = merged << 16 | position, smallest key = minimal rank, leftmost on ties (queue.c:162-164); a dead unit and a pair
// without a rank read 0xFFFF in the upper half.  (The general form below looks its candidates up one by one through
// 64-bit masks: ~10 instructions per candidate and trip, which on words of 33..62 letters was 4/5 of the kernel's time.)
//
// NW = 1: the words d_exc_medium takes (up to 64 units, straight from the exception records, 64 words per wavefront).
// NW = 2, 4: the words of k_exc_quad's list (up to QUAD_UNITS = 256 units, lengths found by d_exc_ends): those of up to 128
// units SIXTEEN per wavefront in rows of 132 dwords, the longer ones EIGHT per wavefront in rows of 260, liveness in two /
// four 64-bit words.  A quarter / an eighth of the lanes work -- in 8.4 KB of LDS, so that k_exc_b's other role keeps its
// resident wavefronts -- and still do more words per microsecond than d_exc_quad's sixteen lanes per word, whose every
// trip pays a DPP reduction and a chain of cross-lane shuffles to find the neighbours.
constexpr int MEDIUM_ROW = MEDIUM_UNITS + 4;  // dwords per lane: 16-byte aligned rows, lanes spread over the banks
template <int NW>
struct LiveBits {  // units still alive, 64 per word
    uint64_t w[NW];
    __device__ __forceinline__ void init(int n) {
#pragma unroll
        for (int k = 0; k < NW; k++) w[k] = n >= 64 * (k + 1) ? ~0ull : n > 64 * k ? ((1ull << (n - 64 * k)) - 1ull) : 0ull;
    }
    __device__ __forceinline__ void clear(int q) {
#pragma unroll
        for (int k = 0; k < NW; k++)
            if ((q >> 6) == k) w[k] &= ~(1ull << (q & 63));
    }
    __device__ __forceinline__ int next_after(int p) const {  // first live unit behind p, or -1
        int r = -1;
#pragma unroll
        for (int k = NW - 1; k >= 0; k--) {
            const int pk = p - 64 * k;
            uint64_t m = w[k];
            if (pk >= 63) m = 0;
            else if (pk >= 0) m &= ~((2ull << pk) - 1ull);
            if (m) r = 64 * k + __builtin_ctzll(m);
        }
        return r;
    }
    __device__ __forceinline__ int prev_before(int p) const {  // last live unit in front of p, or -1
        int r = -1;
#pragma unroll
        for (int k = 0; k < NW; k++) {
            const int pk = p - 64 * k;
            uint64_t m = w[k];
            if (pk <= 0) m = 0;
            else if (pk < 64) m &= (1ull << pk) - 1ull;
            if (m) r = 64 * k + 63 - __builtin_clzll(m);
        }
        return r;
    }
};
template <int NW, int LANES>
__device__ __forceinline__ void d_exc_lane_fast(const DevTables& T, const BatchArgs& A, const Workspace& W, uint32_t vblock,
                                                uint32_t vgrid, uint8_t* lds) {
    constexpr int UNITS = 64 * NW, ROW = UNITS + 4;
    static_assert(UNITS <= QUAD_UNITS, "the quad list holds words of up to QUAD_UNITS units");
    const int lane = threadIdx.x & 63;
    uint32_t* const U = reinterpret_cast<uint32_t*>(lds) + (lane % LANES) * ROW;
    constexpr uint32_t HI = 0xFFFF0000u;
    const uint32_t n_exc = NW == 1 ? W.counters[0] : quad_list_len(W);  // records, or entries of the quad list (both halves)
    // 64 words at a time: the first lot by block index, further ones from a device cursor (counters[3]): words differ in
    // their number of merges, and a fixed share per wavefront left the last ones running alone
    for (uint32_t round = 0;; round++) {
        uint32_t lot = vblock;
        if (round) {
            if (lane == 0) lot = vgrid + atomicAdd(&W.counters[NW == 1 ? 3 : NW == 2 ? 8 : 9], 1u);
            lot = (uint32_t)__shfl((int)lot, 0, 64);
        }
        const uint64_t base = (uint64_t)lot * LANES;
        if (base >= n_exc || (NW == 1 && (int64_t)base >= W.cap_exc)) break;
        const uint64_t at = base + lane;
        bool have = lane < LANES && at < n_exc && (NW > 1 || (int64_t)at < W.cap_exc);
        uint64_t idx = at;  // the word's exception record
        if (NW > 1 && have) idx = quad_list_at(W, at);
        ExcRec rec{};
        if (have) rec = W.exc[idx];
        if (NW == 1) {
            have = have && rec.len >= 1 && rec.len <= LANE_MAX_BYTES && rec.cnt == 0;
            have = have && !T.has_multi;  // (items of several units: every exception word goes to d_exc, which expands them)
        }
        int64_t gbase = 0;
        int n = 0, na = 0;
        LiveBits<NW> live;
        live.init(0);
        uint32_t best = 0xFFFFFFFFu;
        // best key of the lane's row: 16 bytes = four units per read, four reads in flight (the row reads "no rank" from
        // the word's last unit to the next multiple of 16)
        auto scan_row = [&](int nn) -> uint32_t {
            uint32_t b0 = 0xFFFFFFFFu, b1 = 0xFFFFFFFFu;
            for (int i = 0; i < nn; i += 16) {
                uint4 v[4];
#pragma unroll
                for (int j = 0; j < 4; j++) v[j] = *reinterpret_cast<const uint4*>(U + i + 4 * j);
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const uint32_t at = (uint32_t)(i + 4 * j);
                    b0 = min(b0, min((v[j].x & HI) | at, (v[j].y & HI) | (at + 1u)));
                    b1 = min(b1, min((v[j].z & HI) | (at + 2u), (v[j].w & HI) | (at + 3u)));
                }
            }
            return min(b0, b1);
        };
        if (have) {
            const int64_t ws = rec.ws;
            const int nb = rec.len;
            const int64_t d = T.has_prefix ? doc_of(A, W, ws, rec.tile) : 0;  // (needed for the prefix and its room in exc_tok only)
            const bool docfirst = T.has_prefix && word_is_first(A, ws, A.offsets[d]);
            const bool with_prefix = T.has_prefix && docfirst;
            const bool alone = with_prefix && doc_begins_with_space(A, ws);  // core.c:365-366, 421-446
            const int kp = (with_prefix && !alone) ? T.n_prefix : 0;
            na = alone ? T.n_prefix_alone : 0;
            gbase = ws * T.unit_scale + (int64_t)W.pad_per_doc * (docfirst ? d : d + 1);
            if (kp + nb > UNITS || (NW == 4 && kp + nb <= UNITS / 2)) {
                have = false;  // NW == 1: k_exc's or the quad list's; NW == 2: left to the NW == 4 pass; NW == 4: the NW == 2 pass took it
            } else {
                for (int i = 0; i < kp; i++) U[i] = HI | (T.prefix_syms[i] & 0xFFFFu);
                n = kp;
                int looked_up = 0;  // units [0, looked_up) still need their pair result from the pair table
                if (T.is_byte_encoder) {
                    // sixteen units per step: their bytes in flight together, then their (byte, next byte) table entries --
                    // merged symbol of the pair << 16 | symbol of the byte: the row's dword as it is
                    const uint8_t* wb = A.bytes + ws;
                    const uint32_t* bp = reinterpret_cast<const uint32_t*>(T.bytepair);
                    for (int i0 = 0; i0 < nb; i0 += 16) {
                        uint32_t b[17];
#pragma unroll
                        for (int j = 0; j < 17; j++) b[j] = wb[min(i0 + j, nb - 1)];  // (clamped: in bounds, no branch)
                        uint32_t e[16];
#pragma unroll
                        for (int j = 0; j < 16; j++) e[j] = bp[b[j] | (b[j + 1] << 8)];
#pragma unroll
                        for (int j = 0; j < 16; j++)
                            if (i0 + j < nb) U[n + i0 + j] = (i0 + j + 1 < nb) ? e[j] : (e[j] | HI);
                    }
                    looked_up = n;  // (prefix units in front: their pairs, and the one into the word)
                    n += nb;
                } else {
                    for (int i = 0; i < nb;) {
                        const uint32_t b = A.bytes[ws + i];
                        int L = (b < 0x80u) ? 1 : (b >= 0xF0u) ? 4 : (b >= 0xE0u) ? 3 : (b >= 0xC0u) ? 2 : 1;
                        uint32_t sym;
                        if ((b >= 0x80u && (L == 1 || b >= 0xF8u)) || i + L > nb) {
                            raise(A.err, HUTK_E_INVALID_UTF8);
                            sym = SYM_UNK;
                            L = 1;
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
                        U[n] = HI | (sym & 0xFFFFu);
                        n++;
                        i += L;
                    }
                    looked_up = n - 1;
                }
                for (int i0 = 0; i0 < looked_up; i0 += 4) {  // four lookups (eight loads) in flight
                    PairProbe pr[4];
                    uint32_t sy[5];
#pragma unroll
                    for (int j = 0; j < 5; j++) sy[j] = (i0 + j < n) ? (U[i0 + j] & 0xFFFFu) : 0u;
#pragma unroll
                    for (int j = 0; j < 5; j++) sy[j] = sy[j] == 0xFFFFu ? SYM_UNK : sy[j];
#pragma unroll
                    for (int j = 0; j < 4; j++) pr[j] = pair_issue(T, sy[j], sy[j + 1]);
#pragma unroll
                    for (int j = 0; j < 4; j++)
                        if (i0 + j < looked_up && i0 + j + 1 < n) {
                            const uint32_t m = pair_resolve(T, pr[j], sy[j], sy[j + 1]);
                            U[i0 + j] = (m << 16) | (sy[j] & 0xFFFFu);  // (SYM_NONE: 0xFFFF in the upper half)
                        }
                }
                for (int i = n; i < ((n + 15) & ~15); i++) U[i] = 0xFFFFFFFFu;  // the row's last reads cover them
                live.init(n);
                best = scan_row(n);
            }
        }
        // One merge per trip and lane: apply the best pair, issue the lookups of the two new neighbour pairs, search the
        // row again while those loads fly (the pairs that change read "no rank" meanwhile), fold the two new keys in.
        for (;;) {
            const bool act = have && best < HI;
            if (!__any(act)) break;
            if (act) {
                const int p = (int)(best & 0xFFFFu);
                const uint32_t merged = best >> 16;
                const int q = live.next_after(p);  // the unit the merge consumes (there is one: the pair was a candidate)
                live.clear(q);
                const int qn = live.next_after(p), pn = live.prev_before(p);
                const bool right = qn >= 0, left = pn >= 0;
                const int q2 = right ? qn : p;
                const int p0 = left ? pn : p;
                const uint32_t ur = U[q2], ul = U[p0];
                uint32_t sr = ur & 0xFFFFu, sl = ul & 0xFFFFu;
                sr = sr == 0xFFFFu ? SYM_UNK : sr;  // (a unit that is no symbol: never a member of a pair)
                sl = sl == 0xFFFFu ? SYM_UNK : sl;
                const PairProbe pr = pair_issue(T, merged, sr), pl = pair_issue(T, sl, merged);  // both in flight
                U[q] = 0xFFFFFFFFu;
                U[p0] = ul | HI;      // (first: without a left neighbour p0 == p)
                U[p] = HI | merged;
                best = scan_row(n);
                if (right) {
                    const uint32_t m = pair_resolve(T, pr, merged, sr);
                    U[p] = (m << 16) | merged;
                    best = min(best, (m << 16) | (uint32_t)p);
                }
                if (left) {
                    const uint32_t m = pair_resolve(T, pl, sl, merged);
                    U[p0] = (m << 16) | (ul & 0xFFFFu);
                    best = min(best, (m << 16) | (uint32_t)p0);
                }
            }
        }
        if (have) {
            int32_t* out = W.exc_tok + gbase;
            for (int i = 0; i < na; i++) out[i] = T.prefix_alone_ids[i];
            int k = na;
#pragma unroll
            for (int j = 0; j < NW; j++)
                for (uint64_t c = live.w[j]; c;) {  // eight ids per step, as in d_exc_group_fast
                    int pos[8];
                    bool ok[8];
#pragma unroll
                    for (int i = 0; i < 8; i++) {
                        ok[i] = c != 0;
                        pos[i] = ok[i] ? __builtin_ctzll(c) : 0;
                        c &= c - 1;
                    }
                    uint32_t sy[8];
#pragma unroll
                    for (int i = 0; i < 8; i++) sy[i] = U[64 * j + pos[i]] & 0xFFFFu;
                    int32_t id[8];
#pragma unroll
                    for (int i = 0; i < 8; i++) id[i] = sym_to_id(T, sy[i] == 0xFFFFu ? SYM_UNK : sy[i]);
#pragma unroll
                    for (int i = 0; i < 8; i++)
                        if (ok[i]) { out[k] = id[i]; k++; }
                }
            rec.cnt = (uint32_t)k;
            rec.tok_base = gbase;
            W.exc[idx] = rec;
            atomicAdd(&W.tile_count[rec.tile], rec.cnt);
        }
        if (NW == 1) medium_leave(T, W, !have && idx < n_exc && (int64_t)idx < W.cap_exc && rec.len >= 1 && rec.cnt == 0, idx, lane, rec.len);
    }
}

// d_exc_lane_fast's words of 65..256 units with NW LANES PER WORD (round 4): the same rows, 32 (NW = 2: up to 128 units)
// or 16 (NW = 4) words per wavefront -- every lane at work, where the one-lane form kept 16 / 8 of 64 busy to stay within
// 8.4 KB.  What a lane did alone is shared out:
//   * the row's search, 3/4 of a trip's instructions: the row is 16-unit blocks dealt round the group (lane s: blocks s,
//     s + NW, ...: four per lane), a lane keeps the best key of each of its blocks in a register, and a trip searches again
//     only the blocks whose dwords it changed -- at most one per lane, read at a lane-dependent address, so the wavefront
//     runs the 16-unit search ONCE per trip whatever the words' lengths (a word whose live units lie so far apart that
//     a lane owns two changed blocks searches all its blocks: seldom); a DPP minimum over the group ends the trip;
//   * the liveness bits: lane s holds units 64 s .. 64 s + 63, a neighbour is a DPP minimum / maximum of the lanes' answers;
//   * the two pair lookups of a merge: lane 0 the new right pair, lane 1 the new left one.
// The search now follows the lookups (it reads their results) instead of running under them: with two or more
// wavefronts per SIMD the kernel is bound by the instructions it issues, not by a trip's latency (k_exc_b with half its
// wavefronts took the same time, profiles/r04_exc_group_ab.txt).
template <int LPW>
__device__ __forceinline__ uint32_t group_min_u32(uint32_t v) {  // minimum over each group of LPW (2 .. 16) lanes, in every lane
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)v, 0xB1, 0xf, 0xf, false));   // quad_perm [1,0,3,2]
    if (LPW >= 4) v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)v, 0x4E, 0xf, 0xf, false));   // quad_perm [2,3,0,1]
    if (LPW >= 8) v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)v, 0x141, 0xf, 0xf, false));  // row_half_mirror
    if (LPW >= 16) v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)v, 0x140, 0xf, 0xf, false)); // row_mirror
    return v;
}
template <int LPW>
__device__ __forceinline__ uint32_t group_max_u32(uint32_t v) {
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, false));
    if (LPW >= 4) v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, false));
    if (LPW >= 8) v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xf, 0xf, false));
    if (LPW >= 16) v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xf, 0xf, false));
    return v;
}
template <int NW>
__device__ __forceinline__ void d_exc_group_fast(const DevTables& T, const BatchArgs& A, const Workspace& W, uint32_t vblock,
                                                 uint32_t vgrid, uint8_t* lds) {
    constexpr int LPW = NW, UNITS = 64 * NW, ROW = UNITS + 4, WPW = 64 / LPW;
    static_assert(NW == 2 || NW == 4 || NW == 8 || NW == 16, "groups within a DPP row");
    static_assert(UNITS <= GROUP_UNITS, "the lists hold words of up to GROUP_UNITS units");
    const int lane = threadIdx.x & 63, sub = lane % LPW, w = lane / LPW;
    uint32_t* const U = reinterpret_cast<uint32_t*>(lds) + w * ROW;
    constexpr uint32_t HI = 0xFFFF0000u;
    const uint32_t n_exc = W.counters[NW == 2 ? 4 : NW == 4 ? 11 : NW == 8 ? 12 : 13];  // entries of my list (medium_leave)
#if HUTK_LAB_EXC_STAMPS
    long long st_acc[4] = {0, 0, 0, 0}, st_trips = 0, st_lots = 0, st_set[4] = {0, 0, 0, 0};
    const long long st_begin = clock64();
#endif
    for (uint32_t round = 0;; round++) {
        uint32_t lot = vblock;
        if (round) {
            if (lane == 0) lot = vgrid + atomicAdd(&W.counters[NW == 2 ? 8 : NW == 4 ? 9 : NW == 8 ? 14 : 15], 1u);
            lot = (uint32_t)__shfl((int)lot, 0, 64);
        }
#if HUTK_LAB_EXC_STAMPS
        const long long sl0 = clock64();
#endif
        const uint64_t base = (uint64_t)lot * WPW;
        if (base >= n_exc) break;
        const uint64_t at = base + w;
        bool have = at < n_exc;  // (the same for a group's lanes, as everything below that does not mention sub)
        uint64_t idx = 0;
        if (have) idx = NW == 2 ? W.exc_quad[at] : NW == 4 ? W.exc_quad[W.cap_exc - 1 - (int64_t)at] : NW == 8 ? W.exc_mid[at] : W.exc_mid[W.cap_exc - 1 - (int64_t)at];
        ExcRec rec{};
        if (have) rec = W.exc[idx];
        const uint32_t rec_tile = rec.tile;  // (the record itself does not stay in registers over the trips)
#if HUTK_LAB_EXC_STAMPS
        asm volatile("" :: "v"(rec_tile));
        const long long sl1 = clock64();
#endif
        int64_t gbase = 0;
        int n = 0, na = 0;
        uint64_t lv = 0;       // the lane's share of the liveness bits: units 64 sub .. 64 sub + 63
        uint32_t bm[4] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};  // best keys of blocks sub, sub + LPW, ...
        uint32_t best = 0xFFFFFFFFu;
        // best key of the 16 units of block blk: four 16-byte reads in flight (the row reads "no rank" from the word's last
        // unit to the next multiple of 16; a block behind that is not the word's: no key)
        auto scan_block = [&](int blk) -> uint32_t {
            uint4 v[4];
#pragma unroll
            for (int j = 0; j < 4; j++) v[j] = *reinterpret_cast<const uint4*>(U + 16 * blk + 4 * j);
            uint32_t b0 = 0xFFFFFFFFu, b1 = 0xFFFFFFFFu;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const uint32_t at = (uint32_t)(16 * blk + 4 * j);
                b0 = min(b0, min((v[j].x & HI) | at, (v[j].y & HI) | (at + 1u)));
                b1 = min(b1, min((v[j].z & HI) | (at + 2u), (v[j].w & HI) | (at + 3u)));
            }
            return 16 * blk < n ? min(b0, b1) : 0xFFFFFFFFu;
        };
        if (have) {
            const int64_t ws = rec.ws;
            const int nb = rec.len;
            // (the word's document matters for the prefix and its room in exc_tok only: without a prefix, five dependent loads less)
            const int64_t d = T.has_prefix ? doc_of(A, W, ws, rec.tile) : 0;
            const bool docfirst = T.has_prefix && word_is_first(A, ws, A.offsets[d]);
            const bool with_prefix = T.has_prefix && docfirst;
            const bool alone = with_prefix && doc_begins_with_space(A, ws);  // core.c:365-366, 421-446
            const int kp = (with_prefix && !alone) ? T.n_prefix : 0;
            na = alone ? T.n_prefix_alone : 0;
            gbase = ws * T.unit_scale + (int64_t)W.pad_per_doc * (docfirst ? d : d + 1);
            if (kp + nb > UNITS) {
                have = false;  // (cannot be: the lists were made with the prefix counted in)
            } else {
                if (sub == 0)
                    for (int i = 0; i < kp; i++) U[i] = HI | (T.prefix_syms[i] & 0xFFFFu);
                n = kp;
                int looked_up = 0;  // units [0, looked_up) still need their pair result from the pair table
                if (T.is_byte_encoder) {
                    // eight units per step and lane, the steps dealt round the group, two steps at a time: their bytes as three
                    // unaligned dwords each (the ninth byte is the next unit's: its pair), then their sixteen byte-pair entries
                    // in flight together -- two round trips per sixteen of the lane's units
                    const uint8_t* wb = A.bytes + ws;
                    const uint32_t* bp = reinterpret_cast<const uint32_t*>(T.bytepair);
                    const bool wide_ok = ws + ((nb + 7) & ~7) + 4 <= A.n_bytes;  // (a dword may reach 11 bytes past a step's first)
                    for (int i00 = 8 * sub; i00 < nb; i00 += 16 * LPW) {
                        uint32_t wd[2][3];
#pragma unroll
                        for (int h = 0; h < 2; h++) {
                            const int i0 = i00 + 8 * LPW * h;
                            if (wide_ok) {
#pragma unroll
                                for (int j = 0; j < 3; j++) {
                                    uint32_t x;
                                    __builtin_memcpy(&x, wb + min(i0, (nb - 1) & ~7) + 4 * j, 4);
                                    wd[h][j] = x;
                                }
                            } else {
#pragma unroll
                                for (int j = 0; j < 3; j++) {
                                    uint32_t x = 0;
                                    for (int q = 0; q < 4; q++) x |= (uint32_t)wb[min(i0 + 4 * j + q, nb - 1)] << (8 * q);
                                    wd[h][j] = x;
                                }
                            }
                        }
                        uint32_t e[2][8];
#pragma unroll
                        for (int h = 0; h < 2; h++)
#pragma unroll
                            for (int j = 0; j < 8; j++) {
                                const uint32_t b0 = (wd[h][j >> 2] >> (8 * (j & 3))) & 0xFFu;
                                const uint32_t b1 = (wd[h][(j + 1) >> 2] >> (8 * ((j + 1) & 3))) & 0xFFu;
                                e[h][j] = bp[b0 | (b1 << 8)];
                            }
#pragma unroll
                        for (int h = 0; h < 2; h++) {
                            const int i0 = i00 + 8 * LPW * h;
#pragma unroll
                            for (int j = 0; j < 8; j++)
                                if (i0 + j < nb) U[n + i0 + j] = (i0 + j + 1 < nb) ? e[h][j] : (e[h][j] | HI);
                        }
                    }
                    looked_up = n;  // (prefix units in front: their pairs, and the one into the word)
                    n += nb;
                } else {
                    if (sub == 0) {  // (characters of one to four bytes: one lane walks them)
                        for (int i = 0; i < nb;) {
                            const uint32_t b = A.bytes[ws + i];
                            int L = (b < 0x80u) ? 1 : (b >= 0xF0u) ? 4 : (b >= 0xE0u) ? 3 : (b >= 0xC0u) ? 2 : 1;
                            uint32_t sym;
                            if ((b >= 0x80u && (L == 1 || b >= 0xF8u)) || i + L > nb) {
                                raise(A.err, HUTK_E_INVALID_UTF8);
                                sym = SYM_UNK;
                                L = 1;
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
                            U[n] = HI | (sym & 0xFFFFu);
                            n++;
                            i += L;
                        }
                    }
                    n = __shfl(n, lane - sub, 64);
                    looked_up = n - 1;
                }
                for (int i0 = 4 * sub; i0 < looked_up; i0 += 4 * LPW) {  // four lookups (eight loads) in flight per lane
                    PairProbe pr[4];
                    uint32_t sy[5];
#pragma unroll
                    for (int j = 0; j < 5; j++) sy[j] = (i0 + j < n) ? (U[i0 + j] & 0xFFFFu) : 0u;
#pragma unroll
                    for (int j = 0; j < 5; j++) sy[j] = sy[j] == 0xFFFFu ? SYM_UNK : sy[j];
#pragma unroll
                    for (int j = 0; j < 4; j++) pr[j] = pair_issue(T, sy[j], sy[j + 1]);
                    // (a unit's dword is written by the lane that holds its step only: the next step's lane has read the
                    // symbol it needs -- the low half, which stays -- whenever it comes by)
#pragma unroll
                    for (int j = 0; j < 4; j++)
                        if (i0 + j < looked_up && i0 + j + 1 < n) {
                            const uint32_t m = pair_resolve(T, pr[j], sy[j], sy[j + 1]);
                            U[i0 + j] = (m << 16) | (sy[j] & 0xFFFFu);  // (SYM_NONE: 0xFFFF in the upper half)
                        }
                }
                if (sub == 0)
                    for (int i = n; i < ((n + 15) & ~15); i++) U[i] = 0xFFFFFFFFu;  // the row's last reads cover them
                const int mine_n = n - 64 * sub;
                lv = mine_n >= 64 ? ~0ull : mine_n > 0 ? ((1ull << mine_n) - 1ull) : 0ull;
            }
        }
        // (one block at a time: unrolled, the compiler keeps all sixteen reads' registers at once)
        auto scan_all = [&]() {
#pragma unroll 1
            for (int k = 0; k < 4; k++) {
                const uint32_t r = scan_block(sub + LPW * k);
#pragma unroll
                for (int j = 0; j < 4; j++) bm[j] = k == j ? r : bm[j];
            }
        };
#if HUTK_LAB_EXC_STAMPS
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        const long long sl2 = clock64();
#endif
        if (have) {
            scan_all();
            best = group_min_u32<LPW>(min(min(bm[0], bm[1]), min(bm[2], bm[3])));
        }
#if HUTK_LAB_EXC_STAMPS
        st_lots++;
        asm volatile("" :: "v"(best));
        const long long sl3 = clock64();
        st_set[0] += sl1 - sl0; st_set[1] += sl2 - sl1; st_set[2] += sl3 - sl2;
#endif
        // One merge per trip and word
        for (;;) {
            const bool act = have && best < HI;
            if (!__any(act)) break;
#if HUTK_LAB_EXC_STAMPS
            const long long st0 = clock64();
            int st_d1 = 0, st_d2 = 0;
#endif
            if (act) {
                const int p = (int)(best & 0xFFFFu);
                const uint32_t merged = best >> 16;
                auto first_after = [&](int x) -> int {  // first live unit behind x, or 0xFFFF
                    const int sx = x >> 6;
                    const uint64_t m = sub == sx ? (lv & ~((2ull << (x & 63)) - 1ull)) : sub > sx ? lv : 0ull;
                    return (int)group_min_u32<LPW>(m ? (uint32_t)(64 * sub + __builtin_ctzll(m)) : 0xFFFFu);
                };
                const int q = first_after(p);  // the unit the merge consumes (there is one: the pair was a candidate)
                if (sub == (q >> 6)) lv &= ~(1ull << (q & 63));
                const int qn = first_after(q);
                const int sp = p >> 6;
                const uint64_t mb = sub == sp ? (lv & ((1ull << (p & 63)) - 1ull)) : sub < sp ? lv : 0ull;
                const int pn = (int)group_max_u32<LPW>(mb ? (uint32_t)(64 * sub + 64 - __builtin_clzll(mb)) : 0u) - 1;  // last live unit in front of p, or -1
                const bool right = qn != 0xFFFF, left = pn >= 0;
                const bool mine = sub == 0 ? right : sub == 1 ? left : false;  // lane 0: the pair (p, qn), lane 1: (pn, p)
                uint32_t a = 0, b = 0, low = 0;
                PairProbe pr{};
                if (mine) {
                    const uint32_t un = U[sub == 0 ? qn : pn];
                    uint32_t sn = un & 0xFFFFu;
                    sn = sn == 0xFFFFu ? SYM_UNK : sn;  // (a unit that is no symbol: never a member of a pair)
                    a = sub == 0 ? merged : sn;
                    b = sub == 0 ? sn : merged;
                    low = un & 0xFFFFu;
                    pr = pair_issue(T, a, b);
                }
                if (sub == 0) U[q] = 0xFFFFFFFFu;
                // which of its blocks has the lane to search again?  The changed dwords are pn's, p's and q's: blocks bl <= bp <= bq
                const int bq = q >> 4, bl = (left ? pn : p) >> 4, bpp = p >> 4;
                const bool wide = bq - bl >= LPW;  // (a lane may own two of them)
                const int tb = (bq % LPW) == sub ? bq : (bpp % LPW) == sub ? bpp : (bl % LPW) == sub ? bl : sub;
#if HUTK_LAB_EXC_STAMPS
                st_d1 = (int)(clock64() - st0);
#endif
                if (sub == 0 || mine) {  // (lane 0 stores the merged symbol whether or not it has a right neighbour)
                    const uint32_t r = pair_resolve(T, pr, a, b);
                    const uint32_t m = mine ? r : SYM_NONE;
                    U[sub == 0 ? p : pn] = (m << 16) | (sub == 0 ? merged : low);
                }
#if HUTK_LAB_EXC_STAMPS
                st_d2 = (int)(clock64() - st0);
#endif
                if (wide) {
                    scan_all();
                } else {
                    const uint32_t r = scan_block(tb);
                    const int k = tb / LPW;
#pragma unroll
                    for (int j = 0; j < 4; j++) bm[j] = k == j ? r : bm[j];
                }
                best = group_min_u32<LPW>(min(min(bm[0], bm[1]), min(bm[2], bm[3])));
            }
#if HUTK_LAB_EXC_STAMPS
            {
                const int src = __builtin_ctzll(__ballot(act));
                const long long d1 = __shfl(st_d1, src, 64), d2 = __shfl(st_d2, src, 64), d3 = clock64() - st0;
                st_acc[0] += d1; st_acc[1] += d2 - d1; st_acc[2] += d3 - d2; st_trips++;
            }
#endif
        }
#if HUTK_LAB_EXC_STAMPS
        const long long sl4 = clock64();
#endif
        if (have) {
            int32_t* out = W.exc_tok + gbase;
            if (sub == 0)
                for (int i = 0; i < na; i++) out[i] = T.prefix_alone_ids[i];
            // lane s: the units of its 64 liveness bits, behind those of the lanes before it
            const int cnt = __popcll(lv);
            int incl = cnt;  // inclusive scan over the group's lanes
#pragma unroll
            for (int dlt = 1; dlt < LPW; dlt *= 2) {
                const int o = __shfl_up(incl, dlt, LPW);
                if (sub >= dlt) incl += o;
            }
            const int total = na + __shfl(incl, LPW - 1, LPW);
            int k = na + incl - cnt;
            // eight ids per step: their symbols' LDS reads together, their sym_id loads together (one by one, each id was a
            // dependent LDS read and global load: 46 k of a lot's 85 k cycles outside its trips, profiles/r04_exc_group_ab.txt)
            for (uint64_t c = lv; c;) {
                int pos[8];
                bool ok[8];
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    ok[j] = c != 0;
                    pos[j] = ok[j] ? __builtin_ctzll(c) : 0;
                    c &= c - 1;  // (0 stays 0)
                }
                uint32_t sy[8];
#pragma unroll
                for (int j = 0; j < 8; j++) sy[j] = U[64 * sub + pos[j]] & 0xFFFFu;
                int32_t id[8];
#pragma unroll
                for (int j = 0; j < 8; j++) id[j] = sym_to_id(T, sy[j] == 0xFFFFu ? SYM_UNK : sy[j]);
#pragma unroll
                for (int j = 0; j < 8; j++)
                    if (ok[j]) out[k + j] = id[j];
                k += 8;  // (the last step's surplus is not stored)
            }
            if (sub == 0) {
                W.exc[idx].cnt = (uint32_t)total;
                W.exc[idx].tok_base = gbase;
                atomicAdd(&W.tile_count[rec_tile], (uint32_t)total);
            }
        }
#if HUTK_LAB_EXC_STAMPS
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        st_set[3] += clock64() - sl4;
#endif
    }
#if HUTK_LAB_EXC_STAMPS
    if (W.prof && lane == 0 && vblock < (uint32_t)A.n_tiles) {
        long long* o = W.prof + (size_t)vblock * 10;
        if (NW == 2) { o[0] = clock64() - st_begin; o[1] = st_acc[0]; o[2] = st_acc[1]; o[3] = st_acc[2]; o[4] = st_trips; o[5] = st_lots; o[6] = st_set[0]; o[7] = st_set[1]; o[8] = st_set[2]; o[9] = st_set[3]; }
    }
#endif
}

template <typename SymT>
__device__ __forceinline__ void d_exc_medium(const DevTables& T, const BatchArgs& A, const Workspace& W, uint32_t vblock,
                                             uint32_t vgrid, uint8_t* lds) {
    // 16-bit symbols when the vocabulary allows: half the LDS, twice the resident wavefronts (the loop is
    // bound by the latency of its pair lookups)
    // unit i of the lane's word at Sm[i * 64 + lane]; Mm: merged symbol of (unit i, next live unit) or NONE
    SymT* const Sm = reinterpret_cast<SymT*>(lds);
    SymT* const Mm = Sm + MEDIUM_UNITS * 64;
    const int lane = threadIdx.x;
    const uint32_t n_exc = W.counters[0];
    for (uint64_t base = (uint64_t)vblock * 64; base < n_exc && (int64_t)base < W.cap_exc;
         base += (uint64_t)vgrid * 64) {
        const uint64_t idx = base + lane;
        bool have = idx < n_exc && (int64_t)idx < W.cap_exc;
        ExcRec rec{};
        if (have) rec = W.exc[idx];
        have = have && rec.len >= 1 && rec.len <= LANE_MAX_BYTES && rec.cnt == 0;
        have = have && !T.has_multi;  // (items of several units: every exception word goes to d_exc, which expands them)
        int64_t d = 0, gbase = 0;
        int n = 0, na = 0, pairs_to = 0;
        uint64_t live = 0, cand = 0;
        if (have) {
            const int64_t ws = rec.ws;
            const int nb = rec.len;
            d = doc_of(A, W, ws, rec.tile);
            const bool docfirst = word_is_first(A, ws, A.offsets[d]);
            const bool with_prefix = T.has_prefix && docfirst;
            const bool alone = with_prefix && doc_begins_with_space(A, ws);  // core.c:365-366, 421-446
            const int kp = (with_prefix && !alone) ? T.n_prefix : 0;
            na = alone ? T.n_prefix_alone : 0;
            gbase = ws * T.unit_scale + (int64_t)W.pad_per_doc * (docfirst ? d : d + 1);
            if (kp + nb > MEDIUM_UNITS) {
                have = false;  // k_exc
            } else {
                for (int i = 0; i < kp; i++) Sm[i * 64 + lane] = Sym<SymT>::narrow(T.prefix_syms[i]);
                n = kp;
                if (T.is_byte_encoder) {
                    // sixteen units per step: their bytes in flight together, then their (byte, next byte) table entries --
                    // {symbol of the byte, merged symbol of the pair}, as in k_tiles -- together: two round trips per step
                    // instead of two per unit (a word of 47 letters: 6 instead of ~100)
                    const uint8_t* wb = A.bytes + ws;
                    const typename Sym<SymT>::Pair* bp = reinterpret_cast<const typename Sym<SymT>::Pair*>(T.bytepair);
                    for (int i0 = 0; i0 < nb; i0 += 16) {
                        uint32_t b[17];
#pragma unroll
                        for (int j = 0; j < 17; j++) b[j] = wb[min(i0 + j, nb - 1)];  // (clamped: in bounds, no branch)
                        typename Sym<SymT>::Pair e[16];
#pragma unroll
                        for (int j = 0; j < 16; j++) e[j] = bp[b[j] | (b[j + 1] << 8)];
#pragma unroll
                        for (int j = 0; j < 16; j++) {
                            const int i = i0 + j;
                            if (i < nb) {
                                Sm[(n + i) * 64 + lane] = Sym<SymT>::pair_sym(e[j]);
                                if (i + 1 < nb) {
                                    const SymT mv = Sym<SymT>::pair_merged(e[j]);
                                    Mm[(n + i) * 64 + lane] = mv;
                                    if (mv != Sym<SymT>::NONE) cand |= 1ull << (n + i);
                                }
                            }
                        }
                    }
                    pairs_to = n;  // (prefix units in front: their pairs, and the one into the word, are looked up below)
                    n += nb;
                } else {
                    for (int i = 0; i < nb;) {
                        const uint32_t b = A.bytes[ws + i];
                        int L = (b < 0x80u) ? 1 : (b >= 0xF0u) ? 4 : (b >= 0xE0u) ? 3 : (b >= 0xC0u) ? 2 : 1;
                        uint32_t sym;
                        if ((b >= 0x80u && (L == 1 || b >= 0xF8u)) || i + L > nb) {
                            raise(A.err, HUTK_E_INVALID_UTF8);
                            sym = SYM_UNK;
                            L = 1;
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
                        Sm[n * 64 + lane] = Sym<SymT>::narrow(sym);
                        n++;
                        i += L;
                    }
                }
                live = n >= 64 ? ~0ull : ((1ull << n) - 1ull);
                // pair results by table lookup: all of them outside byte-encoder mode; in it only those with a prefix unit
                // (units [0, pairs_to] as left members), the rest came with the (byte, next byte) entries
                for (int i0 = 0; i0 + 1 < (T.is_byte_encoder ? pairs_to + 1 : n); i0 += 4) {  // four lookups (eight loads) in flight
                    PairProbe pr[4];
                    uint32_t sy[5];
#pragma unroll
                    for (int j = 0; j < 5; j++) sy[j] = (i0 + j < n) ? Sym<SymT>::widen(Sm[(i0 + j) * 64 + lane]) : 0u;
#pragma unroll
                    for (int j = 0; j < 4; j++) pr[j] = pair_issue(T, sy[j], sy[j + 1]);
#pragma unroll
                    for (int j = 0; j < 4; j++)
                        if (i0 + j + 1 < n && (!T.is_byte_encoder || i0 + j < pairs_to)) {
                            const uint32_t m = pair_resolve(T, pr[j], sy[j], sy[j + 1]);
                            Mm[(i0 + j) * 64 + lane] = Sym<SymT>::narrow(m);
                            if (m != SYM_NONE) cand |= 1ull << (i0 + j);
                        }
                }
            }
        }
        // One merge per trip and lane: the candidate of minimal rank, leftmost on ties (queue.c:162-164).  As in
        // k_tiles the lane keeps its best pair (br, bp, bm) across trips: a trip applies it, issues the lookups of
        // the two new neighbour pairs, rescans the untouched candidates while those loads fly (four LDS reads per
        // step), and picks the next best among {rescan, new right pair, new left pair}.
        const bool ris = T.rank_is_sym != 0;
        auto RKm = [&](uint32_t m) -> uint32_t { return ris ? m : ((uint32_t)T.sym_id[m] ^ 0x80000000u); };
        auto scan4 = [&](uint64_t c, uint32_t& br, int& bp, uint32_t& bm) {
            while (c) {
                const uint64_t c1 = c & (c - 1), c2 = c1 & (c1 - 1), c3 = c2 & (c2 - 1);
                const int i0 = __builtin_ctzll(c);
                const int i1 = c1 ? __builtin_ctzll(c1) : i0, i2 = c2 ? __builtin_ctzll(c2) : i0,
                          i3 = c3 ? __builtin_ctzll(c3) : i0;
                const uint32_t m0 = Sym<SymT>::widen(Mm[i0 * 64 + lane]), m1 = Sym<SymT>::widen(Mm[i1 * 64 + lane]),
                               m2 = Sym<SymT>::widen(Mm[i2 * 64 + lane]), m3 = Sym<SymT>::widen(Mm[i3 * 64 + lane]);
                const uint32_t r0 = RKm(m0), r1 = RKm(m1), r2 = RKm(m2), r3 = RKm(m3);
                if (r0 < br) { br = r0; bp = i0; bm = m0; }
                if (r1 < br) { br = r1; bp = i1; bm = m1; }
                if (r2 < br) { br = r2; bp = i2; bm = m2; }
                if (r3 < br) { br = r3; bp = i3; bm = m3; }
                c = c3 & (c3 - 1);
            }
        };
        uint32_t br = 0xFFFFFFFFu, bm = 0;
        int bp = 0;
        if (have) scan4(cand, br, bp, bm);
        for (;;) {
            const bool act = have && cand != 0;
            if (!__any(act)) break;
            if (act) {
                const int p = bp;
                const uint32_t merged = bm;
                const uint64_t above = live & ~((2ull << p) - 1ull);
                const int q = __builtin_ctzll(above);  // the unit the merge consumes
                Sm[p * 64 + lane] = Sym<SymT>::narrow(merged);
                live &= ~(1ull << q);
                cand &= ~((1ull << q) | (1ull << p));
                const uint64_t right = above & (above - 1ull);
                const uint64_t left = live & ((1ull << p) - 1ull);
                const int p0 = left ? 63 - __builtin_clzll(left) : 0;
                const uint32_t sr = right ? Sym<SymT>::widen(Sm[__builtin_ctzll(right) * 64 + lane]) : 0u;
                const uint32_t sl = left ? Sym<SymT>::widen(Sm[p0 * 64 + lane]) : 0u;
                const PairProbe pr = pair_issue(T, merged, sr), pl = pair_issue(T, sl, merged);  // both in flight
                if (left) cand &= ~(1ull << p0);
                br = 0xFFFFFFFFu;
                scan4(cand, br, bp, bm);
                if (right) {
                    const uint32_t m = pair_resolve(T, pr, merged, sr);
                    Mm[p * 64 + lane] = Sym<SymT>::narrow(m);
                    if (m != SYM_NONE) {
                        cand |= 1ull << p;
                        const uint32_t r = RKm(m);
                        if (r < br || (r == br && p < bp)) { br = r; bp = p; bm = m; }
                    }
                }
                if (left) {
                    const uint32_t m = pair_resolve(T, pl, sl, merged);
                    Mm[p0 * 64 + lane] = Sym<SymT>::narrow(m);
                    if (m != SYM_NONE) {
                        cand |= 1ull << p0;
                        const uint32_t r = RKm(m);
                        if (r < br || (r == br && p0 < bp)) { br = r; bp = p0; bm = m; }
                    }
                }
            }
        }
        if (have) {
            int32_t* out = W.exc_tok + gbase;
            for (int i = 0; i < na; i++) out[i] = T.prefix_alone_ids[i];
            int k = na;
            for (uint64_t c = live; c; c &= c - 1) out[k++] = sym_to_id(T, Sym<SymT>::widen(Sm[__builtin_ctzll(c) * 64 + lane]));
            rec.cnt = (uint32_t)k;
            rec.tok_base = gbase;
            W.exc[idx] = rec;
            atomicAdd(&W.tile_count[rec.tile], rec.cnt);
        }
        medium_leave(T, W, !have && idx < n_exc && (int64_t)idx < W.cap_exc && rec.len >= 1 && rec.cnt == 0, idx, lane, rec.len);
    }
}

// ------------------------------------------------------------------------
// k_exc_quad: words of 64..256 units (byte-encoder mode, rank == symbol order): SIXTEEN LANES PER WORD, four
// words per wavefront.  Lane l of a group owns units 16l..16l+15: it keeps the best (rank, position) key of
// its own pairs in a register, the group minimum is a DPP row reduction, a consumed unit is marked dead
// (no compaction), and only the lanes whose pairs changed rescan.  A merge costs one round trip of pair
// lookups for four words at once instead of ~2 us for one word in k_exc.
// ------------------------------------------------------------------------
__device__ __forceinline__ uint32_t row_min_u32(uint32_t v) {  // minimum over each row of 16 lanes, in every lane
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)v, 0x111, 0xf, 0xf, false));
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)v, 0x112, 0xf, 0xf, false));
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)v, 0x114, 0xf, 0xf, false));
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)v, 0x118, 0xf, 0xf, false));
    return (uint32_t)__shfl((int)v, (int)((threadIdx.x & 63) | 15), 64);  // lane 15 of the row holds it
}
__device__ __forceinline__ uint32_t row_excl_sum(uint32_t v) {  // exclusive prefix sum inside each row of 16 lanes
    uint32_t inc = v;
    inc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x111, 0xf, 0xf, false);
    inc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x112, 0xf, 0xf, false);
    inc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x114, 0xf, 0xf, false);
    inc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x118, 0xf, 0xf, false);
    return inc - v;
}

__device__ __forceinline__ void d_exc_quad(const DevTables& T, const BatchArgs& A, const Workspace& W, uint32_t vblock,
                                           uint32_t vgrid, uint8_t* lds) {
    uint32_t* const Sq = reinterpret_cast<uint32_t*>(lds);
    uint32_t* const Mq = Sq + 4 * QUAD_UNITS;
    const int lane = threadIdx.x & 63, g = lane >> 4, l = lane & 15, gl0 = lane & 48;  // group, lane in group, its lane 0
    uint32_t* Sg = Sq + g * QUAD_UNITS;
    uint32_t* Mg = Mq + g * QUAD_UNITS;
    const uint32_t n_list = quad_list_len(W);
    for (uint32_t base = vblock * 4; base < n_list; base += vgrid * 4) {
        const uint32_t li = base + g;
        bool have = li < n_list;
        uint32_t idx = 0;
        ExcRec rec{};
        if (have) {
            idx = quad_list_at(W, li);
            rec = W.exc[idx];
        }
        int n = 0, na = 0;
        int64_t gbase = 0;
        if (have) {
            const int64_t ws = rec.ws;
            const int64_t d = T.has_prefix ? doc_of(A, W, ws, rec.tile) : 0;  // (needed for the prefix and its room in exc_tok only)
            const bool docfirst = T.has_prefix && word_is_first(A, ws, A.offsets[d]);
            const bool with_prefix = T.has_prefix && docfirst;
            const bool alone = with_prefix && doc_begins_with_space(A, ws);
            const int kp = (with_prefix && !alone) ? T.n_prefix : 0;
            na = alone ? T.n_prefix_alone : 0;
            gbase = ws * T.unit_scale + (int64_t)W.pad_per_doc * (docfirst ? d : d + 1);
            {
                n = kp + rec.len;  // <= QUAD_UNITS: the ends pass checked
                for (int i = l; i < kp; i += 16) Sg[i] = T.prefix_syms[i];
                for (int i = l; i < rec.len; i += 16) Sg[kp + i] = T.item_sym[A.bytes[ws + i]];
            }
        }
        wave_sync();
        // my 16 units: pair results and liveness
        const int lo = 16 * l;
        uint32_t live16 = 0;
        if (have) {
            const int cnt = n - lo;
            live16 = cnt >= 16 ? 0xFFFFu : cnt > 0 ? ((1u << cnt) - 1u) : 0u;
        }
        for (int k0 = 0; k0 < 16; k0 += 4) {  // four lookups in flight
            PairProbe pr[4];
            uint32_t sy[5];
#pragma unroll
            for (int j = 0; j < 5; j++) sy[j] = (have && lo + k0 + j < n) ? Sg[lo + k0 + j] : 0u;
#pragma unroll
            for (int j = 0; j < 4; j++) pr[j] = pair_issue(T, sy[j], sy[j + 1]);
#pragma unroll
            for (int j = 0; j < 4; j++)
                if (have && lo + k0 + j < n)
                    Mg[lo + k0 + j] = (lo + k0 + j + 1 < n) ? pair_resolve(T, pr[j], sy[j], sy[j + 1]) : SYM_NONE;
        }
        wave_sync();
        bool dirty = true;
        uint32_t mybest = 0xFFFFFFFFu;  // (merged symbol = rank) << 8 | unit index, over my live units
        for (;;) {
            if (dirty) {
                mybest = 0xFFFFFFFFu;
                for (uint32_t c = live16; c; c &= c - 1) {
                    const int i = lo + __builtin_ctz(c);
                    const uint32_t m = Mg[i];
                    if (m != SYM_NONE) mybest = min(mybest, (m << 8) | (uint32_t)i);
                }
                dirty = false;
            }
            const uint32_t gbest = row_min_u32(have ? mybest : 0xFFFFFFFFu);
            const bool act = gbest != 0xFFFFFFFFu;
            if (!__any(act)) break;
            // neighbours of the pair, from the per-lane liveness masks of the group (all shuffles unconditional)
            const int p = (int)(gbest & 0xFFu);
            const uint32_t ne = (uint32_t)(__ballot(live16 != 0) >> gl0) & 0xFFFFu;  // lanes of my group with live units
            auto live_of = [&](int x) -> uint32_t { return (uint32_t)__shfl((int)live16, gl0 | (x & 15), 64); };
            auto next_after = [&](int pos) -> int {
                const int lx = pos >> 4;
                const uint32_t a = live_of(lx) & ~((2u << (pos & 15)) - 1u) & 0xFFFFu;
                const uint32_t m = ne & ~((2u << lx) - 1u) & 0xFFFFu;
                const int ly = m ? __builtin_ctz(m) : 0;
                const uint32_t lv = live_of(ly);
                return a ? (pos & ~15) + __builtin_ctz(a) : (m && lv) ? 16 * ly + __builtin_ctz(lv) : -1;
            };
            auto prev_before = [&](int pos) -> int {
                const int lx = pos >> 4;
                const uint32_t a = live_of(lx) & ((1u << (pos & 15)) - 1u);
                const uint32_t m = ne & ((1u << lx) - 1u);
                const int ly = m ? 31 - __builtin_clz(m) : 0;
                const uint32_t lv = live_of(ly);
                return a ? (pos & ~15) + (31 - __builtin_clz(a)) : (m && lv) ? 16 * ly + (31 - __builtin_clz(lv)) : -1;
            };
            const int q = next_after(act ? p : 0);           // the unit the merge consumes
            const int q2 = next_after(q >= 0 ? q : 0);
            const int p0 = prev_before(act ? p : 0);
            const uint32_t merged = gbest >> 8;
            const uint32_t sr = (act && q >= 0 && q2 >= 0) ? Sg[q2] : 0u;
            const uint32_t sl = (act && p0 >= 0) ? Sg[p0] : 0u;
            const PairProbe prr = pair_issue(T, merged, sr), prl = pair_issue(T, sl, merged);
            const uint32_t mr = (act && q >= 0 && q2 >= 0) ? pair_resolve(T, prr, merged, sr) : SYM_NONE;
            const uint32_t ml = (act && p0 >= 0) ? pair_resolve(T, prl, sl, merged) : SYM_NONE;
            wave_sync();  // everybody has read S before the owners write
            if (act && q >= 0) {
                if (l == (p >> 4)) {
                    Sg[p] = merged;
                    Mg[p] = mr;
                    dirty = true;
                }
                if (l == (q >> 4)) {
                    live16 &= ~(1u << (q & 15));
                    Mg[q] = SYM_NONE;
                    dirty = true;
                }
                if (p0 >= 0 && l == (p0 >> 4)) {
                    Mg[p0] = ml;
                    dirty = true;
                }
            }
            wave_sync();
        }
        // survivors in order: alone ids, then each lane's live units at its row prefix
        const uint32_t mine = (uint32_t)__popc(live16);
        const uint32_t before = row_excl_sum(have ? mine : 0u);
        const uint32_t total = (uint32_t)__shfl((int)(before + (have ? mine : 0u)), lane | 15, 64);
        if (have) {
            int32_t* out = W.exc_tok + gbase;
            for (int i = l; i < na; i += 16) out[i] = T.prefix_alone_ids[i];
            uint32_t k = (uint32_t)na + before;
            for (uint32_t c = live16; c; c &= c - 1) out[k++] = sym_to_id(T, Sg[lo + __builtin_ctz(c)]);
            if (l == 0) {
                rec.cnt = (uint32_t)na + total;
                rec.tok_base = gbase;
                W.exc[idx] = rec;
                atomicAdd(&W.tile_count[rec.tile], rec.cnt);
            }
        }
        wave_sync();
    }
}

// The splitter's window of a wavefront: word starts of EXC_CHUNK positions from `base`, computed with the bytes of document
// `doc` (the others read as zero), one bit per position.  Exception words of one tile follow each other closely, so the
// window that held one word's end usually holds the next one's as well and is not staged again.
struct EndsWin {
    int64_t base = -1, doc = -1;
    unsigned long long bits[EXC_CHUNK / 64];
};
// End of a word whose end its tile could not see (more than 63 bytes, or beyond the tile's window): the splitter's rule
// applied 256 positions at a time (src/parser.c:24-183 as in k_tiles' exact form), or -- regex pre-token path -- the next
// start bit of the host's bitmap.  One wavefront; sb / scode / docm are its LDS scratch, cw its window (above).
// -> end offset; *too_large when the word passes the reference's limit (core.c:402-407).
__device__ int64_t exc_word_end(const DevTables& T, const BatchArgs& A, int64_t ws, int64_t d, int64_t ds, int64_t de, uint8_t* sb,
                                uint8_t* scode, uint32_t* docm, int lane, bool* too_large, EndsWin& cw, bool seams = true) {
    int64_t we = -1;
    *too_large = false;
    if (A.word_bits) {
        for (int64_t wi = (ws + 1) >> 5; we < 0; wi += 64) {
            const int64_t w = wi + lane;
            uint32_t bits = (w << 5) <= A.n_bytes ? A.word_bits[w] : 0u;
            if (w == ((ws + 1) >> 5)) bits &= ~0u << ((ws + 1) & 31);
            const unsigned long long bal = __ballot(bits != 0);
            if (bal) {
                const int l0 = __builtin_ctzll(bal);
                const uint32_t b0 = (uint32_t)__shfl((int)bits, l0, 64);
                we = ((wi + l0) << 5) + __builtin_ctz(b0);
            } else if ((wi << 5) > A.n_bytes) {
                we = A.n_bytes;  // (cannot happen: the host sets the bit at n_bytes)
            }
        }
        if (we - ws > MAX_WORD_BYTES) *too_large = true;
        return we;
    }
    for (int64_t pos = ws + 1; we < 0;) {  // (every value here is the same in all lanes)
        if (pos - ws > MAX_WORD_BYTES + 1) { *too_large = true; break; }
        if (!(cw.doc == d && pos >= cw.base && pos < cw.base + EXC_CHUNK)) {
            const int64_t base = pos;
            const int64_t g0 = base - 16;  // global offset of window index 0
            for (int i = lane; i < EXC_WIN; i += 64) {
                const int64_t q = g0 + i;
                sb[i] = (q >= ds && q < de) ? A.bytes[q] : (uint8_t)0;
            }
            if (lane < EXC_WIN / 32 + 1) docm[lane] = 0;
            wave_sync();
            if (lane == 0) {
                if (ds >= g0 && ds < g0 + EXC_WIN) docm[(ds - g0) >> 5] |= 1u << ((ds - g0) & 31);
                if (de >= g0 && de < g0 + EXC_WIN) docm[(de - g0) >> 5] |= 1u << ((de - g0) & 31);
            }
            wave_sync();
            for (int i = lane; i < EXC_WIN; i += 64)
                scode[i] = (i >= 4 && i < EXC_WIN - 4) ? code_at(sb, docm, i) : (uint8_t)C_BAD;
            wave_sync();
#pragma unroll
            for (int r = 0; r < EXC_CHUNK / 64; r++) {
                const int64_t q = base + 64 * r + lane;
                const int wi = 16 + 64 * r + lane;
                bool st = (q <= de) && word_starts(scode, docm, wi);
                if (seams && T.seam_on && q < de && sb[wi] >= 0xE0u)  // a seam starts a word as well (k_tiles, phase 3)
                    st = st || !((T.seam_hi[sb[wi - 1]] >> (sb[wi] & 31u)) & 1u);
                cw.bits[r] = __ballot(st);
            }
            wave_sync();
            cw.base = base;
            cw.doc = d;
        }
        const int rel = (int)(pos - cw.base);
#pragma unroll
        for (int r = 0; r < EXC_CHUNK / 64; r++) {
            if (we >= 0 || 64 * (r + 1) <= rel) continue;
            unsigned long long m = cw.bits[r];
            if (rel > 64 * r) m &= ~0ull << (rel - 64 * r);
            if (m) we = cw.base + 64 * r + __builtin_ctzll(m);
        }
        pos = cw.base + EXC_CHUNK;
    }
    return we;
}

// d_exc_ends: the words whose end their tile could not see.  One wavefront per tile that has exception words (the list
// k_tiles made): for each such word of the tile the end is found and stored, and the word goes on k_exc_quad's list (16
// lanes per word, up to QUAD_UNITS units, byte mode with rank == symbol order) or on k_exc's (a wavefront per word) --
// collected per tile, one atomic per tile and list.  A word over the reference's limit cuts its document (no list).
constexpr int ENDS_LIST = TILE_BYTES / 2 + 4;  // a tile has at most that many exception words
struct EndsLds {
    __attribute__((aligned(16))) uint8_t sb[EXC_WIN];
    uint8_t scode[EXC_WIN];
    uint32_t docm[EXC_WIN / 32 + 1];
    uint32_t lq[ENDS_LIST], lm[ENDS_LIST], lw[ENDS_LIST];
};
constexpr uint32_t ENDS_SHARE = 1;  // wavefronts that share the words of one tile (one: the splitter's window is reused from word to word)
__device__ __forceinline__ void d_exc_ends(const DevTables& T, const BatchArgs& A, const Workspace& W, uint32_t vblock,
                                           uint32_t vgrid, uint8_t* lds) {
    EndsLds& L = *reinterpret_cast<EndsLds*>(lds);
    uint32_t* const lq = L.lq;
    uint32_t* const lm = L.lm;
    uint32_t* const lw = L.lw;
    const int lane = threadIdx.x;
    const uint32_t n_tiles_exc = W.counters[1];
    // The two lists' places are claimed once per WAVEFRONT, not per tile: the entries of its tiles wait in LDS (a tile's
    // fit behind what is there, or the lists are written out first).  One atomic per tile on the same two words was
    // ~50 k same-address atomics for 800 k words of 70-120 letters -- at ~12 ns each most of k_exc_a's 0.65 ms.
    // (lq: lists 0 and 1 from its two ends, lm: lists 2 and 3, lw: d_exc's -- exc_list_of)
    uint32_t nl[5] = {0, 0, 0, 0, 0};  // (the same in every lane)
    auto lds_slot = [&](int list, uint32_t k) -> uint32_t* {
        return list == 0 ? lq + k : list == 1 ? lq + (ENDS_LIST - 1 - k) : list == 2 ? lm + k : list == 3 ? lm + (ENDS_LIST - 1 - k) : lw + k;
    };
    auto flush = [&]() {
        wave_sync();
#pragma unroll
        for (int l = 0; l < 5; l++) {
            if (nl[l] == 0) continue;
            uint32_t at = 0;
            if (lane == 0) at = atomicAdd(exc_list_count(W, l), nl[l]);
            at = __shfl(at, 0, 64);
            for (uint32_t i = lane; i < nl[l]; i += 64) *exc_list_slot(W, l, at + i) = *lds_slot(l, i);
            nl[l] = 0;
        }
        wave_sync();
    };
    for (uint32_t ti = vblock / ENDS_SHARE; ti < n_tiles_exc; ti += vgrid / ENDS_SHARE) {
        const uint32_t tile = W.exc_tiles[ti];
        const uint32_t first = W.tile_exc_first[tile], nexc = W.tile_nexc[tile];
        if (max(max(nl[0] + nl[1], nl[2] + nl[3]), nl[4]) + nexc > (uint32_t)ENDS_LIST) flush();
        // The tile's records sixty-four at a time, ONE LANE PER WORD of unknown length.  Its end is the first word start of
        // the tiles behind (tile_first_start: what every tile found among its own 1024 positions, seams and document starts
        // included -- my tile saw none between the word and its window's end): one load for a word that ends in the next
        // tile.  (Rounds 2-4 staged and classified the text again here, 256 positions at a time, a wavefront per word:
        // k_exc_a was 1.0 of 3.3 ms on CJK paragraphs without seams, 0.5 of 2.3 on words of 70-120 letters.)
        static_assert(ENDS_SHARE == 1, "one wavefront per tile");
        const unsigned long long below = (1ull << lane) - 1ull;
        for (uint32_t e0 = 0; e0 < nexc; e0 += 64) {
            const uint32_t idx = first + e0 + lane;
            const bool unk = e0 + lane < nexc && (int64_t)idx < W.cap_exc && W.exc[idx].len < 0;
            int list = -1;
            if (unk) {
                const int64_t ws = W.exc[idx].ws;
                int64_t we = -1;
                // (a word over the reference's limit of MAX_WORD_BYTES ends the search: it is refused below whatever its end)
                const int64_t u_end = min((int64_t)A.n_tiles, (int64_t)tile + 3 + MAX_WORD_BYTES / TILE_BYTES);
                for (int64_t u = (int64_t)tile + 1; u < u_end; u++) {
                    const uint32_t fs = W.tile_first_start[u];
                    if (fs != 0xFFFFu) { we = u * TILE_BYTES + fs; break; }
                }
                if (we < 0) we = u_end < A.n_tiles ? ws + MAX_WORD_BYTES + 1 : A.n_bytes;  // (no start within the limit / up to the text's end)
                if (we > A.n_bytes) we = A.n_bytes;
                const int64_t nb = we - ws;
                if (nb > MAX_WORD_BYTES) {
                    const int64_t d = doc_of(A, W, ws, tile), ds = A.offsets[d];
                    raise(A.err, HUTK_E_WORD_TOO_LARGE);
                    if (A.status) A.status[d] = HUTK_DOC_WORD_TOO_LARGE;
                    W.exc[idx].cnt = 0;
                    W.exc[idx].tok_base = -(ws - ds) - 1;  // where the document is cut (negative marks "no ids")
                } else {
                    W.exc[idx].len = (int32_t)nb;
                    // (the list by the length with the prefix, whether or not this word gets it: d_exc_group_fast<NW> takes its list whole)
                    list = exc_list_of(T, nb + T.n_prefix);
                }
            }
#pragma unroll
            for (int l = 0; l < 5; l++) {
                const unsigned long long bl = __ballot(list == l);
                if (list == l) *lds_slot(l, nl[l] + (uint32_t)__popcll(bl & below)) = idx;
                nl[l] += (uint32_t)__popcll(bl);
            }
        }
    }
    if (nl[0] | nl[1] | nl[2] | nl[3] | nl[4]) flush();
}

// d_exc: the words that need a whole wavefront (k_exc's list; their lengths are known by now): first entry by block
// index, further ones from a device cursor.
// LU: units of the two LDS arrays -- EXC_LDS_UNITS (1024), or 2048 where the words of up to 1024 units have gone to
// d_exc_group_fast and the LDS they needed is free (k_exc_b<true>): words of up to 2046 units merge in LDS then, where at
// 1025 they used to fall to the arrays in HBM (CJK paragraphs of 1025..1200 bytes under a dense vocabulary: most of k_exc_b).
template <int LU = EXC_LDS_UNITS>
__device__ __forceinline__ void d_exc(const DevTables& T, const BatchArgs& A, const Workspace& W, uint32_t vblock,
                                      uint32_t vgrid, uint8_t* lds) {
    static_assert(LU == EXC_LDS_UNITS || LU == 2048, "bpe_wave_big's chunk arrays are EXC_LDS_UNITS entries of them");
    uint32_t* const Sl = reinterpret_cast<uint32_t*>(lds);
    uint32_t* const Ml = Sl + LU;

    const int lane = threadIdx.x & 63;  // (k_exc_b runs two of these per workgroup, each wavefront on its own: no s_barrier in here)
    const uint32_t n_list = W.counters[5];
    for (uint32_t round = 0;; round++) {
        uint32_t li = vblock;
        if (round) {
            if (lane == 0) li = vgrid + atomicAdd(&W.counters[2], 1u);
            li = (uint32_t)__builtin_amdgcn_readfirstlane((int)li);
        }
        if (li >= n_list) break;
        const uint32_t idx = W.exc_wave[li];
        ExcRec rec = W.exc[idx];
        const int64_t ws = rec.ws;
        const int64_t d = T.has_prefix ? doc_of(A, W, ws, rec.tile) : 0;  // (needed for the prefix and its room in exc_tok only)
        const int64_t nb = rec.len;
        const bool docfirst = T.has_prefix && word_is_first(A, ws, A.offsets[d]);
        const bool with_prefix = T.has_prefix && docfirst;
        const bool alone = with_prefix && doc_begins_with_space(A, ws);  // core.c:365-366, 421-446
        const int kp = (with_prefix && !alone) ? T.n_prefix : 0;
        const int64_t gbase = ws * T.unit_scale + (int64_t)W.pad_per_doc * (docfirst ? d : d + 1);  // unit_scale slots per byte: room for an expanded word

        // unit count
        int64_t n_units;
        // units of the item that starts at byte i (0 inside a character): one, unless its replacement has several or none
        auto item_units = [&](int64_t i, uint32_t b) -> uint32_t {
            if (i >= nb || (!T.is_byte_encoder && is_cont(b))) return 0u;
            return (T.is_byte_encoder || T.item_direct[b]) ? T.item_units_off[b + 1] - T.item_units_off[b] : 1u;
        };
        if (T.has_multi) {
            int64_t cnt = 0;
            for (int64_t i0 = 0; i0 < nb; i0 += 64) {
                const int64_t i = i0 + lane;
                uint32_t c = item_units(i, i < nb ? A.bytes[ws + i] : 0x80u), tot;
                (void)wave_excl_scan(c, lane, &tot);
                cnt += tot;
            }
            n_units = cnt;
        } else if (T.is_byte_encoder) {
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
        const bool in_lds = n <= (LU == EXC_LDS_UNITS ? EXC_LDS_UNITS : FAST_LDS_UNITS);
        LdsArr Sl_a{Sl}, Ml_a{Ml};
        HbmArr Sg_a{W.exc_sym + gbase}, Mg_a{W.exc_mrg + gbase};

        // initial symbols
        for (int i = lane; i < kp; i += 64) {
            if (in_lds) Sl_a.set(i, T.prefix_syms[i]); else Sg_a.set(i, T.prefix_syms[i]);
        }
        if (T.has_multi) {
            // expansion: every item writes its units behind those of the items in front of it
            int64_t ubase = kp;
            for (int64_t i0 = 0; i0 < nb; i0 += 64) {
                const int64_t i = i0 + lane;
                const uint32_t b = i < nb ? A.bytes[ws + i] : 0x80u;
                const uint32_t c = item_units(i, b);
                uint32_t tot;
                const int64_t u = ubase + wave_excl_scan(c, lane, &tot);
                if (i < nb && (T.is_byte_encoder || T.item_direct[b])) {
                    const uint32_t* units = T.item_units + T.item_units_off[b];
                    for (uint32_t k = 0; k < c; k++) {
                        if (in_lds) Sl_a.set(u + k, units[k]); else Sg_a.set(u + k, units[k]);
                    }
                } else if (c) {  // a multi-byte character without replacement
                    const int L = (b >= 0xF0u) ? 4 : (b >= 0xE0u) ? 3 : (b >= 0xC0u) ? 2 : 1;
                    uint32_t sym = SYM_UNK;
                    if (L == 1 || b >= 0xF8u || i + L > nb) {
                        raise(A.err, HUTK_E_INVALID_UTF8);
                    } else {
                        uint32_t packed = b | ((uint32_t)A.bytes[ws + i + 1] << 8);
                        if (L > 2) packed |= (uint32_t)A.bytes[ws + i + 2] << 16;
                        if (L > 3) packed |= (uint32_t)A.bytes[ws + i + 3] << 24;
                        sym = char_lookup(T, packed);
                    }
                    if (in_lds) Sl_a.set(u, sym); else Sg_a.set(u, sym);
                } else if (i == 0 && i < nb) {
                    raise(A.err, HUTK_E_INVALID_UTF8);  // a word cannot begin inside a character
                }
                ubase += tot;
            }
        } else if (T.is_byte_encoder) {
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
        wave_wg_sync();

        // In LDS: short words by shifting the tail left after every merge (bpe_wave), longer ones by the same dead-unit
        // marks and per-chunk best keys as the words in HBM (chunks of 64 units, at most 16 of them: a merge costs three
        // chunk rescans instead of a shift of half the word, barriers and all)
        __shared__ uint32_t s_l1_all[2][2 * (LU / 64)];
        uint32_t* const s_l1 = s_l1_all[threadIdx.x >> 6];
        const bool fast = in_lds && T.rank_is_sym && n >= 2 && HUTK_LAB_EXC_FAST;  // (rank == symbol order: 32-bit keys, bpe_wave_fast)
        const int64_t left = !in_lds ? bpe_wave_big(T, Sg_a, Mg_a, Sl, Ml, n, lane)
                           : fast ? bpe_wave_fast(T, Sl, Ml, s_l1, (int)n, lane)
                           : n > EXC_SHIFT_MAX ? bpe_wave_big(T, Sl_a, Ml_a, s_l1, s_l1 + LU / 64, n, lane)
                                               : bpe_wave(T, Sl_a, Ml_a, n, lane);
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
        wave_wg_sync();
    }
}

// The exception words in TWO launches (their kernels do nothing at all in most batches, and a launch is ~5 us):
//   k_exc_a  workgroups [0, n_medium): d_exc_medium, words of known length up to 63 bytes, one per lane;
//            the others: d_exc_ends, the lengths of the words whose end no tile saw, and the two lists for ...
//   k_exc_b  workgroups [0, EXB_QUAD): d_exc_quad, sixteen lanes per word; the others: d_exc, a wavefront per word
// The two roles of a launch share one LDS area (a role's arrays would otherwise be allocated for every workgroup).
#ifndef HUTK_EXB_WAVE
#define HUTK_EXB_WAVE 4864  // (19 per CU: what the role's 8.4 KB of LDS lets a CU hold; 4096: -6 % on words of 300-900 letters)
#endif
#ifndef HUTK_EXB_QUAD
#define HUTK_EXB_QUAD 5120
#endif
constexpr int EXA_MEDIUM16 = 2560, EXA_MEDIUM32 = 1280, EXA_ENDS = 4096, EXB_QUAD = HUTK_EXB_QUAD, EXB_WAVE = HUTK_EXB_WAVE;
constexpr size_t cmax(size_t a, size_t b) { return a > b ? a : b; }
template <typename SymT>
__global__ __launch_bounds__(64) void k_exc_a(DevTables T, BatchArgs A, Workspace W, uint32_t n_medium) {
    __shared__ __attribute__((aligned(16))) uint8_t lds[cmax(cmax(2 * MEDIUM_UNITS * 64 * sizeof(SymT), sizeof(EndsLds)),
                                                             sizeof(SymT) == 2 ? 64 * MEDIUM_ROW * 4 : 0)];
    if (W.counters[0] == 0) return;  // no exception word in this batch
    if (blockIdx.x < n_medium) {
        if (sizeof(SymT) == 2 && T.rank_is_sym) d_exc_lane_fast<1, 64>(T, A, W, blockIdx.x, n_medium, lds);
        else d_exc_medium<SymT>(T, A, W, blockIdx.x, n_medium, lds);
    }
    else d_exc_ends(T, A, W, blockIdx.x - n_medium, gridDim.x - n_medium, lds);
}
// FAST: 16-bit symbols with rank == symbol order: the quad list goes to d_exc_lane_fast (one lane per word: first the words
// of up to 128 units, then the longer ones) instead of d_exc_quad
#if HUTK_LAB_EXC_GROUP
constexpr size_t LANE_FAST_LDS = cmax(32 * (128 + 4) * 4, 16 * (256 + 4) * 4);  // d_exc_group_fast: 32 / 16 / 8 / 4 rows (the first the largest)
#else
constexpr size_t LANE_FAST_LDS = cmax(16 * (128 + 4) * 4, 8 * (256 + 4) * 4);
#endif
// k_exc_b<true> (16-bit symbols, rank == symbol order): 2304 workgroups of ONE wavefront, 16.5 KB of LDS each, nine per
// CU, all resident: each walks the four lists of d_exc_group_fast and then d_exc's (words beyond 1024 units, in the same
// LDS: up to 2046 units).  k_exc_b<false> (other vocabularies): workgroups of TWO wavefronts that never meet, 8 KB of LDS
// each (d_exc_quad, d_exc: 18 per CU) -- nothing in these roles is a workgroup barrier (wave_wg_sync).
constexpr size_t EXB_WAVE_LDS = cmax(2 * 4 * QUAD_UNITS * 4, 2 * EXC_LDS_UNITS * 4);  // per wavefront of d_exc_quad / d_exc<1024>
constexpr int EXB_FAST_WGS = 2304;
#ifndef HUTK_EXB_EU
#define HUTK_EXB_EU 5
#endif
template <bool FAST>
__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(FAST ? 3 : HUTK_EXB_EU))) void k_exc_b(DevTables T, BatchArgs A, Workspace W) {
    __shared__ __attribute__((aligned(16))) uint8_t lds[FAST ? cmax(HUTK_LAB_EXC_GROUP ? LANE_FAST_LDS : 2 * LANE_FAST_LDS, 2 * 2048 * 4) : 2 * EXB_WAVE_LDS];
    if (W.counters[0] == 0) return;
    const uint32_t wv = threadIdx.x >> 6;
    if (FAST) {
#if HUTK_LAB_EXC_GROUP
        d_exc_group_fast<2>(T, A, W, blockIdx.x, EXB_FAST_WGS, lds);
        wave_sync();
        d_exc_group_fast<4>(T, A, W, blockIdx.x, EXB_FAST_WGS, lds);
        wave_sync();
        d_exc_group_fast<8>(T, A, W, blockIdx.x, EXB_FAST_WGS, lds);
        wave_sync();
        d_exc_group_fast<16>(T, A, W, blockIdx.x, EXB_FAST_WGS, lds);
#else
        d_exc_lane_fast<2, 16>(T, A, W, blockIdx.x, EXB_FAST_WGS, lds);
        wave_sync();
        d_exc_lane_fast<4, 8>(T, A, W, blockIdx.x, EXB_FAST_WGS, lds);
#endif
        wave_sync();
        d_exc<2048>(T, A, W, blockIdx.x, EXB_FAST_WGS, lds);
    } else if (blockIdx.x < (uint32_t)EXB_QUAD / 2) {
        d_exc_quad(T, A, W, 2 * blockIdx.x + wv, EXB_QUAD, lds + wv * EXB_WAVE_LDS);
    } else {
        d_exc<EXC_LDS_UNITS>(T, A, W, 2 * (blockIdx.x - EXB_QUAD / 2) + wv, EXB_WAVE, lds + wv * EXB_WAVE_LDS);
    }
}

// one-off: merge a short symbol sequence (the prefix encoded as its own word)
__global__ __launch_bounds__(64) void k_bpe_symbols(DevTables T, uint32_t* syms, int n, int32_t* ids_out,
                                                     int32_t* n_out) {
    __shared__ uint32_t Sl[EXC_LDS_UNITS];
    __shared__ uint32_t Ml[EXC_LDS_UNITS];
    const int lane = threadIdx.x;
    if (n > EXC_LDS_UNITS) n = EXC_LDS_UNITS;
    for (int i = lane; i < n; i += 64) Sl[i] = syms[i];
    __syncthreads();
    const int64_t left = bpe_wave(T, LdsArr{Sl}, LdsArr{Ml}, n, lane);
    for (int i = lane; i < left; i += 64) {
        ids_out[i] = sym_to_id(T, Sl[i]);
        syms[i] = Sl[i];
    }
    if (lane == 0) *n_out = (int32_t)left;
}

// ------------------------------------------------------------------------
// k_scan: exclusive scan of tile_count -> tile_base (and the grand total at tile_base[n_tiles]) in ONE launch.  A workgroup
// scans SCAN_BLOCK tiles, publishes its total at once and its inclusive prefix as soon as it knows it; it adds up its
// predecessors' totals back to the nearest published prefix (decoupled look-back, as hutk_decode.hip does).  A
// workgroup's place in the chain is a TICKET (counters[7], zeroed by k_pre), not its block index: HIP promises no
// dispatch order, but every smaller ticket belongs to a workgroup that has started.  Flag and value share one 64-bit
// word, so relaxed agent-scope atomics suffice.  The spin is bounded all the same (HUTK_E_DEVICE instead of a hang).
// ------------------------------------------------------------------------
constexpr int SCAN_BLOCK = 2048, SCAN_THREADS = 256, SCAN_PER_THREAD = SCAN_BLOCK / SCAN_THREADS;
constexpr unsigned long long SC_MASK = 3ull << 62, SC_TOTAL = 1ull << 62, SC_PREFIX = 2ull << 62;

__device__ __forceinline__ int64_t block_excl_scan(int64_t v, int64_t* sh, int tid, int n, int64_t* total) {
    sh[tid] = v;
    __syncthreads();
    for (int off = 1; off < n; off <<= 1) {
        const int64_t o = tid >= off ? sh[tid - off] : 0;
        __syncthreads();
        sh[tid] += o;
        __syncthreads();
    }
    *total = sh[n - 1];
    return sh[tid] - v;
}

__global__ __launch_bounds__(SCAN_THREADS) void k_scan(BatchArgs A, Workspace W) {
    __shared__ int64_t sh[SCAN_THREADS];
    __shared__ int64_t s_base, s_ticket;
    const int tid = threadIdx.x;
    if (tid == 0) s_ticket = (int64_t)atomicAdd(&W.counters[7], 1u);
    if (tid == 0 && blockIdx.x == 0) W.counters[10] = 0;  // k_pre's sample of this batch has been read by the tile kernels: zero for the next one
    __syncthreads();
    const int64_t blk = s_ticket;
    const int64_t base = blk * SCAN_BLOCK + (int64_t)tid * SCAN_PER_THREAD;
    uint32_t c[SCAN_PER_THREAD];
    int64_t sum = 0;
    for (int k = 0; k < SCAN_PER_THREAD; k++) {
        c[k] = (base + k < A.n_tiles) ? W.tile_count[base + k] : 0u;
        sum += c[k];
    }
    int64_t total;
    int64_t run = block_excl_scan(sum, sh, tid, SCAN_THREADS, &total);
    unsigned long long* st = W.scan_state;
    if (tid == 0 && blk > 0)
        __hip_atomic_store(&st[blk], SC_TOTAL | (unsigned long long)total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tid < 64) {  // the first wavefront looks back, 64 predecessors per round trip, nearest first
        const int lane = tid;
        int64_t excl = 0;
        bool found = blk == 0;
        for (int64_t hi = blk - 1; !found; hi -= 64) {
            unsigned long long v;
            uint32_t spins = 0;
            for (;;) {
                const int64_t p = hi - lane;
                v = SC_PREFIX;  // before block 0: an empty prefix
                if (p >= 0) v = __hip_atomic_load(&st[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (__all((v & SC_MASK) != 0)) break;
                if (++spins > (1u << 22)) {  // seconds: a predecessor never published (it cannot be waiting for us); fail loudly
                    raise(A.err, HUTK_E_DEVICE);
                    v = SC_PREFIX;
                    break;
                }
            }
            const unsigned long long has_prefix = __ballot((v & SC_MASK) == SC_PREFIX);
            const int stop = has_prefix ? __builtin_ctzll(has_prefix) : 63;  // up to and including the nearest prefix
            int64_t part = (lane <= stop) ? (int64_t)(v & ~SC_MASK) : 0;
            for (int o = 32; o; o >>= 1) part += __shfl_xor(part, o, 64);
            excl += part;
            found = has_prefix != 0;
        }
        if (lane == 0) {
            __hip_atomic_store(&st[blk], SC_PREFIX | (unsigned long long)(excl + total), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
            s_base = excl;
            if (blk == (int64_t)gridDim.x - 1) W.tile_base[A.n_tiles] = excl + total;
        }
    }
    __syncthreads();
    run += s_base;
    for (int k = 0; k < SCAN_PER_THREAD; k++) {
        if (base + k < A.n_tiles) W.tile_base[base + k] = run;
        run += c[k];
    }
}

// ------------------------------------------------------------------------
// k_gather: tile runs and exception words -> ids_out
// ------------------------------------------------------------------------
constexpr int GATHER_EXC_LDS = 1024;  // >= TILE_BYTES: at most one exception word per byte

constexpr int GATHER_THREADS = 64;
// Tiles without exception words (nearly all): a plain copy, symbol -> id on the way.  One wavefront takes
// GATHER_TILES consecutive tiles (their output is one contiguous stretch of ids_out); a workgroup per tile
// was bound by workgroup dispatch, not by memory.
constexpr int GATHER_TILES = 4, GATHER_WAVES = 4, GATHER_UNROLL = 5;
template <typename RunT>
__device__ __forceinline__ void d_gather(const DevTables& T, const BatchArgs& A, const Workspace& W, int64_t vwave) {
    const int lane = threadIdx.x & 63;
    const int64_t first = vwave * GATHER_TILES;  // vwave: index of this wavefront among all that gather
    // everything is issued before anything is consumed: the metadata of all tiles, then up to
    // GATHER_UNROLL x 64 symbols of each tile (a tile has ~250), then the stores
    int64_t base[GATHER_TILES];
    uint32_t dense[GATHER_TILES];
#pragma unroll
    for (int t = 0; t < GATHER_TILES; t++) {
        const int64_t tile = first + t;
        const bool ok = tile < A.n_tiles;
        base[t] = ok ? W.tile_base[tile] : 0;
        dense[t] = ok ? W.tile_dense[tile] : 0u;
        if (ok && W.tile_nexc[tile]) dense[t] = 0;  // k_gather_exc
    }
    RunT v[GATHER_TILES][GATHER_UNROLL];
#pragma unroll
    for (int t = 0; t < GATHER_TILES; t++) {
        if (base[t] + (int64_t)dense[t] > A.ids_cap) {
            if (lane == 0) raise(A.err, HUTK_E_CAPACITY);
            dense[t] = 0;
        }
        const RunT* run = reinterpret_cast<const RunT*>(W.run) + (first + t) * RUN_STRIDE;
#pragma unroll
        for (int j = 0; j < GATHER_UNROLL; j++) {
            const uint32_t k = lane + 64 * j;
            v[t][j] = k < dense[t] ? run[k] : (RunT)0;
        }
    }
#pragma unroll
    for (int t = 0; t < GATHER_TILES; t++) {
        int32_t* out = A.ids_out + base[t];
#pragma unroll
        for (int j = 0; j < GATHER_UNROLL; j++) {
            const uint32_t k = lane + 64 * j;
            if (k < dense[t]) out[k] = sym_to_id(T, (uint32_t)v[t][j]);
        }
        const RunT* run = reinterpret_cast<const RunT*>(W.run) + (first + t) * RUN_STRIDE;
        for (uint32_t k = lane + 64 * GATHER_UNROLL; k < dense[t]; k += 64) out[k] = sym_to_id(T, (uint32_t)run[k]);
    }
}

// Tiles with exception words, from the list k_tiles made: their ids are interleaved with the dense run.
// CAP: exception words of a tile the wavefront's LDS arrays hold.  k_finish runs FOUR wavefronts per workgroup on tiles of
// up to GATHER_EXC_SMALL words (BIG = -1: the others are skipped), then its first wavefront on the rest with the four
// areas as one (BIG = 1: only tiles of more than GATHER_EXC_SMALL words) -- a tile is eight dependent stages of a few
// loads each, so the wavefronts in flight set the pace, and 18 KB for each meant eight per CU (round 4: 0.40 -> 0.2 ms on
// 80 k tiles of 70-120-letter words).  BIG = 0: every tile (k_tail_small).
constexpr int GATHER_EXC_SMALL = 256;
constexpr size_t gather_exc_lds(int cap) { return (size_t)cap * 4 + ((size_t)cap + 1) * 4 + 4 + (size_t)cap * 8 + (size_t)cap * 2; }
template <int CAP, int BIG>
__device__ __forceinline__ void d_gather_exc(const DevTables& T, const BatchArgs& A, const Workspace& W, uint32_t vblock,
                                             uint32_t vgrid, uint8_t* lds) {
    int64_t* const e_tok = reinterpret_cast<int64_t*>(lds);                      // (the 8-byte array first: alignment)
    uint32_t* const e_pos = reinterpret_cast<uint32_t*>(e_tok + CAP);
    uint32_t* const e_cum = e_pos + CAP;                                          // CAP + 1 (+ 1 unused: keeps e_ws 8-byte aligned)
    uint16_t* const e_ws = reinterpret_cast<uint16_t*>(e_cum + CAP + 2);         // start of the word inside the tile
    const int tid = threadIdx.x & 63;
    const uint32_t n_list = W.counters[1];
    for (uint32_t li = vblock; li < n_list; li += vgrid) {
    const int64_t tile = W.exc_tiles[li];
    const uint32_t nexc = W.tile_nexc[tile];
    if (BIG != 0 && (nexc > (uint32_t)GATHER_EXC_SMALL) != (BIG > 0)) continue;
    const int64_t base = W.tile_base[tile];
    const uint32_t dense = W.tile_dense[tile];
    // the run holds 16-bit symbols when the LDS arrays do (T.sym16), else 32-bit ones
    const uint16_t* run16 = reinterpret_cast<const uint16_t*>(W.run) + tile * RUN_STRIDE + W.tile_run_start[tile];
    const uint32_t* run32 = W.run + tile * RUN_STRIDE + W.tile_run_start[tile];
    auto run_sym = [&](uint32_t k) -> uint32_t { return T.sym16 ? (uint32_t)run16[k] : run32[k]; };
    if (base + (int64_t)W.tile_count[tile] > A.ids_cap) {
        if (tid == 0) raise(A.err, HUTK_E_CAPACITY);
        continue;
    }
    if ((int64_t)W.tile_exc_first[tile] + (int64_t)nexc > W.cap_exc) continue;  // (k_tiles has raised HUTK_E_MEMORY: records are missing)
    wave_sync();  // the LDS arrays are reused from the previous tile (one wavefront runs this: LDS order is enough)
    ExcRec* recs = W.exc + W.tile_exc_first[tile];
    // positions and id counts of the tile's exception words, one record per lane (all loads in flight
    // together), then the running sum of the counts over LDS
    for (uint32_t e = tid; e < nexc; e += GATHER_THREADS) {
        e_pos[e] = recs[e].wpos;
        e_cum[e + 1] = recs[e].tok_base < 0 ? 0u : recs[e].cnt;  // (a word that was too large has no ids)
        e_tok[e] = recs[e].tok_base;
        e_ws[e] = (uint16_t)(recs[e].ws - tile * TILE_BYTES);
    }
    wave_sync();
    {   // running sum of the counts, 64 at a time (a tile of long-word text has hundreds of them)
        uint32_t carry = 0;
        for (uint32_t e0 = 0; e0 < nexc; e0 += GATHER_THREADS) {
            const uint32_t e = e0 + tid;
            const uint32_t v = e < nexc ? e_cum[e + 1] : 0u;
            uint32_t tot;
            const uint32_t before = wave_excl_scan(v, tid, &tot);
            if (e < nexc) e_cum[e + 1] = carry + before + v;
            carry += tot;
        }
        if (tid == 0) e_cum[0] = 0;
    }
    wave_sync();
    for (uint32_t k = tid; k < dense; k += GATHER_THREADS) {
        // exceptions that come before dense id k: those with wpos <= k
        uint32_t lo = 0, hi = nexc;
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if (e_pos[mid] <= k) lo = mid + 1; else hi = mid;
        }
        A.ids_out[base + k + e_cum[lo]] = sym_to_id(T, run_sym(k));
    }
    for (uint32_t e = tid; e < nexc; e += GATHER_THREADS) recs[e].out_pos = base + e_pos[e] + e_cum[e];
    {   // out_offsets of the documents that start in this tile (d_doc_off leaves them to us: the ids of the exception
        // words in front of a document are a prefix sum we hold in LDS, there they were a loop over the records)
        const int64_t t0 = tile * TILE_BYTES;
        const int64_t t_end = (t0 + TILE_BYTES < A.n_bytes) ? t0 + TILE_BYTES : A.n_bytes;
        for (int64_t d = W.tile_first_doc[tile] + tid; d <= A.n_docs; d += GATHER_THREADS) {
            const int64_t o = A.offsets[d];
            if (o >= t_end) break;
            if (o < t0) continue;
            const uint32_t r = (uint32_t)(o - t0);
            uint32_t lo = 0, hi = nexc;  // records with ws < o
            while (lo < hi) {
                const uint32_t mid = (lo + hi) >> 1;
                if (e_ws[mid] < r) lo = mid + 1; else hi = mid;
            }
            A.out_offsets[d] = base + W.doc_tile_pos[d] + e_cum[lo];
        }
    }
    // the ids of all exception words of the tile as one flat range: element k belongs to the word e with
    // e_cum[e] <= k < e_cum[e + 1] and lands at base + wpos(e) + k; lanes are independent, so the loads of a
    // whole stride are in flight together (one word after the other, each trip waited for its own loads)
    const uint32_t n_flat = e_cum[nexc];
    for (uint32_t k = tid; k < n_flat; k += GATHER_THREADS) {
        uint32_t lo = 0, hi = nexc;  // last e with e_cum[e] <= k
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if (e_cum[mid] <= k) lo = mid; else hi = mid;
        }
        A.ids_out[base + e_pos[lo] + k] = W.exc_tok[e_tok[lo] + (k - e_cum[lo])];
    }
    }
}

// ------------------------------------------------------------------------
// k_doc_off: out_offsets[d] for d in [0, n_docs]
// ------------------------------------------------------------------------
__device__ __forceinline__ void d_doc_off(const BatchArgs& A, const Workspace& W, int64_t vblock) {
    const int64_t d = vblock * blockDim.x + threadIdx.x;
    if (d > A.n_docs) return;
    const int64_t o = A.offsets[d];
    if (o >= A.n_bytes) {
        A.out_offsets[d] = W.tile_base[A.n_tiles];
        return;
    }
    const int64_t tile = o / TILE_BYTES;
    const int64_t v = W.tile_base[tile] + W.doc_tile_pos[d];
    if (W.tile_nexc[tile]) return;  // a tile with exception words: d_gather_exc writes its documents' offsets
    A.out_offsets[d] = v;
}


// k_finish: everything behind the scan in ONE launch -- the workgroups [0, g_gather) copy tile runs to ids_out,
// [g_gather, g_gather + g_exc) do the same for the tiles that also hold exception words, the rest write out_offsets.
constexpr int FINISH_EXC_BLOCKS = 2048;  // (18 KB of LDS each: eight per CU, 2048 resident)
template <typename RunT>
__global__ __launch_bounds__(64 * GATHER_WAVES) void k_finish(DevTables T, BatchArgs A, Workspace W, uint32_t g_gather) {
    const uint32_t b = blockIdx.x;
    if (b < g_gather) {
        d_gather<RunT>(T, A, W, (int64_t)b * GATHER_WAVES + (threadIdx.x >> 6));
    } else if (b < g_gather + FINISH_EXC_BLOCKS) {
        __shared__ __attribute__((aligned(16))) uint8_t lds_exc[cmax(GATHER_WAVES * ((gather_exc_lds(GATHER_EXC_SMALL) + 15) & ~(size_t)15), gather_exc_lds(GATHER_EXC_LDS))];
        if (W.counters[1] != 0) {  // (uniform: the barrier below is safe)
            const uint32_t wv = threadIdx.x >> 6;
            d_gather_exc<GATHER_EXC_SMALL, -1>(T, A, W, (b - g_gather) * GATHER_WAVES + wv, FINISH_EXC_BLOCKS * GATHER_WAVES,
                                               lds_exc + wv * ((gather_exc_lds(GATHER_EXC_SMALL) + 15) & ~(size_t)15));
            __syncthreads();
            if (wv == 0)  // the tiles my four wavefronts skipped
                for (uint32_t w4 = 0; w4 < (uint32_t)GATHER_WAVES; w4++)
                    d_gather_exc<GATHER_EXC_LDS, 1>(T, A, W, (b - g_gather) * GATHER_WAVES + w4, FINISH_EXC_BLOCKS * GATHER_WAVES, lds_exc);
        }
    } else {
        d_doc_off(A, W, (int64_t)(b - g_gather - FINISH_EXC_BLOCKS));
    }
}

// ------------------------------------------------------------------------
// k_cut: the reference ends a document at a word of more than MAX_WORD_BYTES bytes and says nothing (core.c:402-407 sets
// error_msg, core.c:503 clears it): the document keeps the ids of the words in front of that word.  Here such a word may
// have been encoded as a run of shorter ones (seams), so it is found by what it leaves behind whatever it is made of: at
// least CUT_MIN_RUN tiles in a row without a word start of the reference's own (k_tiles' noreal_bits).  ONE workgroup:
//   A. its first wavefront walks the runs of such tiles; the word that covers a run starts at the last real start of the
//      tile in front of it (tile_lastreal: position and ids before it) and ends at the first real start behind the run
//      (the splitter, without seams).  Over-long: the document is cut at the word's first id.
//   B. all threads close the gaps in ids_out (segments move left, chunk by chunk, reads before writes) and shift
//      out_offsets.
// Returns at once in a batch that has no such run -- every batch of ordinary text.
// ------------------------------------------------------------------------
constexpr uint32_t CUT_MIN_RUN = (uint32_t)((MAX_WORD_BYTES + 1 - TILE_BYTES) / TILE_BYTES);  // tiles a word of MAX_WORD_BYTES + 1 bytes covers whole
constexpr int CUT_THREADS = 1024;
__global__ __launch_bounds__(CUT_THREADS) void k_cut(DevTables T, BatchArgs A, Workspace W) {
    if (W.counters[6] < CUT_MIN_RUN) return;
    __shared__ EndsLds L;
    __shared__ uint32_t s_n_cuts;
    __shared__ int32_t s_vals[CUT_THREADS];
    uint32_t* const cut_doc = W.exc_quad;   // (the exception lists are free by now; a cut needs 272 tiles: they are long enough)
    uint32_t* const cut_keep = W.exc_wave;
    const int tid = threadIdx.x, lane = tid & 63;
    if (tid == 0) s_n_cuts = 0;
    __syncthreads();
    if (tid < 64) {
        const int64_t n_words = (A.n_tiles + 31) / 32;
        auto bit = [&](int64_t t) -> bool { return t < A.n_tiles && ((W.noreal_bits[t >> 5] >> (t & 31)) & 1u); };
        int64_t last_doc = -1;
        uint32_t nc = 0;
        EndsWin cw;
        // A run of CUT_MIN_RUN tiles holds whole words of ones; text that sets a bit here and there (paragraphs of CJK
        // characters) has none, and is done with after this scan.  A run is taken up at its first whole word.
        for (int64_t w0 = 0; w0 < n_words; w0 += 64) {
            const int64_t w = w0 + lane;
            const uint32_t bits = w < n_words ? W.noreal_bits[w] : 0u;
            const uint32_t prev = (w > 0 && w <= n_words) ? W.noreal_bits[w - 1] : 0u;
            const bool head = bits == 0xFFFFFFFFu && prev != 0xFFFFFFFFu;
            for (unsigned long long bal = __ballot(head); bal; bal &= bal - 1) {
                const int l = __builtin_ctzll(bal);
                {   // (every value below is the same in all lanes)
                    const uint32_t pv = (uint32_t)__shfl((int)prev, l, 64);
                    const int64_t t_first = (w0 + l) * 32 - __builtin_clz(~pv);  // the ones at the top of the word in front belong to the run
                    int64_t t_end = (w0 + l) * 32;  // first tile behind the run
                    while (bit(t_end)) {
                        if ((t_end & 31) == 0 && t_end + 32 <= A.n_tiles && W.noreal_bits[t_end >> 5] == 0xFFFFFFFFu) t_end += 32;
                        else t_end++;
                    }
                    if (t_end - t_first < (int64_t)CUT_MIN_RUN || t_first == 0) continue;
                    const int64_t tp = t_first - 1;  // it has a start of the reference's own, and none in its halo: tile_lastreal is set
                    const uint32_t lr = W.tile_lastreal[tp];
                    const int64_t P = tp * TILE_BYTES + (int64_t)(lr & 0xFFFFu);
                    const int64_t d = doc_of(A, W, P, (uint32_t)tp), ds = A.offsets[d], de = A.offsets[d + 1];
                    if (d == last_doc) continue;  // cut already, further in front
                    int64_t E = de;
                    if (t_end * TILE_BYTES < de) {
                        bool unused;
                        E = exc_word_end(T, A, t_end * TILE_BYTES - 1, d, ds, de, L.sb, L.scode, L.docm, lane, &unused, cw, false);
                        if (E > de) E = de;
                    }
                    if (E - P <= MAX_WORD_BYTES) continue;
                    // ids of the document in front of the word: the dense ones k_tiles counted, and those of the tile's
                    // exception words in front of P
                    uint32_t exc_ids = 0;
                    const uint32_t nexc = W.tile_nexc[tp], first = W.tile_exc_first[tp];
                    for (uint32_t e = lane; e < nexc; e += 64) {
                        const uint64_t idx = (uint64_t)first + e;
                        if ((int64_t)idx >= W.cap_exc) break;
                        const ExcRec r = W.exc[idx];
                        if (r.ws < P && r.tok_base >= 0) exc_ids += r.cnt;
                    }
                    for (int o = 32; o; o >>= 1) exc_ids += (uint32_t)__shfl_xor((int)exc_ids, o, 64);
                    const int64_t idpos = W.tile_base[tp] + (int64_t)(lr >> 16) + exc_ids;
                    const int64_t keep = idpos - A.out_offsets[d];
                    if (lane == 0) {
                        cut_doc[nc] = (uint32_t)d;
                        cut_keep[nc] = (uint32_t)keep;
                        if (A.status) A.status[d] = HUTK_DOC_WORD_TOO_LARGE;
                        raise(A.err, HUTK_E_WORD_TOO_LARGE);
                    }
                    nc++;
                    last_doc = d;
                }
            }
        }
        if (lane == 0) s_n_cuts = nc;
    }
    __threadfence();
    __syncthreads();
    const uint32_t nc = s_n_cuts;
    if (nc == 0) return;
    // B. segment c = the ids between the end of cut document c - 1 and the kept end of cut document c: it moves left by
    // what the cuts in front of it removed (segment 0 stays); behind the last cut: the rest of the batch
    int64_t shift = 0, seg_begin = 0, doc_begin = 0;
    const int64_t total = A.out_offsets[A.n_docs];
    for (uint32_t c = 0; c <= nc; c++) {
        int64_t seg_end, doc_end, removed = 0;
        if (c < nc) {
            const int64_t d = cut_doc[c], a = A.out_offsets[d], b = A.out_offsets[d + 1];
            seg_end = a + (int64_t)cut_keep[c];
            removed = b - seg_end;
            doc_end = d + 1;  // documents [doc_begin, doc_end) start inside this segment
            __syncthreads();  // (everybody has read the offsets of document d before they move)
            seg_end = seg_end < a ? a : seg_end;
        } else {
            seg_end = total;
            doc_end = A.n_docs + 1;
        }
        if (shift) {
            for (int64_t base = seg_begin; base < seg_end; base += CUT_THREADS) {
                const int64_t i = base + tid;
                if (i < seg_end) s_vals[tid] = A.ids_out[i];
                __syncthreads();
                if (i < seg_end) A.ids_out[i - shift] = s_vals[tid];
                __syncthreads();
            }
            for (int64_t x = doc_begin + tid; x < doc_end; x += CUT_THREADS) A.out_offsets[x] -= shift;
        }
        __syncthreads();
        seg_begin = seg_end + removed;
        doc_begin = doc_end;
        shift += removed;
    }
}

// k_tail_small: everything behind k_tiles for a batch of a few tiles (a sentence, a handful of documents), where the
// launches themselves are the cost: ONE wavefront runs the exception stages, the scan and the copy-out one after the
// other.  Between stages an agent-scope fence makes the wavefront's own global writes visible to its later loads.
constexpr int SMALL_TILES = 32;
template <typename SymT>
__global__ __launch_bounds__(64) void k_tail_small(DevTables T, BatchArgs A, Workspace W) {
    __shared__ __attribute__((aligned(16))) uint8_t lds[cmax(cmax(cmax(2 * MEDIUM_UNITS * 64 * sizeof(SymT), sizeof(EndsLds)),
                                                                  cmax(2 * 4 * QUAD_UNITS * 4, 2 * EXC_LDS_UNITS * 4)),
                                                             sizeof(SymT) == 2 ? cmax(64 * MEDIUM_ROW * 4, LANE_FAST_LDS) : 0)];
    const int lane = threadIdx.x;
    const bool fast = sizeof(SymT) == 2 && T.rank_is_sym;  // (as in k_exc_a / k_exc_b: the quad list may hold words of any mode then)
    auto stage_done = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
        __syncthreads();
    };
    if (W.counters[0] != 0) {  // exception words
        if (fast) d_exc_lane_fast<1, 64>(T, A, W, 0, 1, lds);
        else d_exc_medium<SymT>(T, A, W, 0, 1, lds);
        stage_done();
        for (uint32_t sub = 0; sub < ENDS_SHARE; sub++) d_exc_ends(T, A, W, sub, ENDS_SHARE, lds);
        stage_done();
        if (fast) {
#if HUTK_LAB_EXC_GROUP
            d_exc_group_fast<2>(T, A, W, 0, 1, lds);
            stage_done();
            d_exc_group_fast<4>(T, A, W, 0, 1, lds);
            stage_done();
            d_exc_group_fast<8>(T, A, W, 0, 1, lds);
            stage_done();
            d_exc_group_fast<16>(T, A, W, 0, 1, lds);
#else
            d_exc_lane_fast<2, 16>(T, A, W, 0, 1, lds);
            stage_done();
            d_exc_lane_fast<4, 8>(T, A, W, 0, 1, lds);
#endif
        } else {
            d_exc_quad(T, A, W, 0, 1, lds);
        }
        stage_done();
        d_exc(T, A, W, 0, 1, lds);
        stage_done();
    }
    if (lane == 0) W.counters[10] = 0;  // (k_pre's sample: zero for the next batch, as k_scan does)
    // exclusive scan of the tiles' id counts (at most SMALL_TILES <= 64)
    {
        uint32_t total;
        const uint32_t mine = lane < A.n_tiles ? W.tile_count[lane] : 0u;
        const uint32_t before = wave_excl_scan(mine, lane, &total);
        if (lane < A.n_tiles) W.tile_base[lane] = before;
        if (lane == 0) W.tile_base[A.n_tiles] = total;
    }
    stage_done();
    for (int64_t w = 0; w * GATHER_TILES < A.n_tiles; w++) d_gather<SymT>(T, A, W, w);
    if (W.counters[1] != 0) {
        __shared__ __attribute__((aligned(16))) uint8_t lds_exc[gather_exc_lds(GATHER_EXC_LDS)];
        d_gather_exc<GATHER_EXC_LDS, 0>(T, A, W, 0, 1, lds_exc);
    }
    stage_done();  // (d_gather_exc writes the exception records' output positions; nothing below reads them, but keep the stages uniform)
    for (int64_t vb = 0; vb * 64 <= A.n_docs; vb++) d_doc_off(A, W, vb);
}

// ------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------
void launch_pre(const BatchArgs& a, const Workspace& w, hipStream_t s) {
    const unsigned g = (unsigned)((a.n_tiles + 255) / 256);
    hipLaunchKernelGGL(k_pre, dim3(g), dim3(256), 0, s, a, w);
}
void launch_tiles(const DevTables& t, const BatchArgs& a, const Workspace& w, hipStream_t s) {
    // TILE_WAVES tiles per workgroup: they pool their merge-loop words and share one copy of the splitter's
    // transition table in LDS (one tile per workgroup outside byte-encoder mode was 17 % slower once the table
    // was there: fewer resident wavefronts)
#define HUTK_LAUNCH(ST, BM, RS, WV)                                                                     \
    do {                                                                                                \
        const dim3 g_((unsigned)(((a.n_tiles + WV - 1) / WV + 7) / 8 * 8)), b_(64 * WV);                \
        if (t.has_multi) hipLaunchKernelGGL((k_tiles<ST, BM, RS, WV, true>), g_, b_, 0, s, t, a, w);   \
        else hipLaunchKernelGGL((k_tiles<ST, BM, RS, WV, false>), g_, b_, 0, s, t, a, w);              \
    } while (0)
    const int variant = (t.sym16 ? 4 : 0) | (t.is_byte_encoder ? 2 : 0) | (t.rank_is_sym ? 1 : 0);
    switch (variant) {
        case 7: HUTK_LAUNCH(uint16_t, true, true, TILE_WAVES); break;
        case 6: HUTK_LAUNCH(uint16_t, true, false, TILE_WAVES); break;
        case 5: HUTK_LAUNCH(uint16_t, false, true, TILE_WAVES); break;
        case 4: HUTK_LAUNCH(uint16_t, false, false, TILE_WAVES); break;
        case 3: HUTK_LAUNCH(uint32_t, true, true, TILE_WAVES); break;
        case 2: HUTK_LAUNCH(uint32_t, true, false, TILE_WAVES); break;
        case 1: HUTK_LAUNCH(uint32_t, false, true, TILE_WAVES); break;
        default: HUTK_LAUNCH(uint32_t, false, false, TILE_WAVES); break;
    }
#undef HUTK_LAUNCH
}
// the whole pipeline of a batch of at most TILE_WAVES tiles in one launch (k_tiles<..., ONE>); false: no such kernel for this vocabulary
bool one_shot_takes(const DevTables& t, const BatchArgs& a) {
    return TILE_WAVES <= 4 && a.n_tiles >= 1 && a.n_tiles <= TILE_WAVES && !t.has_multi && !a.word_bits && !a.first_bits;
}
void launch_one_shot(const DevTables& t, const BatchArgs& a, const Workspace& w, hipStream_t s) {
#define HUTK_LAUNCH1(ST, BM, RS) hipLaunchKernelGGL((k_tiles<ST, BM, RS, TILE_WAVES, false, true>), dim3(1), dim3(64 * TILE_WAVES), 0, s, t, a, w)
    const int variant = (t.sym16 ? 4 : 0) | (t.is_byte_encoder ? 2 : 0) | (t.rank_is_sym ? 1 : 0);
    switch (variant) {
        case 7: HUTK_LAUNCH1(uint16_t, true, true); break;
        case 6: HUTK_LAUNCH1(uint16_t, true, false); break;
        case 5: HUTK_LAUNCH1(uint16_t, false, true); break;
        case 4: HUTK_LAUNCH1(uint16_t, false, false); break;
        case 3: HUTK_LAUNCH1(uint32_t, true, true); break;
        case 2: HUTK_LAUNCH1(uint32_t, true, false); break;
        case 1: HUTK_LAUNCH1(uint32_t, false, true); break;
        default: HUTK_LAUNCH1(uint32_t, false, false); break;
    }
#undef HUTK_LAUNCH1
}
bool small_tail(const BatchArgs& a) { return a.n_tiles <= SMALL_TILES && a.n_docs <= 4096; }
void launch_tail_small(const DevTables& t, const BatchArgs& a, const Workspace& w, hipStream_t s) {
    if (t.sym16) hipLaunchKernelGGL(k_tail_small<uint16_t>, dim3(1), dim3(64), 0, s, t, a, w);
    else hipLaunchKernelGGL(k_tail_small<uint32_t>, dim3(1), dim3(64), 0, s, t, a, w);
}
void launch_exceptions(const DevTables& t, const BatchArgs& a, const Workspace& w, hipStream_t s) {
    // two launches, fixed grids; every wavefront pulls work until its device list runs out
    if (t.sym16) hipLaunchKernelGGL(k_exc_a<uint16_t>, dim3(EXA_MEDIUM16 + EXA_ENDS), dim3(64), 0, s, t, a, w, (uint32_t)EXA_MEDIUM16);
    else hipLaunchKernelGGL(k_exc_a<uint32_t>, dim3(EXA_MEDIUM32 + EXA_ENDS), dim3(64), 0, s, t, a, w, (uint32_t)EXA_MEDIUM32);
    if (t.sym16 && t.rank_is_sym) hipLaunchKernelGGL(k_exc_b<true>, dim3(EXB_FAST_WGS), dim3(64), 0, s, t, a, w);
    else hipLaunchKernelGGL(k_exc_b<false>, dim3(EXB_QUAD / 2 + EXB_WAVE / 2), dim3(128), 0, s, t, a, w);
}
void launch_scan(const BatchArgs& a, const Workspace& w, hipStream_t s) {
    hipLaunchKernelGGL(k_scan, dim3((unsigned)w.n_scan_blocks), dim3(SCAN_THREADS), 0, s, a, w);
}
int64_t scan_blocks(int64_t n_tiles) { return (n_tiles + SCAN_BLOCK - 1) / SCAN_BLOCK; }
void launch_cut(const DevTables& t, const BatchArgs& a, const Workspace& w, hipStream_t s) {
    hipLaunchKernelGGL(k_cut, dim3(1), dim3(CUT_THREADS), 0, s, t, a, w);
}
void launch_finish(const DevTables& t, const BatchArgs& a, const Workspace& w, hipStream_t s) {
    const int64_t per_wg = (int64_t)GATHER_TILES * GATHER_WAVES;
    const unsigned g_gather = (unsigned)((a.n_tiles + per_wg - 1) / per_wg);
    const unsigned g_doc = (unsigned)((a.n_docs + 1 + 64 * GATHER_WAVES - 1) / (64 * GATHER_WAVES));
    const dim3 g(g_gather + FINISH_EXC_BLOCKS + g_doc), b(64 * GATHER_WAVES);
    if (t.sym16) hipLaunchKernelGGL(k_finish<uint16_t>, g, b, 0, s, t, a, w, g_gather);
    else hipLaunchKernelGGL(k_finish<uint32_t>, g, b, 0, s, t, a, w, g_gather);
}
// chunked host path (hutk_api.cpp): document offsets of a chunk made relative to its first byte, and the
// chunk's out_offsets made absolute by the ids of the chunks before it (a device scalar)
__global__ void k_rebase_offsets(const int64_t* in, int64_t* out, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[i] - in[0];
}
__global__ void k_add_base(int64_t* v, int64_t n, const int64_t* base) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) v[i] += *base;
}
__global__ void k_copy_one(int64_t* dst, const int64_t* src) { *dst = *src; }

void launch_rebase_offsets(const int64_t* in, int64_t* out, int64_t n, hipStream_t s) {
    hipLaunchKernelGGL(k_rebase_offsets, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, in, out, n);
}
void launch_add_base(int64_t* v, int64_t n, int64_t* base, hipStream_t s) {
    hipLaunchKernelGGL(k_add_base, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, v, n, base);
    hipLaunchKernelGGL(k_copy_one, dim3(1), dim3(1), 0, s, base, v + (n - 1));
}

void launch_bpe_symbols(const DevTables& t, uint32_t* d_syms, int n, int32_t* d_ids_out,
                        int32_t* d_n_out, hipStream_t s) {
    hipLaunchKernelGGL(k_bpe_symbols, dim3(1), dim3(64), 0, s, t, d_syms, n, d_ids_out, d_n_out);
}

}  // namespace hutk
