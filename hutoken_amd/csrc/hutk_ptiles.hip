// hutk_ptiles.hip -- the PERSISTENT form of the tile kernel (gfx950 / CDNA4, wave64).
//
// k_tiles (hutk_kernels.hip) gives every workgroup four tiles and ends their merge phase with a workgroup barrier: one
// wavefront runs the merge trips of the pooled words while three wait, and a trip's sixty-four lanes are a quarter full
// (DESIGN.md section 5).  Here ONE workgroup of PT_WAVES wavefronts stays on a compute unit for the whole launch and
// there is no workgroup barrier after the tables are staged.  A tile lives in a SLOT of LDS from its front end to its
// epilogue; the three stages are taken by whichever wavefront is free:
//
//   front end   stage the tile's bytes, classify (the reference's splitter, src/parser.c:24-183, as an automaton),
//               one-byte words resolved by the lane that owns the position, the other words round-robin to the lanes:
//               whole-word table probe; the words that need the merge loop go into the workgroup's QUEUE
//   merge       a wavefront takes up to 64 queued words (of any tiles), one lane per word, one merge per trip
//               (src/core.c:66-209, leftmost pair of minimal rank: src/queue.c:152-199); a lane whose word is finished
//               publishes it and takes the next queued word (refill), so the trips stay full; the last word of a tile
//               to finish marks the tile READY
//   epilogue    counts -> DPP scan -> the tile's run of symbols, exception records, ids in front of each document
//
// All hand-offs are LDS atomics (no s_barrier, no global flags): LDS instructions of a wavefront execute in order and
// the LDS serves one instruction at a time, so "write the data, then set the flag" needs no wait, only compiler order.
//
// Byte-encoder mode, 16-bit symbols, rank == symbol order (GPT-2-shaped files, with or without a merges file); every
// other vocabulary shape, the regex pre-token path and small batches stay with k_tiles.  Same outputs as k_tiles: the
// kernels behind it (exception words, scan, finish, cut) do not know which of the two ran.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "hutk_kdev.h"

namespace hutk {

#ifndef HUTK_PT_WAVES
#define HUTK_PT_WAVES 16
#endif
#ifndef HUTK_PT_PROF
#define HUTK_PT_PROF 0
#endif
#ifndef HUTK_PT_PERTURB_VALU
#define HUTK_PT_PERTURB_VALU 0
#endif
#ifndef HUTK_PT_PERTURB_SLEEP
#define HUTK_PT_PERTURB_SLEEP 0
#endif
#ifndef HUTK_PT_MARKS
#define HUTK_PT_MARKS 0
#endif
#ifndef HUTK_PT_SWAR
#define HUTK_PT_SWAR 0
#endif
#ifndef HUTK_PT_REFILL
#define HUTK_PT_REFILL 48
#endif
#if HUTK_PT_MARKS  // (tools/ptiles_isa.py builds with -DHUTK_PT_MARKS=1: comments in the ISA between which it counts instructions)
#define PT_MARK(name) asm volatile("; PTMARK " name)
#else
#define PT_MARK(name) do {} while (0)
#endif
constexpr int PT_WAVES = HUTK_PT_WAVES;       // wavefronts of the one workgroup a compute unit holds
#ifndef HUTK_PT_WGS
#define HUTK_PT_WGS 1  // workgroups per compute unit (2: eight wavefronts per SIMD, 64 VGPRs, half the LDS each)
#endif
#ifndef HUTK_PT_SLOTS
#define HUTK_PT_SLOTS (HUTK_PT_WGS == 1 ? 30 : 13)
#endif
#ifndef HUTK_PT_ARENAS
#define HUTK_PT_ARENAS (HUTK_PT_WGS == 1 ? 5 : 3)
#endif
constexpr int PT_WGS = HUTK_PT_WGS;
constexpr int PT_SLOTS = HUTK_PT_SLOTS;       // tiles in flight per workgroup (one bit each in the control masks)
constexpr int PT_QCAP = HUTK_PT_WGS == 1 ? 2048 : 1024;  // entries of the merge-word queue (a power of two)
constexpr int PT_ARENAS = HUTK_PT_ARENAS;     // wavefronts that can merge at the same time
constexpr int PT_ROW = 34;                    // 16-bit entries of one lane's row of pair results: 32 units + one dword, so that lane l's row begins one bank behind lane l - 1's
constexpr int PT_REFILL_MIN = HUTK_PT_REFILL;  // a merging wavefront takes new words when that many of its lanes are without one
constexpr int PT_STAGE = 256;                 // word starts a front end stages at a time (a tile has ~180 words of 2..14 bytes)
constexpr int PT_ROOM_AHEAD = 64;             // queue entries a front end reserves before it knows its words (a tile has ~11)
constexpr uint32_t PT_Q_VALID = 0x80000000u;
constexpr int PT_NPOS = TILE_BYTES + HALO;    // 1024 classified positions, 16 per lane
static_assert(PT_NPOS == 64 * 16, "16 positions per lane");
static_assert(PT_SLOTS <= 32 && (PT_QCAP & (PT_QCAP - 1)) == 0, "control masks / ring size");

// A tile in flight.  Written by its front end, read by the merging wavefronts (bytes, word starts; they write symbols and
// surviving units) and by its epilogue.
struct PtSlot {
    __attribute__((aligned(16))) uint8_t sb[WINDOW];
    __attribute__((aligned(16))) uint16_t S[PT_NPOS];               // symbol of unit i of the word at ws: S[ws + i]
    __attribute__((aligned(8))) uint16_t wmask16[64 + 8];           // word starts, 16 positions per entry
    __attribute__((aligned(8))) uint32_t livem[PT_NPOS / 32 + 2];   // surviving units
    __attribute__((aligned(8))) uint32_t excm[PT_NPOS / 32 + 2];    // starts of exception words
    int32_t pending;     // merge-loop words of the tile that are not merged yet (may be negative until the front end has added its count)
    uint32_t tile;
    uint32_t cutpos;     // k_cut: 1 + position of the tile's last word start of the reference's own, when none follows in the halo
    uint32_t exc_first;  // index of the tile's first exception record ...
    uint32_t exc_list;   // ... and its place on the list of tiles with exception words
    uint32_t dfirst_lo, dfirst_hi;  // first document that can touch the tile
    uint32_t pad_;
};
// what a wavefront keeps to itself
struct PtWave {
    uint32_t docm[WINDOW / 32 + 3];                               // document starts of the window (front end)
    __attribute__((aligned(8))) uint32_t mergem[PT_NPOS / 32 + 2];  // starts of the words for the queue (front end -> enqueue)
    uint16_t stage[PT_STAGE];  // word starts handed to the lanes (front end); dummy slots (merge set-up); lane prefix (epilogue)
};
struct PtCtl {
    uint32_t next_tile;   // tiles of the workgroup's range handed out
    uint32_t done_tiles;  // ... whose epilogue is finished
    uint32_t free_slots;  // bit s: slot s is free
    uint32_t ready;       // bit s: every merge-loop word of slot s is merged
    uint32_t arena_free;  // bit a: merge arena a is free
    uint32_t q_head;      // queue: entries [q_head, q_tail) are reserved by producers and not yet claimed by a merging wavefront
    uint32_t q_tail;
    int32_t q_room;       // ring entries that no producer has reserved and every consumer has read
};

__global__ __launch_bounds__(64 * PT_WAVES) __attribute__((amdgpu_waves_per_eu(PT_WAVES * PT_WGS / 4, PT_WAVES * PT_WGS / 4))) void k_ptiles(DevTables T, BatchArgs A, Workspace W) {
    typedef uint16_t SymT;
    __shared__ PtSlot slots[PT_SLOTS];
    __shared__ __attribute__((aligned(16))) uint8_t s_dfa[dfa::TABLE_BYTES + 256];  // transition table (seam map in its rows' padding), then byte classes
    __shared__ uint16_t s_item[256];  // input byte -> initial symbol
    __shared__ __attribute__((aligned(16))) uint8_t s_mask[17 * 16];  // entry n: n bytes of ones, then zeros (the key of an n-byte word)
    __shared__ uint32_t s_ring[PT_QCAP];
    __shared__ __attribute__((aligned(16))) uint16_t s_arena[PT_ARENAS][64 * PT_ROW + 2];
    __shared__ PtWave s_wave[PT_WAVES];
    __shared__ PtCtl ctl;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    PtWave& my = s_wave[wv];
    uint16_t* const stage = my.stage;

    // ---- the workgroup's tiles: workgroups are dealt round-robin to the 8 XCDs, each with its own L2; every XCD walks one
    // contiguous eighth of the batch (for speed only: nothing below depends on where a workgroup runs)
    const uint32_t G = gridDim.x;  // a multiple of 8
    const uint32_t rr = (blockIdx.x & 7u) * (G >> 3) + (blockIdx.x >> 3);
    const int64_t per = (A.n_tiles + G - 1) / G;
    int64_t tb = (int64_t)rr * per;
    if (tb > A.n_tiles) tb = A.n_tiles;
    int64_t te = tb + per;
    if (te > A.n_tiles) te = A.n_tiles;
    const uint32_t n_my = (uint32_t)(te - tb);
    if (n_my == 0) return;  // (uniform for the workgroup)
    if (select_skips(W, A.n_tiles)) return;  // (both tile kernels are enqueued and this batch is the other one's)

    // ---- tables into LDS, control words; the one workgroup barrier of the kernel
    constexpr int DFA_CHUNKS = (dfa::TABLE_BYTES + 256) / 16;
    for (int i = tid; i < DFA_CHUNKS; i += 64 * PT_WAVES) reinterpret_cast<uint4*>(s_dfa)[i] = T.split_dfa[i];
    for (int i = tid; i < 256; i += 64 * PT_WAVES) s_item[i] = (uint16_t)T.item_sym[i];
    for (int i = tid; i < 17 * 16; i += 64 * PT_WAVES) s_mask[i] = (i & 15) < (i >> 4) ? 0xFFu : 0u;
    for (int i = tid; i < PT_QCAP; i += 64 * PT_WAVES) s_ring[i] = 0;
    for (int i = tid; i < PT_SLOTS; i += 64 * PT_WAVES) slots[i].pending = 0;
    if (tid == 0) {
        ctl.next_tile = 0;
        ctl.done_tiles = 0;
        ctl.free_slots = PT_SLOTS >= 32 ? 0xFFFFFFFFu : ((1u << PT_SLOTS) - 1u);
        ctl.ready = 0;
        ctl.arena_free = (1u << PT_ARENAS) - 1u;
        ctl.q_head = 0;
        ctl.q_tail = 0;
        ctl.q_room = PT_QCAP;
    }
    __syncthreads();

    // LDS control words: relaxed workgroup-scope accesses (ds_ instructions); the order against the data they guard is
    // program order (wave_sync() keeps the compiler from moving LDS accesses across it)
    auto ld = [](const uint32_t* p) -> uint32_t { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };
    auto uni = [](uint32_t v) -> uint32_t { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); };

    // diagnostic (hutk_debug_profile): cycles of this wavefront by activity -> W.prof[16 * wavefront + k]
    //   0 total  1 front end  2 merge  3 epilogue  4 idle  5 tiles  6 merge calls  7 trips  8 refills  9 words merged  10 enqueue waits
    // (a build switch, -DHUTK_PT_PROF=1: the counters cost registers even when they are off)
    const bool prof_on = HUTK_PT_PROF && W.prof != nullptr && A.n_tiles * 10 >= (int64_t)G * PT_WAVES * 16;
    long long pc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};  // 11..15: front end: stage + documents, classify, one-byte words + lists, short words, long words + tail
    const long long t_start = prof_on ? clock64() : 0;

    // ---- the next tile of this wavefront: claimed, and its bytes requested, one tile ahead
    int64_t pf_tile = -1, pf_dfirst = 0, pf_pre_o = 0, pf_dfirst_v = 0;
    bool pf_whole = false;
    uint4 pre0 = make_uint4(0, 0, 0, 0), pre1 = pre0;
    // claim: the tile, its bytes and the index of its first document are requested when the front end of the tile before
    // it begins; the offsets of its first documents (which need that index) when that front end ends: nothing of it waits
    auto claim_next = [&]() {
        uint32_t k = 0;
        if (lane == 0) k = atomicAdd(&ctl.next_tile, 1u);
        k = uni(k);
        const bool have = k < n_my;
        pf_tile = have ? tb + (int64_t)k : -1;
        const int64_t gw = pf_tile * TILE_BYTES - LOOKBACK;
        pf_whole = have && gw >= 0 && gw + WINDOW <= A.n_bytes;
        static_assert(WINDOW / 16 > 64 && WINDOW / 16 <= 128, "two chunks per lane");
        if (pf_whole) {
            pre0 = *reinterpret_cast<const uint4*>(A.bytes + gw + 16 * lane);
            if (lane < WINDOW / 16 - 64) pre1 = *reinterpret_cast<const uint4*>(A.bytes + gw + 16 * (lane + 64));
        }
        // (a vector load on purpose: a scalar one shares its counter with the LDS, and every LDS wait behind it would wait for it)
        const int64_t* fd = W.tile_first_doc + (have ? pf_tile : 0);
        asm volatile("" : "+v"(fd));
        pf_dfirst_v = *fd;
    };
    auto request_offsets = [&]() {
        pf_dfirst = (int64_t)(((uint64_t)uni((uint32_t)((uint64_t)pf_dfirst_v >> 32)) << 32) | uni((uint32_t)pf_dfirst_v));
        pf_pre_o = 0;
        if (pf_tile >= 0 && pf_dfirst + lane <= A.n_docs) pf_pre_o = A.offsets[pf_dfirst + lane];
    };

    // =====================================================================================================
    // epilogue of the tile in slot s: counts -> scan -> symbols out, exception records, ids before each document start
    // (k_tiles phases 7 and 8; byte-encoder mode has neither arena words nor prefix-alone ids)
    // =====================================================================================================
    auto epilogue = [&](const int s) {
        PT_MARK("epi_start");
        PtSlot& me = slots[s];
        const int64_t tile = (int64_t)me.tile;
        const int64_t t0 = tile * TILE_BYTES;
        const int64_t tile_end = (t0 + TILE_BYTES < A.n_bytes) ? t0 + TILE_BYTES : A.n_bytes;
        const uint32_t* wmask32 = reinterpret_cast<const uint32_t*>(me.wmask16);
        // (the offsets of the documents that start here are used last and requested first)
        const int64_t dfirst = (int64_t)(((uint64_t)uni(me.dfirst_hi) << 32) | uni(me.dfirst_lo));
        int64_t o_first = 0;
        if (dfirst + lane <= A.n_docs) o_first = A.offsets[dfirst + lane];
        uint16_t* lanepref = stage;  // ids before lane l's positions
        const uint32_t live16 = reinterpret_cast<const uint16_t*>(me.livem)[lane];
        const uint32_t exc16 = reinterpret_cast<const uint16_t*>(me.excm)[lane];
        // one scan for two counts: ids (bits 0-10, at most RUN_STRIDE), exception words (11-20)
        const uint32_t mine = (uint32_t)__popc(live16) + ((uint32_t)__popc(exc16) << 11);
        uint32_t total;
        const uint32_t run = wave_excl_scan(mine, lane, &total);
        lanepref[lane] = (uint16_t)(run & 0x7FFu);
        const uint32_t n_dense = total & 0x7FFu, n_exc = (total >> 11) & 0x3FFu;
        static_assert(RUN_STRIDE < 2048 && TILE_BYTES < 1024, "count fields");
        const uint32_t exc_first = me.exc_first;
        // the tile's first word start, for the words of EARLIER tiles that end here (d_exc_ends) -- here, not in the front end:
        // there its three registers were two spills and 12 bytes of scratch, and a kernel with scratch is slow to leave
        const uint32_t wm16 = reinterpret_cast<const uint16_t*>(me.wmask16)[lane];
        const unsigned long long starts = __ballot(wm16 != 0);
        const int l0 = starts ? __builtin_ctzll(starts) : 0;
        const uint32_t f0 = (uint32_t)__builtin_amdgcn_readlane((int)wm16, l0);
        if (lane == 0) {
            W.tile_first_start[tile] = starts ? (uint32_t)(16 * l0 + __builtin_ctz(f0)) : 0xFFFFu;
            if (n_exc) W.exc_tiles[me.exc_list] = (uint32_t)tile;
            W.tile_count[tile] = n_dense;
            W.tile_dense[tile] = n_dense;
            W.tile_run_start[tile] = 0;
            W.tile_exc_first[tile] = exc_first;
            W.tile_nexc[tile] = n_exc;
        }
        SymT* dst = reinterpret_cast<SymT*>(W.run) + tile * RUN_STRIDE;
        {
            uint32_t pos = run & 0x7FFu, eidx = (run >> 11) & 0x3FFu;
            for (uint32_t ev = live16 | exc16; ev; ev &= ev - 1) {
                const int j = __builtin_ctz(ev);
                const int ws = 16 * lane + j;
                if ((exc16 >> j) & 1u) {
                    const uint64_t slot = (uint64_t)exc_first + eidx;
                    if ((int64_t)slot < W.cap_exc) {
                        const uint64_t nxt = bits64(wmask32, ws + 1) & 0x7FFFFFFFFFFFFFFFull;
                        int nb = nxt ? 1 + __builtin_ctzll(nxt) : 64;
                        bool known_end = nxt != 0 && (ws + nb < PT_NPOS || t0 + ws + nb >= A.n_bytes);
                        if (nxt == 0) {
                            // more than 63 bytes: the word may still end inside this tile's 1024 classified positions; then
                            // its length is known here and d_exc_ends has nothing to do for it
                            int e = -1;
                            for (int k = (ws + 64) >> 5; k < PT_NPOS / 32 && e < 0; k++) {
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
                dst[pos++] = me.S[ws];  // (a live bit and an exception bit never share a position)
            }
        }
        wave_sync();
        // ... and before the tile's last start of the reference's own, when none follows in the halo: that is where k_cut
        // cuts a document if the word turns out to be over-long (rare: a word of 64 bytes and more)
        if (const uint32_t cp = me.cutpos; cp != 0) {
            if (lane == 0) {
                const int pos = (int)cp - 1;
                const int lr = pos >> 4;
                const uint32_t below = (1u << (pos & 15)) - 1u;
                uint32_t before = lanepref[lr];
                before += (uint32_t)__popc(reinterpret_cast<const uint16_t*>(me.livem)[lr] & below);
                W.tile_lastreal[tile] = (uint32_t)pos | (before << 16);
            }
        }
        // ids emitted before each document that starts in this tile
        for (int64_t d = dfirst + lane; d <= A.n_docs; d += 64) {
            const int64_t o = d == dfirst + lane ? o_first : A.offsets[d];
            if (o >= tile_end) break;
            if (o < t0) continue;
            const int r = (int)(o - t0);
            const int lr = r >> 4;
            const uint32_t below = (1u << (r & 15)) - 1u;
            uint32_t before = lanepref[lr];
            before += (uint32_t)__popc(reinterpret_cast<const uint16_t*>(me.livem)[lr] & below);
            W.doc_tile_pos[d] = before;
        }
        PT_MARK("epi_end");
        wave_sync();  // (every LDS read of the slot is issued before the slot is given back: the LDS serves them in order)
        if (lane == 0) {
            atomicOr(&ctl.free_slots, 1u << s);
            atomicAdd(&ctl.done_tiles, 1u);
        }
    };

    // =====================================================================================================
    // front end of the prefetched tile in slot s; returns the number of words for the merge queue (their starts are
    // in my.mergem)
    // =====================================================================================================
    // room in the ring for nm entries?  (taken at once when there is)
    auto reserve = [&](const uint32_t nm) -> bool {
        uint32_t ok = 1;
        if (lane == 0) {
            const int old = atomicSub(&ctl.q_room, (int)nm);
            if (old < (int)nm) {
                atomicAdd(&ctl.q_room, (int)nm);
                ok = 0;
            }
        }
        return uni(ok) != 0;
    };
    auto front_end = [&](const int s) -> uint32_t {
        PtSlot& me = slots[s];
        uint8_t* const sb = me.sb;
        uint32_t* const docm = my.docm;
        uint16_t* const wmask16 = me.wmask16;
        uint32_t* const mergem = my.mergem;
        uint32_t* const excm = me.excm;
        uint32_t* const livem = me.livem;
        SymT* const S = me.S;
        const int64_t tile = pf_tile;
        const int64_t t0 = tile * TILE_BYTES;
        const int64_t gw = t0 - LOOKBACK;  // global offset of window index 0
        const uint32_t* wmask32 = reinterpret_cast<const uint32_t*>(wmask16);
        const int64_t tile_end = (t0 + TILE_BYTES < A.n_bytes) ? t0 + TILE_BYTES : A.n_bytes;
        const int64_t dfirst = pf_dfirst;
        const int64_t pre_o = pf_pre_o;  // offsets[dfirst + lane]
        const bool whole = pf_whole;
        // room in the queue for the tile's merge-loop words: asked for now, looked at when they are known
        // The tile's count of unmerged words starts at ONE, taken away when the front end is over: a word that is merged while
        // the front end still runs must not make the tile look finished.
        int room_old = 0;
        if (lane == 0) {
            room_old = atomicSub(&ctl.q_room, PT_ROOM_AHEAD);
            atomicAdd(&me.pending, 1);
        }
        uint32_t pushed = 0;  // words of this tile put into the queue by the rounds themselves
        long long fe_t = HUTK_PT_PROF && prof_on ? clock64() : 0;
#define PT_FE_STAMP(k)                              \
    do {                                            \
        if (HUTK_PT_PROF && prof_on) {              \
            const long long n_ = clock64();         \
            pc[k] += n_ - fe_t;                     \
            fe_t = n_;                              \
        }                                           \
    } while (0)

        PT_MARK("fe_stage");
        // ---- 1. stage bytes -----------------------------------------------------------
        if (whole) {
            *reinterpret_cast<uint4*>(sb + 16 * lane) = pre0;
            if (lane < WINDOW / 16 - 64) *reinterpret_cast<uint4*>(sb + 16 * (lane + 64)) = pre1;
        } else {
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
        }
        if (lane < WINDOW / 32 + 3) docm[lane] = 0;
        if (lane < PT_NPOS / 32 + 2) { mergem[lane] = 0; excm[lane] = 0; livem[lane] = 0; }
        if (lane < 8) wmask16[64 + lane] = 0xFFFFu;
        if (lane == 0) {
            me.cutpos = 0;
            me.tile = (uint32_t)tile;
            me.exc_first = 0;
            me.exc_list = 0;
            me.dfirst_lo = (uint32_t)dfirst;
            me.dfirst_hi = (uint32_t)((uint64_t)dfirst >> 32);
        }
        wave_sync();

        PT_MARK("fe_docs");
        // ---- 2. document starts inside the window --------------------------------------
        for (int64_t d = dfirst + lane; d <= A.n_docs; d += 64) {
            const int64_t o = d == dfirst + lane ? pre_o : A.offsets[d];
            if (o >= gw + WINDOW) break;
            const int li = (int)(o - gw);
            if (li >= 0) atomicOr(&docm[li >> 5], 1u << (li & 31));
        }
        // The next tile's bytes are requested HERE: pre0 / pre1 are stored, the offsets above are the last thing this front
        // end waits for before the table loads of its rounds, and a wait for vector memory waits for everything requested.
        claim_next();
        wave_sync();

        PT_FE_STAMP(11);
        PT_MARK("fe_classify");
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
#if HUTK_PT_SWAR
            flags = classify16(dw, dbits, &exotic);  // byte-parallel mask algebra: ~650 integer instructions, NO LDS lookup
#else
            flags = classify16_dfa2(dw, dbits, reinterpret_cast<const uint16_t*>(s_dfa), s_dfa + dfa::TABLE_BYTES, &exotic);  // (two walks side by side: a shorter chain)
#endif
            if (exotic) {
                // overlong encodings: per-position decode.  Its window lives in LDS meanwhile (the slot's symbols are not written
                // yet): the decode indexes it dynamically, which in registers means scratch memory -- and a launch whose scratch
                // size differs from its predecessor's makes the queue drain first (~40 us when this kernel only looks and returns).
                uint32_t* const tmp = reinterpret_cast<uint32_t*>(S) + 8 * lane;
#pragma unroll
                for (int i = 0; i < 8; i++) tmp[i] = dw[i];
                flags = classify16_exact(*reinterpret_cast<const uint32_t(*)[8]>(tmp), dbits);
            }
            // (k_cut, below: the lanes with a word start of the reference's own, and the starts of the last such lane of the tile)
            with_start = __ballot(flags != 0);
            last_real16 = (uint32_t)__builtin_amdgcn_readlane((int)flags, 63 - __builtin_clzll((with_start & 0x0FFFFFFFFFFFFFFFull) | 1ull));
            if (T.seam_on) {
                // Seams (hutk_internal.h, Tables::seam_hi): where no merge can join the input byte x to the lead byte y of the
                // three- or four-byte character behind it, y starts a word of its own
                const uint64_t K8 = 0x8080808080808080ull;
                uint64_t m0 = w.b & (w.b << 1) & (w.b << 2) & K8;  // bytes >= 0xE0 among my positions 0..7
                uint64_t m1 = w.c & (w.c << 1) & (w.c << 2) & K8;  // ... 8..15
                uint32_t ask2 = 0;  // my positions the map leaves alone ("may join"): the second level's, below
                if (m0 | m1) {
                    const uint64_t p0 = (w.b << 8) | (w.a >> 56), p1 = (w.c << 8) | (w.b >> 56);  // the byte in front of each
                    for (; m0; m0 &= m0 - 1) {
                        const int sh = __builtin_ctzll(m0) - 7;
                        const uint32_t y = (uint32_t)(w.b >> sh) & 0xFFu, x = (uint32_t)(p0 >> sh) & 0xFFu;
                        const uint32_t sm = *reinterpret_cast<const uint32_t*>(s_dfa + dfa::seam_offset(x));
                        if (!((sm >> (y & 31u)) & 1u)) flags |= 1u << (sh >> 3);
                        else ask2 |= 1u << (sh >> 3);
                    }
                    for (; m1; m1 &= m1 - 1) {
                        const int sh = __builtin_ctzll(m1) - 7;
                        const uint32_t y = (uint32_t)(w.c >> sh) & 0xFFu, x = (uint32_t)(p1 >> sh) & 0xFFu;
                        const uint32_t sm = *reinterpret_cast<const uint32_t*>(s_dfa + dfa::seam_offset(x));
                        if (!((sm >> (y & 31u)) & 1u)) flags |= 1u << (8 + (sh >> 3));
                        else ask2 |= 1u << (8 + (sh >> 3));
                    }
                }
                // Second level (Tables::seam2_*): whole three-byte characters A | B on both sides of such a boundary -- a lane has
                // at most six lead bytes among its sixteen positions; their two loads each are all in flight before any is used
                if (T.seam2_on && __any(ask2 != 0)) {
                    for (uint32_t nn = ask2; __any(nn != 0);) {  // three boundaries at a time (registers), at most two rounds
                        uint32_t a3v[3], b3v[3], hit[3];  // hit: bit 0 = some entry may join here; bits 8.. = my position
#pragma unroll
                        for (int k = 0; k < 3; k++) {
                            a3v[k] = b3v[k] = 0;
                            hit[k] = 1u;  // ("may join": nothing to ask)
                            if (nn) {
                                const int j = __builtin_ctz(nn);
                                nn &= nn - 1;
                                const int oa = j + 5, ob = j + 8;  // byte offsets of A and B in the 32-byte window
                                const int qa = oa >> 3, qb = ob >> 3;
                                const uint32_t a3 = win24(qa == 0 ? w.a : qa == 1 ? w.b : w.c, qa == 0 ? w.b : qa == 1 ? w.c : w.d, 8 * (oa & 7));
                                const uint32_t b3 = win24(qb == 1 ? w.b : w.c, qb == 1 ? w.c : w.d, 8 * (ob & 7));
                                if (seam2_char3(a3) && seam2_char3(b3)) {
                                    a3v[k] = a3;
                                    b3v[k] = b3;
                                    hit[k] = ((T.seam2_part[(a3 >> 16) & 0xFFu] >> (b3 & 31u)) & 1u) | ((uint32_t)j << 8);
                                }
                            }
                        }
                        // the kinds of keys this vocabulary's set holds (hutk_seam2.h), one after the other, a kind's three lookups together
                        auto ask = [&](uint32_t ka, uint32_t kb) {
                            if (!(T.seam2_cats & seam2_cat_bit(ka, kb))) return;  // (the same for every lane)
                            uint32_t wd[3], hb[3];
#pragma unroll
                            for (int k = 0; k < 3; k++) {
                                const uint32_t a = ka == 3 ? a3v[k] : (a3v[k] >> 16) & 0xFFu;
                                const uint32_t bb = kb == 3 ? b3v[k] : kb == 2 ? (b3v[k] & 0xFFFFu) : (b3v[k] & 0xFFu);
                                const uint32_t h = seam2_hash(a, bb, ka, kb) >> T.seam2_shift;
                                wd[k] = T.seam2_bits[h >> 5];
                                hb[k] = h & 31u;
                            }
#pragma unroll
                            for (int k = 0; k < 3; k++) hit[k] |= (wd[k] >> hb[k]) & 1u;
                        };
                        ask(3, 3); ask(3, 2); ask(3, 1); ask(1, 3); ask(1, 2);
#pragma unroll
                        for (int k = 0; k < 3; k++)
                            if (!(hit[k] & 1u)) flags |= 1u << (hit[k] >> 8);
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
        wmask16[lane] = (uint16_t)flags;
        wave_sync();

#if HUTK_PT_PERTURB_VALU
        {   // MEASUREMENT ONLY: extra VALU instructions (a dependent chain, every lane): is the kernel bound by VALU issue?
            uint32_t x = flags | 1u;
            for (int i = 0; i < HUTK_PT_PERTURB_VALU / 4; i++) {
                x ^= x << 13; x ^= x >> 17;
                asm volatile("" : "+v"(x));
            }
            if (x == 0x9E3779B9u) raise(A.err, HUTK_E_MEMORY);
        }
#endif
#if HUTK_PT_PERTURB_SLEEP
        for (int i = 0; i < HUTK_PT_PERTURB_SLEEP; i++) __builtin_amdgcn_s_sleep(127);  // MEASUREMENT ONLY: ~8 k idle cycles each
#endif
        PT_FE_STAMP(12);
        PT_MARK("fe_words");
        // ---- 4. words ------------------------------------------------------------------
        const int limit = (int)(tile_end - t0);  // words are starts at tile offsets < limit
        uint32_t own = flags;                    // starts that are words of this tile
        {
            const int lo = 16 * lane;
            if (lo >= limit) own = 0;
            else if (lo + 16 > limit) own &= (1u << (limit - lo)) - 1u;
        }
        {
            // k_cut's notes (see k_tiles): tiles without a word start of the reference's own; both cases are rare
            const unsigned long long minel = with_start & ((1ull << ((limit + 15) >> 4)) - 1ull);
            if (minel == 0 || (with_start >> 60) == 0) {
                if (lane == 0) {
                    if (minel == 0) {
                        atomicOr(&W.noreal_bits[tile >> 5], 1u << (tile & 31));
                        atomicAdd(&W.counters[6], 1u);
                    } else {  // last_real16: the starts (without seams) of the tile's last lane that has one
                        me.cutpos = 1u + (uint32_t)(16 * (63 - __builtin_clzll(minel)) + 31 - __builtin_clz(last_real16 & 0xFFFFu));
                    }
                }
            }
        }
        reinterpret_cast<uint16_t*>(livem)[lane] = (uint16_t)own;  // the first unit of a word always survives
        // 4a. The starts of my 16 positions by the length of their word, all sixteen at once: f32 = my starts and the next
        // lane's; a word is ONE byte when the next position is a start too, SHORT (2..14 bytes: the key of the whole-word
        // table) when a start follows within 14 positions, else LONG.  With a prefix the first word of a document is an
        // exception word whatever its length (core.c:364-366, 421-451): the long words' pass decides that.
        const uint32_t f32 = flags | ((uint32_t)wmask16[lane + 1] << 16);
        uint32_t x14 = f32 >> 1;
        x14 |= x14 >> 1;
        x14 |= x14 >> 2;
        x14 |= x14 >> 4;
        x14 |= x14 >> 6;  // bit j: a start among positions j + 1 .. j + 14
        uint32_t singles = own & (f32 >> 1);
        uint32_t shorts = own & ~singles & x14;
        uint32_t longs = own & ~x14;
        if (T.has_prefix) {
            const uint32_t pdoc = own & (uint32_t)bits64(docm, 16 * lane + LOOKBACK);
            singles &= ~pdoc;
            shorts &= ~pdoc;
            longs |= pdoc;
        }
        // One-byte words need no table: the lane that owns the position has the byte in its window.
        for (uint32_t m = singles; m; m &= m - 1) {
            const int j = __builtin_ctz(m);
            const uint64_t v = j < 8 ? w.b : w.c;
            S[16 * lane + j] = s_item[(uint32_t)(v >> (8 * (j & 7))) & 0xFFu];
        }
        // 4b. Short words, spread evenly over the lanes: word j of them goes to lane j % 64.  The owning lanes put the
        // starts of up to PT_STAGE words into the staging buffer, then the lanes take them 64 at a time.  A round is one
        // memory round trip: both candidate slots of the whole-word table (raw bytes, zero padded to 16 -> symbol of the
        // single token the word encodes to; entries verified by the pipeline itself at context creation) are loaded
        // together.  The word's bytes and the bits behind its start come out of LDS with one unaligned read each.
        //   livem  units that survive: starts as "every word start"; the merge loop adds the other survivors of its words
        //          (unit i of the word at ws is bit ws + i, and S[ws + i] its symbol); exception words are taken out
        //   excm   starts of exception words
        uint32_t nSL;
        const uint32_t sl_base = wave_excl_scan((uint32_t)__popc(shorts) | ((uint32_t)__popc(longs) << 16), lane, &nSL);
        const uint32_t nS = nSL & 0xFFFFu, nL = nSL >> 16;
        PT_FE_STAMP(13);
        PT_MARK("fe_short");
        {
            uint32_t widx = sl_base & 0xFFFFu, rest = shorts;
            for (uint32_t c0 = 0; c0 < nS; c0 += PT_STAGE) {
                while (rest && widx - c0 < (uint32_t)PT_STAGE) {  // (each start is visited once, in the chunk it belongs to)
                    stage[widx - c0] = (uint16_t)(16 * lane + __builtin_ctz(rest));
                    rest &= rest - 1;
                    widx++;
                }
                wave_sync();
                const uint32_t c_end = nS - c0 < (uint32_t)PT_STAGE ? nS - c0 : (uint32_t)PT_STAGE;
                // the rounds of a chunk do not depend on one another: the table loads of all of them are in flight together
                constexpr int NR = PT_STAGE / 64;
                uint32_t wsv[NR];
                uint4 kv[NR], s1v[NR], s2v[NR];
#pragma unroll
                for (int r = 0; r < NR; r++) {
                    wsv[r] = 0xFFFFFFFFu;
                    if (64u * r + lane < c_end) {
                        const uint32_t ws = stage[64 * r + lane];
                        uint32_t wb;  // the start bits behind ws (at least 24 of them: the next start is within 14)
                        __builtin_memcpy(&wb, reinterpret_cast<const uint8_t*>(wmask16) + ((ws + 1u) >> 3), 4);
                        const uint32_t nb = 1u + (uint32_t)__builtin_ctz(wb >> ((ws + 1u) & 7u));
                        uint4 key;
                        __builtin_memcpy(&key, sb + LOOKBACK + ws, 16);
                        const uint4 msk = *reinterpret_cast<const uint4*>(s_mask + 16 * nb);
                        kv[r] = make_uint4(key.x & msk.x, key.y & msk.y, key.z & msk.z, key.w & msk.w);
                        wsv[r] = ws;
                        if (T.word_mask) {  // (uniform)
                            const uint32_t wh = word_hash(kv[r].x, kv[r].y, kv[r].z, kv[r].w);
                            s1v[r] = reinterpret_cast<const uint4*>(T.word_tab)[wh & T.word_mask];
                            s2v[r] = reinterpret_cast<const uint4*>(T.word_tab)[word_slot2(wh, T.word_mask)];
                        }
                    }
                }
                uint32_t missv = 0;  // bit r: my word of round r needs the merge loop
#pragma unroll
                for (int r = 0; r < NR; r++) {
                    if (wsv[r] != 0xFFFFFFFFu) {
                        const uint32_t ws = wsv[r];
                        const uint4 k = kv[r], s1 = s1v[r], s2 = s2v[r];
                        // bitwise on purpose: with && the compiler fetches one word first and the rest only on a match
                        // (the word's bytes beyond 14 are zero: k.w has nothing in the symbol's place; an empty slot is all zero,
                        // a key's first bytes never are)
                        // (without a table -- HUTK_NO_WORD_TABLE -- nothing was loaded into s1 / s2: every word goes on)
                        const bool hit1 = T.word_mask != 0 && ((s1.x ^ k.x) | (s1.y ^ k.y) | (s1.z ^ k.z) | ((s1.w ^ k.w) << 16)) == 0;
                        const bool hit2 = T.word_mask != 0 && ((s2.x ^ k.x) | (s2.y ^ k.y) | (s2.z ^ k.z) | ((s2.w ^ k.w) << 16)) == 0;
                        if (hit1 || hit2) S[ws] = (SymT)((hit1 ? s1.w : s2.w) >> 16);
                        else missv |= 1u << r;
                    }
                }
                // The words that need the merge loop go into the queue at once (one LDS atomic per chunk), into the room this
                // front end asked for when it began; without that room they wait in mergem for the end of the front end.
                if (__any(missv != 0)) {  // (uniform)
                    uint32_t n_push;
                    uint32_t off = wave_excl_scan((uint32_t)__popc(missv), lane, &n_push);
                    const uint32_t have = (int)uni((uint32_t)room_old) >= PT_ROOM_AHEAD ? (uint32_t)PT_ROOM_AHEAD : 0u;
                    if (pushed + n_push <= have) {
                        uint32_t base = 0;
                        if (lane == 0) {
                            atomicAdd(&me.pending, (int)n_push);  // (before the entries can be seen: LDS order)
                            base = atomicAdd(&ctl.q_tail, n_push);
                        }
                        off += uni(base);
#pragma unroll
                        for (int r = 0; r < NR; r++)
                            if ((missv >> r) & 1u) s_ring[off++ & (uint32_t)(PT_QCAP - 1)] = PT_Q_VALID | ((uint32_t)s << 10) | wsv[r];
                        pushed += n_push;
                    } else {
#pragma unroll
                        for (int r = 0; r < NR; r++)
                            if ((missv >> r) & 1u) atomicOr(&mergem[wsv[r] >> 5], 1u << (wsv[r] & 31));
                    }
                }
                wave_sync();
            }
        }
        PT_FE_STAMP(14);
        PT_MARK("fe_long");
        // 4c. Long words (rare in ordinary text): 15..28 bytes have the companion table (28 key bytes and the symbol in two
        // consecutive 16-byte slots behind the main table); up to 32 bytes a lane merges; the others, those whose end is
        // not among the classified positions, and -- with a prefix -- the first words of documents are exception words.
        if (nL) {  // (uniform)
            uint32_t widx = sl_base >> 16, rest = longs;
            for (uint32_t c0 = 0; c0 < nL; c0 += PT_STAGE) {
                while (rest && widx - c0 < (uint32_t)PT_STAGE) {
                    stage[widx - c0] = (uint16_t)(16 * lane + __builtin_ctz(rest));
                    rest &= rest - 1;
                    widx++;
                }
                wave_sync();
                const uint32_t c_end = nL - c0 < (uint32_t)PT_STAGE ? nL - c0 : (uint32_t)PT_STAGE;
                for (uint32_t r0 = 0; r0 < c_end; r0 += 64) {
                    if (r0 + lane < c_end) {
                        const int ws = stage[r0 + lane];
                        // end of the word: the next start bit within 63 positions
                        const uint64_t nxt = bits64(wmask32, ws + 1) & 0x7FFFFFFFFFFFFFFFull;
                        const int nb = nxt ? 1 + __builtin_ctzll(nxt) : 64;
                        const bool known_end = nxt != 0 && (ws + nb < PT_NPOS || t0 + ws + nb >= A.n_bytes);
                        bool exc = !known_end || nb > LANE_MAX_UNITS;  // (a unit per byte: more than 32 bytes is more than a lane merges)
                        if (T.has_prefix) exc = exc || bit_at(docm, ws + LOOKBACK);
                        bool done = false;
                        if (!exc && T.wordl_mask && nb > WORD_KEY_BYTES_16 && nb <= WORDL_KEY_BYTES) {
                            uint4 ka, kb4;
                            __builtin_memcpy(&ka, sb + LOOKBACK + ws, 16);
                            __builtin_memcpy(&kb4, sb + LOOKBACK + ws + 16, 16);
                            const int nb2 = nb - 16;  // -1 .. 12 bytes beyond the first sixteen
                            const uint4 ma = *reinterpret_cast<const uint4*>(s_mask + 16 * (nb < 16 ? nb : 16));
                            const uint4 mb = *reinterpret_cast<const uint4*>(s_mask + 16 * (nb2 > 0 ? nb2 : 0));
                            const uint32_t k0 = ka.x & ma.x, k1 = ka.y & ma.y, k2 = ka.z & ma.z, k3 = ka.w & ma.w;
                            const uint32_t k4 = kb4.x & mb.x, k5 = kb4.y & mb.y, k6 = kb4.z & mb.z;
                            const uint32_t o1 = T.wordl_off + 2u * (word_hash_long(k0, k1, k2, k3, k4, k5, k6) & T.wordl_mask);
                            const uint4 s1 = reinterpret_cast<const uint4*>(T.word_tab)[o1];
                            const uint4 s2 = reinterpret_cast<const uint4*>(T.word_tab)[o1 + 1u];
                            done = ((s1.x ^ k0) | (s1.y ^ k1) | (s1.z ^ k2) | (s1.w ^ k3) | (s2.x ^ k4) | (s2.y ^ k5) | (s2.z ^ k6)) == 0;
                            if (done) S[ws] = (SymT)s2.w;
                        }
                        if (exc) {
                            atomicOr(&excm[ws >> 5], 1u << (ws & 31));
                            atomicAnd(&livem[ws >> 5], ~(1u << (ws & 31)));
                        } else if (!done) {
                            atomicOr(&mergem[ws >> 5], 1u << (ws & 31));  // needs the merge loop
                        }
                    }
                }
                wave_sync();
            }
        }
        PT_MARK("fe_tail");
        {   // exception words of my tile (excm is final here): their records, and the tile's place on the list
            const uint32_t e16 = reinterpret_cast<const uint16_t*>(excm)[lane];
            if (__any(e16 != 0)) {  // (rare)
                uint32_t tot;
                (void)wave_excl_scan((uint32_t)__popc(e16), lane, &tot);
                if (lane == 0) {
                    const unsigned long long old = atomicAdd(reinterpret_cast<unsigned long long*>(W.counters), (1ull << 32) | (unsigned long long)tot);
                    me.exc_first = (uint32_t)old;
                    me.exc_list = (uint32_t)(old >> 32);
                }
            }
        }
        request_offsets();
        // the room asked for at the start: what the rounds did not use goes back
        if (lane == 0) {
            const int have = room_old >= PT_ROOM_AHEAD ? PT_ROOM_AHEAD : 0;
            const int back = have == 0 ? PT_ROOM_AHEAD : have - (int)pushed;  // (it was not there: undo)
            if (back) atomicAdd(&ctl.q_room, back);
        }
        // the words the rounds left in mergem (long words, a tile with more words than the room asked for): the caller enqueues them
        uint32_t nm = 0;
        {
            const uint32_t m16 = reinterpret_cast<const uint16_t*>(mergem)[lane];
            if (__any(m16 != 0)) (void)wave_excl_scan((uint32_t)__popc(m16), lane, &nm);
        }
        wave_sync();
        PT_FE_STAMP(15);
        PT_MARK("fe_end");
#undef PT_FE_STAMP
        return nm;
    };

    // the nm (> 0) words of my.mergem (tile in slot s) into the reserved entries.  The tile's count of unmerged words goes
    // up BEFORE the entries become visible (LDS order), so the word that brings it back to zero is the tile's last.
    auto enqueue = [&](const int s, const uint32_t nm) {
        PT_MARK("enq");
        PtSlot& me = slots[s];
        uint32_t m16 = reinterpret_cast<const uint16_t*>(my.mergem)[lane];
        uint32_t tot;
        uint32_t off = wave_excl_scan((uint32_t)__popc(m16), lane, &tot);
        uint32_t base = 0;
        if (lane == 0) {
            atomicAdd(&me.pending, (int)nm);
            base = atomicAdd(&ctl.q_tail, nm);
        }
        base = uni(base);
        for (; m16; m16 &= m16 - 1) {
            s_ring[(base + off) & (uint32_t)(PT_QCAP - 1)] = PT_Q_VALID | ((uint32_t)s << 10) | (uint32_t)(16 * lane + __builtin_ctz(m16));
            off++;
        }
        wave_sync();
    };

    // the front end of the tile in slot s is over and all its words are in the queue: its own count goes; true when no word
    // is left unmerged (the tile's epilogue is this wavefront's, then)
    auto fe_over = [&](const int s) -> bool {
        uint32_t last = 0;
        if (lane == 0) last = atomicSub(&slots[s].pending, 1) == 1 ? 1u : 0u;
        return uni(last) != 0;
    };

    // =====================================================================================================
    // merge: one lane per word, one merge per trip; lanes without a word take the next queued ones
    // =====================================================================================================
    auto merge = [&](const int arena) {
        constexpr uint32_t NOKEY = 0xFFFFFFFFu;
        SymT* const Mw = s_arena[arena] + 2 + lane * PT_ROW;  // pair results of units (i, next live) of my word at Mw[i]
        const uint32_t mw_lds = lds_addr(Mw);
        SymT* const dummy = reinterpret_cast<SymT*>(stage) + lane;
        const uint32_t* bp = reinterpret_cast<const uint32_t*>(T.bytepair);
        bool holding = false;  // my lane holds a word that is not published yet
        uint32_t live = 0, cand = 0, best = NOKEY, again = 0;
        int p = 0, ws = 0, slot = 0;
        SymT* Sw = dummy;
        // A pair is the 32-bit KEY merged symbol << 5 | position: the smallest key is the pair of minimal rank, leftmost on
        // ties (queue.c:162-164), so the best pair is one register and every comparison a v_min.
        auto scan_key = [&](uint32_t c) -> uint32_t {  // four candidates per step, their LDS reads in flight together
            uint32_t b = NOKEY;
            while (c) {
                // fewer than four left: the index of "no bit" is -1, whose key is all ones whatever the entry in front of the row holds
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
        for (;;) {
            PT_MARK("mg_top");
            const bool active = holding && (best != NOKEY || again != 0);
            const unsigned long long am = __ballot(active);
            const int n_active = __popcll(am);
            if (n_active <= 64 - PT_REFILL_MIN) {
                if (holding && !active) {
                    // finished: the surviving units (unit 0 is in livem already), then the tile's count of unmerged words;
                    // the last word of a tile hands the tile to its epilogue
                    PtSlot& X = slots[slot];
                    const uint64_t lm = (uint64_t)(live & ~1u) << (ws & 31);
                    if ((uint32_t)lm) atomicOr(&X.livem[ws >> 5], (uint32_t)lm);
                    if ((uint32_t)(lm >> 32)) atomicOr(&X.livem[(ws >> 5) + 1], (uint32_t)(lm >> 32));
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    if (atomicSub(&X.pending, 1) == 1) atomicOr(&ctl.ready, 1u << slot);
                    holding = false;
                }
                // up to 64 - n_active queued words
                uint32_t h = 0, c = 0;
                if (lane == 0) {
                    const uint32_t t = ld(&ctl.q_tail);
                    h = ld(&ctl.q_head);
                    const uint32_t avail = t - h;
                    if ((int32_t)avail > 0) {
                        c = avail < (uint32_t)(64 - n_active) ? avail : (uint32_t)(64 - n_active);
                        if (atomicCAS(&ctl.q_head, h, h + c) != h) c = 0;  // (another wavefront was faster: next trip)
                    }
                }
                h = uni(h);
                c = uni(c);
                if (c == 0) {
                    if (n_active == 0) break;
                } else {
                    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(~am >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)~am, 0u));  // my index among the lanes without a word
                    if (!active && rank < c) {
                        const uint32_t qi = (h + rank) & (uint32_t)(PT_QCAP - 1);
                        uint32_t e = 0;
                        for (uint32_t spins = 0;; spins++) {  // (the producer writes the entry right after it reserved it)
                            e = ld(&s_ring[qi]);
                            if (e & PT_Q_VALID) break;
                            if (spins > (1u << 22)) { raise(A.err, HUTK_E_DEVICE); break; }
                        }
                        s_ring[qi] = 0;
                        if (e & PT_Q_VALID) {
                            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                            slot = (int)((e >> 10) & 31u);
                            ws = (int)(e & 1023u);
                            PtSlot& X = slots[slot];
                            const int n = 1 + __builtin_ctzll(bits64(reinterpret_cast<const uint32_t*>(X.wmask16), ws + 1));  // 2..32 units
                            Sw = X.S + ws;
                            live = (n >= 32) ? 0xFFFFFFFFu : ((1u << n) - 1u);
                            cand = 0;
                            // Set-up, eight units per step and no branch per unit: the word's bytes come out of LDS as three
                            // aligned dwords; two consecutive bytes are the index of the (byte, next byte) table, whose entry is
                            // {symbol of the byte, merged symbol of the pair}.  Units beyond the word are looked up all the
                            // same and stored to a dummy slot.
                            const int li = ws + LOOKBACK;
                            for (int i0 = 0; i0 < n; i0 += 8) {
                                const int a = (li + i0) & ~3, o8 = 8 * ((li + i0) & 3);
                                const uint32_t* sw = reinterpret_cast<const uint32_t*>(X.sb + a);
                                const uint32_t q0 = sw[0], q1 = sw[1], q2 = sw[2];
                                const uint64_t lo = (uint64_t)funnel_r(q1, q0, o8) | ((uint64_t)funnel_r(q2, q1, o8) << 32);
                                const uint32_t k2 = q2 >> o8;  // its low byte is byte 8 of the stretch
                                uint32_t ev[8];
#pragma unroll
                                for (int j = 0; j < 8; j++) {
                                    const uint32_t idx = j < 7 ? (uint32_t)(lo >> (8 * j)) & 0xFFFFu
                                                               : ((uint32_t)(lo >> 56) | ((k2 & 0xFFu) << 8));
                                    ev[j] = bp[idx];
                                }
                                uint32_t cb = 0;
#pragma unroll
                                for (int j = 0; j < 8; j++) {
                                    const bool in = i0 + j < n;
                                    SymT* ds = in ? Sw + i0 + j : dummy;
                                    SymT* dm = in ? Mw + i0 + j : dummy;
                                    *ds = (SymT)ev[j];
                                    *dm = (SymT)(ev[j] >> 16);
                                    cb |= (ev[j] < 0xFFFF0000u ? 1u : 0u) << j;
                                }
                                cand |= cb << i0;
                            }
                            cand &= (1u << (n - 1)) - 1u;  // the last unit has no next one (n >= 2)
                            best = scan_key(cand);
                            again = 0;
                            holding = true;
                        }
                    }
                    if (HUTK_PT_PROF) { pc[8]++; pc[9] += c; }
                    if (lane == 0) atomicAdd(&ctl.q_room, (int)c);  // (the entries are read: their places are free again)
                }
            }
            // One merge per trip: apply the best pair, ISSUE the lookups of the two new neighbour pairs, rescan the untouched
            // candidates while those loads fly, then fold the two new keys in.  A lookup that must go on in the pair's
            // SECOND bucket (a filter bit of the first one says so; under 1 % of the lookups) is not followed up inside the
            // trip: the lane remembers which of its two lookups it was (`again`) and REPEATS both in the next trip, from the
            // buckets they need, beside the other lanes' ordinary ones.
            PT_MARK("mg_trip");
            if (HUTK_PT_PROF) pc[7]++;
            if (holding && (best != NOKEY || again != 0)) {
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
                if (again == 0) {  // (a repeating lane's rescan is done)
                    cand &= ~(1u << p0);
                    best = scan_key(cand);
                }
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
        }
        wave_sync();
    };

    // =====================================================================================================
    // the wavefront's loop: whatever is there to do, most urgent first
    // =====================================================================================================
#define PT_ACC(k, expr)                                  \
    do {                                                 \
        if (prof_on) {                                   \
            const long long t0_ = clock64();             \
            expr;                                        \
            pc[k] += clock64() - t0_;                    \
        } else {                                         \
            expr;                                        \
        }                                                \
    } while (0)
    claim_next();
    request_offsets();
    int pend_slot = -1;  // a tile whose front end is done but whose words found no room in the queue yet
    uint32_t pend_nm = 0;
    int epi_slot = -1;   // a tile whose epilogue this wavefront runs next
    uint32_t idle = 0;
    for (;;) {
        if (epi_slot >= 0) {
            PT_ACC(3, epilogue(epi_slot));
            epi_slot = -1;
            idle = 0;
            continue;
        }
        PT_MARK("loop_top");
        // the control words, all requested together: one LDS round trip per turn of the loop
        const uint32_t c_ready = ld(&ctl.ready), c_free = ld(&ctl.free_slots), c_tail = ld(&ctl.q_tail), c_head = ld(&ctl.q_head),
                       c_arena = ld(&ctl.arena_free), c_done = ld(&ctl.done_tiles);
        const uint32_t rd = uni(c_ready), free_now = uni(c_free), avail = uni(c_tail) - uni(c_head), af = uni(c_arena);
        bool can_fe = pf_tile >= 0 && free_now != 0;
        // (a) words waiting for room in the queue
        if (pend_slot >= 0) {
            if (reserve(pend_nm)) {
                enqueue(pend_slot, pend_nm);
                if (fe_over(pend_slot)) epi_slot = pend_slot;
                pend_slot = -1;
                idle = 0;
                continue;
            }
            if (HUTK_PT_PROF) pc[10]++;
            can_fe = false;
        }
        // (b) a tile whose words are all merged: its epilogue, which gives the slot back
        if (rd) {
            uint32_t got = 0;
            const int s = __builtin_ctz(rd);
            if (lane == 0) got = (atomicAnd(&ctl.ready, ~(1u << s)) >> s) & 1u;
            if (uni(got)) epi_slot = s;
            idle = 0;
            continue;
        }
        // (c) a wavefront's worth of queued words -- or whatever is queued, when this wavefront has no tile to go on with
        if (((int32_t)avail >= 64 || ((int32_t)avail > 0 && !can_fe)) && af) {
            const int a = __builtin_ctz(af);
            uint32_t got = 0;
            if (lane == 0) got = (atomicAnd(&ctl.arena_free, ~(1u << a)) >> a) & 1u;
            if (uni(got)) {
                PT_ACC(2, merge(a));
                if (HUTK_PT_PROF) pc[6]++;
                if (lane == 0) atomicOr(&ctl.arena_free, 1u << a);
                idle = 0;
                continue;
            }
        }
        // (d) the front end of my next tile
        if (can_fe) {
            const int s = __builtin_ctz(free_now);
            uint32_t got = 0;
            if (lane == 0) got = (atomicAnd(&ctl.free_slots, ~(1u << s)) >> s) & 1u;
            if (uni(got)) {
                uint32_t nm;
                PT_ACC(1, nm = front_end(s));
                if (HUTK_PT_PROF) pc[5]++;
                if (nm != 0 && !reserve(nm)) {
                    pend_slot = s;
                    pend_nm = nm;
                } else {
                    if (nm != 0) enqueue(s, nm);
                    if (fe_over(s)) epi_slot = s;
                }
            }
            idle = 0;
            continue;
        }
        // (e) nothing to do right now
        if (pf_tile < 0 && pend_slot < 0 && uni(c_done) >= n_my) break;
        if (prof_on) {
            const long long t0_ = clock64();
            __builtin_amdgcn_s_sleep(8);
            pc[4] += clock64() - t0_;
        } else {
            __builtin_amdgcn_s_sleep(8);
        }
        if (++idle > (1u << 24)) {  // seconds: something above never finished; fail loudly instead of hanging
            if (lane == 0) raise(A.err, HUTK_E_DEVICE);
            break;
        }
    }
    if (prof_on && lane == 0) {
        pc[0] = clock64() - t_start;
        long long* out = W.prof + ((int64_t)blockIdx.x * PT_WAVES + wv) * 16;
        for (int k = 0; k < 16; k++) out[k] = pc[k];
    }
#undef PT_ACC
}

static int pt_grid() {
    static int g = 0;
    if (g == 0) {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 8)
            cus = 256;
        g = cus / 8 * 8 * PT_WGS;
    }
    return g;
}

// Does the persistent kernel take this batch?  (Byte-encoder mode, 16-bit symbols, rank == symbol order, every item one
// unit, the hand-written splitter, and enough tiles to give every compute unit a few.)
bool ptiles_takes(const DevTables& t, const BatchArgs& a) {
    static const int64_t min_tiles = getenv("HUTK_PTILES_MIN_TILES") ? atoll(getenv("HUTK_PTILES_MIN_TILES")) : 4 * (int64_t)pt_grid();  // (the tests set 1)
    return t.sym16 && t.is_byte_encoder && t.rank_is_sym && !t.has_multi && !a.word_bits && !a.first_bits &&
           a.n_tiles >= min_tiles;
}
void launch_ptiles(const DevTables& t, const BatchArgs& a, const Workspace& w, hipStream_t s) {
    hipLaunchKernelGGL(k_ptiles, dim3((unsigned)pt_grid()), dim3(64 * PT_WAVES), 0, s, t, a, w);
}

}  // namespace hutk
