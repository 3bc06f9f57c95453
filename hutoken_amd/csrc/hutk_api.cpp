// hutk_api.cpp -- C ABI of include/hutoken_amd.h: context life cycle, device
// tables, workspace and the launch sequence of one batch.
//
// There is no CPU compute path in this file: every encode entry point enqueues the
// HIP kernels of hutk_kernels.hip or fails with HUTK_E_DEVICE.
#include <hip/hip_runtime.h>

#include <regex.h>
#if defined(__linux__)
#include <sys/syscall.h>
#include <unistd.h>
#endif

#include <algorithm>
#include <atomic>
#include <cctype>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "hutk_classify.h"
#include "hutk_device.h"
#ifndef HUTK_PTILES_DEFAULT
#define HUTK_PTILES_DEFAULT 2  // 0: k_tiles; 1: k_ptiles; 2: auto
#endif

using namespace hutk;

namespace {
constexpr int64_t SMALL_BYTES = 32 * 1024, SMALL_DOCS = 1024;
constexpr size_t SMALL_HOST_BYTES = 256 * 1024;
thread_local std::string g_err = "";

int set_err(int code, const std::string& msg) {
    g_err = msg;
    return code;
}

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e__ = (expr);                                                               \
        if (e__ != hipSuccess)                                                                 \
            return set_err(HUTK_E_DEVICE, std::string("HIP error: ") + hipGetErrorString(e__) + \
                                              " at " #expr);                                    \
    } while (0)

template <class T>
struct DevBuf {
    T* p = nullptr;
    size_t cap = 0;  // elements
    hipError_t reserve(size_t n) {
        if (n <= cap) return hipSuccess;
        if (p) {
            (void)hipDeviceSynchronize();  // an earlier asynchronous call may still be using the old allocation
            (void)hipFree(p);
        }
        p = nullptr;
        cap = 0;
        size_t want = n + n / 8 + 64;
        hipError_t e = hipMalloc((void**)&p, want * sizeof(T));
        if (e == hipSuccess) cap = want;
        return e;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
};
}  // namespace

struct hutk_ctx {
    Tables tab;
    int device = -1;
    bool host_only = false;
    bool timing = true;

    // device tables
    DevBuf<uint64_t> d_pair, d_char;
    DevBuf<int32_t> d_sym_id, d_prefix_alone;
    DevBuf<uint32_t> d_item_sym, d_prefix_syms, d_prefix_alone_syms, d_seam, d_seam2, d_item_units;
    DevBuf<uint8_t> d_item_direct, d_split_dfa;
    DevBuf<uint32_t> d_bytepair16;  // {symbol, merged} as 16 + 16 bits
    DevBuf<WordSlot> d_word_tab;
    int64_t n_word_entries = 0, n_wordl_entries = 0;  // whole-word table entries in all, and those of the long-word companion
    DevBuf<uint64_t> d_bytepair32;  // {symbol, merged} as 32 + 32 bits
    DevBuf<long long> w_prof;
    bool profile = false;
    // regex pre-token path: the pattern of initialize() (empty: the hand-written splitter) and the bitmaps of a batch
    std::string pattern;
    DevBuf<uint32_t> w_wbits, w_gbits, w_fbits, w_abits;
    DevTables dt{};

    // workspace
    DevBuf<uint32_t> w_run;
    DevBuf<int32_t> w_exc_tok;
    DevBuf<uint32_t> w_exc_sym, w_exc_mrg, w_tile_u32, w_doc_pos, w_counters;
    DevBuf<int64_t> w_tile_i64;
    DevBuf<ExcRec> w_exc;
    DevBuf<uint32_t> w_exc_quad, w_exc_mid, w_exc_wave;

    // decode direction: tables and workspace
    DevBuf<uint2> d_dec_ent, d_dec_sent;
    DevBuf<uint8_t> d_dec_blob;
    DecTables dec{};
    DevBuf<uint32_t> dw_first;
    DevBuf<unsigned long long> dw_state;
    DevBuf<int64_t> dw_tfd;
    DevBuf<int32_t> ds_ids, ds_status;
    DevBuf<int64_t> ds_offs, ds_oo;
    DevBuf<uint8_t> ds_bytes;
    DevBuf<int32_t> w_err;

    // staging for the host-buffer entry point
    DevBuf<uint8_t> s_bytes;
    DevBuf<int64_t> s_offsets, s_out_offsets;
    DevBuf<int32_t> s_ids, s_status;
    // small batches: one page-locked host buffer, one device buffer each way
    DevBuf<uint8_t> s_small_in, s_small_out;
    void* small_host = nullptr;
    // pipelined host path (hutk_encode_batch on large batches): two sets of chunk buffers, copy streams,
    // pinned staging for the rebased offsets and the small per-chunk results
    struct Pipe {
        // THREE sets of chunk buffers: the copy up of chunk c is enqueued while chunk c - 2's copy down is still under way
        // (with two sets the host had to see that copy end first: a host round trip in the pipeline's critical path)
        static constexpr int NB = 3;
        DevBuf<uint8_t> bytes[NB];
        DevBuf<int64_t> offs[NB], offs_abs[NB], oo[NB], base;  // base: ids of the chunks already encoded
        DevBuf<int32_t> ids[NB], status[NB], err[NB];
        hipStream_t s_in = nullptr, s_out = nullptr;
        hipEvent_t ev_in[NB] = {}, ev_comp[NB] = {}, ev_out[NB] = {};
        // page-locked landing place of a chunk's error word and id total: a copy to PAGEABLE memory (a stack variable)
        // waits for the copy engine's whole queue -- the next chunk's copy up included -- and the two directions then
        // take turns instead of overlapping (tools/pipe_trace.py)
        int64_t* h_small = nullptr;
        bool ready = false;
    } pipe;

    hipStream_t stream = nullptr;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    bool ev_valid = false;
    // One workspace per context: calls on a context are SERIALISED.  The mutex orders the host side (calls from several
    // threads), the event orders the device side: every asynchronous call records it when its last kernel is enqueued,
    // and the next call's stream waits for it before its first kernel -- whatever streams the two calls run on.
    std::recursive_mutex mu;
    hipEvent_t ev_busy = nullptr;
    bool busy_valid = false;

    // Single-process multi-device dispatch (hutk_ctx_add_device): further contexts with the same tables on other
    // devices; hutk_encode_batch cuts a large batch into byte-balanced runs of whole documents, one per device, and
    // every run is encoded by its device's context on a host thread of its own.  peer_ids: page-locked landing area
    // of a peer's ids (they are copied to their place once the runs before them are counted).
    std::vector<hutk_ctx*> peers;
    struct PeerBuf { int32_t* p = nullptr; size_t cap = 0; };
    std::vector<PeerBuf> peer_ids;
};

namespace {

int upload_tables(hutk_ctx* c) {
    Tables& T = c->tab;
#define UP(buf, vec)                                                                            \
    do {                                                                                        \
        HIP_TRY((buf).reserve((vec).size() ? (vec).size() : 1));                                \
        if ((vec).size())                                                                       \
            HIP_TRY(hipMemcpy((buf).p, (vec).data(), (vec).size() * sizeof((vec)[0]), hipMemcpyHostToDevice)); \
    } while (0)
    UP(c->d_pair, T.pair_slots);
    UP(c->d_char, T.char_slots);
    UP(c->d_sym_id, T.sym_id);
    UP(c->d_prefix_syms, T.prefix_syms);
    // byte-encoder mode: (first byte, second byte) -> {symbol of the first byte, merged symbol of the pair}
    // in one entry, so the merge loop's set-up is one load per unit (second byte 0 = "no next unit").  The index is
    // first | second << 8: two consecutive input bytes read as one little-endian 16-bit word
    std::vector<uint32_t> bp16;
    std::vector<uint64_t> bp32;
    if (T.is_byte_encoder) {
        if (T.sym16) {
            bp16.assign(65536, 0xFFFFFFFFu);
            for (uint32_t b1 = 0; b1 < 256; b1++)
                for (uint32_t b2 = 0; b2 < 256; b2++)
                    bp16[b1 | (b2 << 8)] = (T.item_sym[b1] & 0xFFFFu) | ((uint32_t)T.bytepair16[(b1 << 8) | b2] << 16);
        } else {
            bp32.assign(65536, ~0ull);
            for (uint32_t b1 = 0; b1 < 256; b1++)
                for (uint32_t b2 = 0; b2 < 256; b2++)
                    bp32[b1 | (b2 << 8)] = (uint64_t)T.item_sym[b1] | ((uint64_t)T.bytepair32[(b1 << 8) | b2] << 32);
        }
    }
    UP(c->d_bytepair16, bp16);
    UP(c->d_bytepair32, bp32);
#undef UP
    HIP_TRY(c->d_item_sym.reserve(256));
    HIP_TRY(hipMemcpy(c->d_item_sym.p, T.item_sym, sizeof T.item_sym, hipMemcpyHostToDevice));
    HIP_TRY(c->d_item_direct.reserve(256));
    HIP_TRY(hipMemcpy(c->d_item_direct.p, T.item_direct, sizeof T.item_direct, hipMemcpyHostToDevice));
    HIP_TRY(c->d_prefix_alone.reserve(EXC_LDS_UNITS));

    DevTables& D = c->dt;
    D.pair_buckets = reinterpret_cast<const uint4*>(c->d_pair.p);
    D.pair_shift = T.pair_shift;
    D.sym_id = c->d_sym_id.p;
    D.n_vocab_sym = T.n_vocab_sym;
    D.n_sym = T.n_sym;
    {  // splitter automaton: transition table + byte classes, independent of the vocabulary
        std::vector<uint8_t> buf(dfa::TABLE_BYTES + 256);
        dfa::build(reinterpret_cast<uint16_t*>(buf.data()), buf.data() + dfa::TABLE_BYTES);
        for (uint32_t x = 0; x < 256; x++)  // the seam map rides in the rows' padding (hutk_classify.h)
            memcpy(buf.data() + dfa::seam_offset(x), &T.seam_hi[x], 4);
        HIP_TRY(c->d_split_dfa.reserve(buf.size()));
        HIP_TRY(hipMemcpy(c->d_split_dfa.p, buf.data(), buf.size(), hipMemcpyHostToDevice));
        D.split_dfa = reinterpret_cast<const uint4*>(c->d_split_dfa.p);
    }
    HIP_TRY(c->d_seam.reserve(256));
    HIP_TRY(hipMemcpy(c->d_seam.p, T.seam_hi, sizeof T.seam_hi, hipMemcpyHostToDevice));
    D.seam_hi = c->d_seam.p;
    D.seam_on = T.seam_on && !(getenv("HUTK_NO_SEAM") && atoi(getenv("HUTK_NO_SEAM"))) ? 1 : 0;
    {   // second level: seam2_part, then the hashed set of character pairs, in one buffer
        const size_t nb = T.seam2_on ? T.seam2_bits.size() : 1;
        std::vector<uint32_t> buf(256 + nb, 0u);
        memcpy(buf.data(), T.seam2_part, sizeof T.seam2_part);
        if (T.seam2_on) memcpy(buf.data() + 256, T.seam2_bits.data(), nb * 4);
        HIP_TRY(c->d_seam2.reserve(buf.size()));
        HIP_TRY(hipMemcpy(c->d_seam2.p, buf.data(), buf.size() * 4, hipMemcpyHostToDevice));
        D.seam2_part = c->d_seam2.p;
        D.seam2_bits = c->d_seam2.p + 256;
        D.seam2_shift = T.seam2_shift;
        D.seam2_cats = T.seam2_cats;
        D.seam2_on = D.seam_on && T.seam2_on && !(getenv("HUTK_NO_SEAM2") && atoi(getenv("HUTK_NO_SEAM2"))) ? 1 : 0;
    }
    {   // multi bits [8], unit offsets [257], units: one allocation
        std::vector<uint32_t> iu(T.multi_bits, T.multi_bits + 8);
        iu.insert(iu.end(), T.item_units_off, T.item_units_off + 257);
        iu.insert(iu.end(), T.item_units.begin(), T.item_units.end());
        HIP_TRY(c->d_item_units.reserve(iu.size()));
        HIP_TRY(hipMemcpy(c->d_item_units.p, iu.data(), iu.size() * 4, hipMemcpyHostToDevice));
        D.multi_bits = c->d_item_units.p;
        D.item_units_off = c->d_item_units.p + 8;
        D.item_units = c->d_item_units.p + 8 + 257;
        D.has_multi = T.has_multi ? 1 : 0;
        D.unit_scale = (int32_t)T.max_units_per_item;
    }
    D.item_sym = c->d_item_sym.p;
    D.item_direct = c->d_item_direct.p;
    D.char_slots = c->d_char.p;
    D.char_mask = T.char_mask;
    D.char_shift = T.char_shift;
    D.prefix_syms = c->d_prefix_syms.p;
    D.n_prefix = (int32_t)T.prefix_syms.size();
    D.prefix_alone_ids = c->d_prefix_alone.p;
    D.n_prefix_alone = 0;
    D.prefix_alone_syms = nullptr;
    D.is_byte_encoder = T.is_byte_encoder;
    D.has_prefix = T.has_prefix;
    D.rank_is_sym = T.rank_is_sym;
    D.ident_ids = T.ident_ids;
    D.sym16 = T.sym16;
    D.bytepair = T.sym16 ? (const void*)c->d_bytepair16.p : (const void*)c->d_bytepair32.p;
    D.word_tab = nullptr;
    D.word_mask = 0;
    D.wordl_off = 0;
    D.wordl_mask = 0;

    // the prefix encoded as a word of its own (core.c:421-446) is a constant of the
    // context: merge its units once, on the device, with the batch path's own loop
    if (T.has_prefix && T.prefix_alone_final) {
        // id-keyed path: the prefix as a word of its own was merged on the host (it stays string-keyed)
        if (T.prefix_alone_syms.size() > (size_t)EXC_LDS_UNITS) return set_err(HUTK_E_UNSUPPORTED, "prefix too long");
        HIP_TRY(c->d_prefix_alone_syms.reserve(T.prefix_alone_syms.size() + 1));
        if (!T.prefix_alone_syms.empty()) {
            HIP_TRY(hipMemcpy(c->d_prefix_alone_syms.p, T.prefix_alone_syms.data(), T.prefix_alone_syms.size() * 4,
                              hipMemcpyHostToDevice));
            HIP_TRY(hipMemcpy(c->d_prefix_alone.p, T.prefix_alone_ids.data(), T.prefix_alone_ids.size() * 4,
                              hipMemcpyHostToDevice));
        }
        D.n_prefix_alone = (int32_t)T.prefix_alone_syms.size();
        D.prefix_alone_syms = c->d_prefix_alone_syms.p;
    } else if (T.has_prefix && !T.prefix_alone_syms.empty()) {
        if (T.prefix_alone_syms.size() > (size_t)EXC_LDS_UNITS)
            return set_err(HUTK_E_UNSUPPORTED, "prefix too long");
        DevBuf<uint32_t>& d_syms = c->d_prefix_alone_syms;  // in: units; out: the merged symbols
        DevBuf<int32_t> d_n;
        HIP_TRY(d_syms.reserve(T.prefix_alone_syms.size()));
        HIP_TRY(d_n.reserve(1));
        HIP_TRY(hipMemcpy(d_syms.p, T.prefix_alone_syms.data(), T.prefix_alone_syms.size() * 4,
                          hipMemcpyHostToDevice));
        launch_bpe_symbols(D, d_syms.p, (int)T.prefix_alone_syms.size(), c->d_prefix_alone.p, d_n.p, c->stream);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(c->stream));
        int32_t n = 0;
        HIP_TRY(hipMemcpy(&n, d_n.p, 4, hipMemcpyDeviceToHost));
        D.n_prefix_alone = n;
        D.prefix_alone_syms = d_syms.p;
        d_n.release();
    }
    // ---- decode direction ----
    {
        const size_t N = (size_t)T.dec_n;
        std::vector<uint2> ent(N ? N : 1), sent;
        auto pack = [&](uint32_t off, uint32_t len, bool bad) {
            if (bad) return make_uint2(DEC_TAG_BAD, 0u);
            if (len > DEC_INLINE_MAX) return make_uint2(DEC_TAG_LONG | (len << 8), off);
            uint64_t v = len;
            for (uint32_t j = 0; j < len; j++) v |= (uint64_t)T.dec_blob[off + j] << (8 * (j + 1));
            return make_uint2((uint32_t)v, (uint32_t)(v >> 32));
        };
        // DEC_F_PFX_PARTIAL only matters at the front of a document
        for (size_t i = 0; i < N; i++)
            ent[i] = pack(T.dec_off[i], T.dec_len[i], T.dec_len[i] == DEC_BAD || (T.dec_flag[i] & ~DEC_F_PFX_PARTIAL));
        HIP_TRY(c->d_dec_ent.reserve(ent.size()));
        HIP_TRY(hipMemcpy(c->d_dec_ent.p, ent.data(), ent.size() * sizeof(uint2), hipMemcpyHostToDevice));
        c->dec.sent = nullptr;
        if (!T.dec_slen.empty()) {
            sent.resize(N ? N : 1);
            for (size_t i = 0; i < N; i++) {
                const bool strip = T.dec_slen[i] != DEC_NOSTRIP;
                const uint32_t len = strip ? T.dec_slen[i] : T.dec_len[i];
                sent[i] = pack(strip ? T.dec_soff[i] : T.dec_off[i], len, len == DEC_BAD || T.dec_flag[i]);
            }
            HIP_TRY(c->d_dec_sent.reserve(sent.size()));
            HIP_TRY(hipMemcpy(c->d_dec_sent.p, sent.data(), sent.size() * sizeof(uint2), hipMemcpyHostToDevice));
            c->dec.sent = c->d_dec_sent.p;
        }
        HIP_TRY(c->d_dec_blob.reserve(T.dec_blob.size() + 16));
        HIP_TRY(hipMemcpy(c->d_dec_blob.p, T.dec_blob.data(), T.dec_blob.size(), hipMemcpyHostToDevice));
        c->dec.ent = c->d_dec_ent.p;
        c->dec.blob = c->d_dec_blob.p;
        c->dec.n = T.dec_n;
    }
    return HUTK_OK;
}

int pad_per_doc(const hutk_ctx* c) {
    return c->tab.has_prefix ? (int)(c->tab.prefix_syms.size() + c->tab.prefix_alone_syms.size()) : 0;
}

// tile metadata packs five uint32 arrays and two int64 arrays into two allocations
int ensure_workspace(hutk_ctx* c, int64_t n_bytes, int64_t n_docs, int64_t n_tiles, Workspace& W) {
    const int64_t pad = pad_per_doc(c);
    const size_t run_elems = (size_t)(n_tiles * RUN_STRIDE + 512);
    const size_t exc_elems = (size_t)(n_bytes * (int64_t)c->tab.max_units_per_item + pad * (n_docs + 2) + 512);
    HIP_TRY(c->w_run.reserve(run_elems));
    HIP_TRY(c->w_exc_tok.reserve(exc_elems));
    HIP_TRY(c->w_exc_sym.reserve(exc_elems));
    HIP_TRY(c->w_exc_mrg.reserve(exc_elems));
    HIP_TRY(c->w_tile_u32.reserve((size_t)n_tiles * 8 + n_tiles / 32 + 16));
    HIP_TRY(c->w_tile_i64.reserve((size_t)n_tiles * 2 + n_tiles / 2048 + 16));
    HIP_TRY(c->w_doc_pos.reserve((size_t)n_docs + 2));
    if (!c->w_counters.p) {  // (zeroed once: counters[10], k_pre's sample, is zeroed for the NEXT call by k_scan / k_tail_small)
        HIP_TRY(c->w_counters.reserve(16));
        HIP_TRY(hipMemset(c->w_counters.p, 0, 16 * sizeof(uint32_t)));
    }
    HIP_TRY(c->w_err.reserve(1));
    // exception words: longer than a lane takes (more than LANE_MAX_UNITS bytes), first of their document, or cut off by
    // a tile's budget -- unless items of several units make ANY word one (then: at most a word per byte)
    const int64_t cap_exc = (c->tab.has_multi ? n_bytes : n_bytes / LANE_MAX_UNITS) + n_docs + n_tiles + 64;
    HIP_TRY(c->w_exc.reserve((size_t)cap_exc));
    HIP_TRY(c->w_exc_quad.reserve((size_t)cap_exc));
    HIP_TRY(c->w_exc_mid.reserve((size_t)cap_exc));
    HIP_TRY(c->w_exc_wave.reserve((size_t)cap_exc));
    W.run = c->w_run.p;
    W.exc_tok = c->w_exc_tok.p;
    W.exc_sym = c->w_exc_sym.p;
    W.exc_mrg = c->w_exc_mrg.p;
    uint32_t* u = c->w_tile_u32.p;
    W.tile_count = u;
    W.tile_dense = u + n_tiles;
    W.tile_run_start = u + 2 * n_tiles;
    W.tile_exc_first = u + 3 * n_tiles;
    W.tile_nexc = u + 4 * n_tiles;
    W.exc_tiles = u + 5 * n_tiles;
    W.tile_lastreal = u + 6 * n_tiles;
    W.tile_first_start = u + 7 * n_tiles;
    W.noreal_bits = u + 8 * n_tiles;
    W.tile_first_doc = c->w_tile_i64.p;
    W.tile_base = c->w_tile_i64.p + n_tiles;
    W.scan_state = reinterpret_cast<unsigned long long*>(c->w_tile_i64.p + 2 * n_tiles + 2);
    W.n_scan_blocks = scan_blocks(n_tiles);
    W.doc_tile_pos = c->w_doc_pos.p;
    W.exc = c->w_exc.p;
    W.exc_quad = c->w_exc_quad.p;
    W.exc_mid = c->w_exc_mid.p;
    W.exc_wave = c->w_exc_wave.p;
    W.counters = c->w_counters.p;
    W.cap_exc = cap_exc;
    W.pad_per_doc = (int32_t)pad;
    W.prof = nullptr;
    if (c->profile) {
        HIP_TRY(c->w_prof.reserve((size_t)n_tiles * 10 + 16));
        W.prof = c->w_prof.p;
    }
    return HUTK_OK;
}

void destroy(hutk_ctx* c) {
    if (!c) return;
    for (hutk_ctx* p : c->peers) destroy(p);
    c->peers.clear();
    for (auto& b : c->peer_ids)
        if (b.p) (void)hipHostFree(b.p);
    c->peer_ids.clear();
    if (!c->host_only && c->device >= 0) {
        (void)hipSetDevice(c->device);
        c->d_pair.release(); c->d_char.release(); c->d_sym_id.release(); c->d_prefix_alone.release();
        c->d_item_sym.release(); c->d_prefix_syms.release(); c->d_prefix_alone_syms.release(); c->d_item_direct.release(); c->d_split_dfa.release(); c->d_seam.release(); c->d_seam2.release(); c->d_item_units.release();
        c->d_bytepair16.release(); c->d_bytepair32.release(); c->w_prof.release();
        c->d_word_tab.release(); c->w_wbits.release(); c->w_gbits.release(); c->w_fbits.release(); c->w_abits.release();
        c->w_run.release(); c->w_exc_tok.release(); c->w_exc_sym.release(); c->w_exc_mrg.release();
        c->w_tile_u32.release(); c->w_doc_pos.release(); c->w_counters.release(); c->w_tile_i64.release();
        c->w_exc.release(); c->w_exc_quad.release(); c->w_exc_mid.release(); c->w_exc_wave.release();
        c->d_dec_ent.release(); c->d_dec_sent.release(); c->d_dec_blob.release(); c->dw_first.release();
        c->dw_state.release(); c->dw_tfd.release(); c->ds_ids.release(); c->ds_status.release();
        c->ds_offs.release(); c->ds_oo.release(); c->ds_bytes.release(); c->w_err.release();
        c->s_bytes.release(); c->s_offsets.release(); c->s_out_offsets.release(); c->s_ids.release();
        c->s_status.release(); c->s_small_in.release(); c->s_small_out.release();
        if (c->small_host) (void)hipHostFree(c->small_host);
        for (int b = 0; b < hutk_ctx::Pipe::NB; b++) {
            c->pipe.bytes[b].release(); c->pipe.offs[b].release(); c->pipe.oo[b].release(); c->pipe.offs_abs[b].release();
            c->pipe.ids[b].release(); c->pipe.status[b].release(); c->pipe.err[b].release();
            if (c->pipe.ev_in[b]) (void)hipEventDestroy(c->pipe.ev_in[b]);
            if (c->pipe.ev_comp[b]) (void)hipEventDestroy(c->pipe.ev_comp[b]);
            if (c->pipe.ev_out[b]) (void)hipEventDestroy(c->pipe.ev_out[b]);
        }
        c->pipe.base.release();
        if (c->pipe.h_small) (void)hipHostFree(c->pipe.h_small);
        if (c->pipe.s_in) (void)hipStreamDestroy(c->pipe.s_in);
        if (c->pipe.s_out) (void)hipStreamDestroy(c->pipe.s_out);
        for (auto& e : c->ev)
            if (e) (void)hipEventDestroy(e);
        if (c->ev_busy) (void)hipEventDestroy(c->ev_busy);
        if (c->stream) (void)hipStreamDestroy(c->stream);
    }
    delete c;
}

// Whole-word table (byte-encoder mode, no prefix).  Every vocabulary key that can be
// written in raw input bytes (<= 16) is pushed through the device pipeline as a document
// of its own; the ones that come back as exactly their own id become table entries.
// Nothing is assumed about the vocabulary: the merge kernel itself is the judge.
int build_word_table(hutk_ctx* c) {
    Tables& T = c->tab;
    const size_t n = T.cand_sym.size();
    if (!n) return HUTK_OK;
    std::vector<int64_t> offs(n + 1), oo(n + 1);
    for (size_t i = 0; i <= n; i++) offs[i] = T.cand_off[i];
    std::vector<int32_t> ids((size_t)hutk_ids_capacity(c, (int64_t)T.cand_bytes.size(), (int64_t)n));
    // candidates are encoded one per document, as words that are NOT first in their document: no prefix
    const int had_prefix = c->dt.has_prefix;
    c->dt.has_prefix = 0;
    int rc = hutk_encode_batch(c, T.cand_bytes.data(), offs.data(), (int64_t)n, ids.data(), (int64_t)ids.size(),
                               oo.data(), nullptr);
    c->dt.has_prefix = had_prefix;
    if (rc) return rc;
    const int64_t key_bytes = T.sym16 ? WORD_KEY_BYTES_16 : WORD_KEY_BYTES_32;  // (WordSlot, hutk_device.h)
    std::vector<size_t> keep, keep_long;
    for (size_t i = 0; i < n; i++)
        if (oo[i + 1] - oo[i] == 1 && ids[oo[i]] == T.sym_id[T.cand_sym[i]] && offs[i + 1] - offs[i] >= 2 &&
            offs[i + 1] - offs[i] <= WORDL_KEY_BYTES)
            (offs[i + 1] - offs[i] <= key_bytes ? keep : keep_long).push_back(i);
    auto key_of = [&](size_t i, uint32_t* k) {  // the word's raw bytes as seven little-endian dwords, zero padded
        for (int q = 0; q < 7; q++) k[q] = 0;
        const size_t len = (size_t)(offs[i + 1] - offs[i]);
        for (size_t j = 0; j < len; j++) k[j >> 2] |= (uint32_t)T.cand_bytes[offs[i] + j] << (8 * (j & 3));
    };
    // main table: two-choice cuckoo of 16-byte slots (hutk_device.h); a word that cannot be placed is simply left out
    std::vector<WordSlot> slots(1, WordSlot{{0, 0, 0, 0}});
    uint32_t cap = 0;
    if (!keep.empty()) {
        std::vector<uint4> kk(keep.size());
        for (size_t q = 0; q < keep.size(); q++) {
            uint32_t k[7];
            key_of(keep[q], k);
            kk[q] = make_uint4(k[0], k[1], k[2], k[3]);
        }
        cap = 1024;
        while (cap < keep.size() * 5 / 2 + 16) cap <<= 1;
        std::vector<uint32_t> where, homeless;
        auto hash_of = [&](uint32_t j) { const uint4 k = kk[j]; return word_hash(k.x, k.y, k.z, k.w); };
        cuckoo_place(keep.size(), cap, [&](uint32_t j) { return hash_of(j) & (cap - 1); },
                     [&](uint32_t j) { return word_slot2(hash_of(j), cap - 1); }, where, &homeless);
        slots.assign(cap + 1, WordSlot{{0, 0, 0, 0}});
        for (size_t j = 0; j < keep.size(); j++) {
            if (where[j] == 0xFFFFFFFFu) continue;
            const uint4 k = kk[j];
            slots[where[j]] = T.sym16 ? WordSlot{{k.x, k.y, k.z, k.w | (T.cand_sym[keep[j]] << 16)}}
                                      : WordSlot{{k.x, k.y, k.z, T.cand_sym[keep[j]]}};
            c->n_word_entries++;
        }
    }
    // the companion (WordSlotLong = two consecutive 16-byte slots behind the main table): ONE slot per word
    // (word_hash_long & mask), first come first served in vocabulary order; a word that finds its slot taken is left out
    // and goes through the merge loop as before (the table is 8 x the words: about 1 in 17)
    uint32_t lcap = 0, loff = 0;
    if (cap && !keep_long.empty() && !getenv("HUTK_NO_LONG_WORD_TABLE")) {
        lcap = 256;
        while (lcap < keep_long.size() * 8 + 16) lcap <<= 1;
        loff = (uint32_t)slots.size();
        loff += loff & 1u;  // (32-byte aligned)
        slots.resize((size_t)loff + 2 * (size_t)lcap, WordSlot{{0, 0, 0, 0}});
        for (size_t j = 0; j < keep_long.size(); j++) {
            uint32_t k[7];
            key_of(keep_long[j], k);
            const size_t at = (size_t)loff + 2 * (size_t)(word_hash_long(k[0], k[1], k[2], k[3], k[4], k[5], k[6]) & (lcap - 1));
            if (slots[at].k[0] | slots[at].k[1] | slots[at].k[2] | slots[at].k[3]) continue;
            slots[at] = WordSlot{{k[0], k[1], k[2], k[3]}};
            slots[at + 1] = WordSlot{{k[4], k[5], k[6], T.cand_sym[keep_long[j]]}};
            c->n_word_entries++;
            c->n_wordl_entries++;
        }
    }
    if (!cap) return HUTK_OK;
    HIP_TRY(c->d_word_tab.reserve(slots.size()));
    HIP_TRY(hipMemcpy(c->d_word_tab.p, slots.data(), slots.size() * sizeof(WordSlot), hipMemcpyHostToDevice));
    c->dt.word_tab = c->d_word_tab.p;
    c->dt.word_mask = cap - 1;
    c->dt.wordl_off = loff;
    c->dt.wordl_mask = lcap ? lcap - 1 : 0;
    return HUTK_OK;
}

// The device half of a context whose host tables are loaded: stream, events, tables and whole-word table on `device`
// (-1: the calling thread's current device).
int attach_device(hutk_ctx* c, int device) {
    int n_dev = 0;
    hipError_t e = hipGetDeviceCount(&n_dev);
    if (e != hipSuccess || n_dev <= 0)
        return set_err(HUTK_E_DEVICE, "no HIP device available: the hutoken_amd encode path runs on the GPU only");
    if (device < 0) {
        if (hipGetDevice(&device) != hipSuccess) device = 0;
    }
    if (device >= n_dev) return set_err(HUTK_E_ARG, "device ordinal out of range");
    c->device = device;
    if (hipSetDevice(device) != hipSuccess) return set_err(HUTK_E_DEVICE, "hipSetDevice failed");
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess)
        return set_err(HUTK_E_DEVICE, "hipStreamCreate failed");
    bool ok = true;
    for (auto& ev : c->ev) ok = ok && hipEventCreate(&ev) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&c->ev_busy, hipEventDisableTiming) == hipSuccess;
    if (!ok) return set_err(HUTK_E_DEVICE, "hipEventCreate failed");
    int rc = upload_tables(c);
    if (rc == HUTK_OK && !getenv("HUTK_NO_WORD_TABLE")) rc = build_word_table(c);
    return rc;
}

}  // namespace

extern "C" {

const char* hutk_last_error(void) { return g_err.c_str(); }

int hutk_ctx_create(hutk_ctx** out, const char* vocab_path, const char* special_path,
                    const char* prefix, int is_byte_encoder, int device) {
    return hutk_ctx_create_merges(out, vocab_path, special_path, prefix, is_byte_encoder, nullptr, device);
}

int hutk_ctx_create_merges(hutk_ctx** out, const char* vocab_path, const char* special_path, const char* prefix,
                           int is_byte_encoder, const char* merges_path, int device) {
    if (!out) return set_err(HUTK_E_ARG, "out is NULL");
    *out = nullptr;
    if (!vocab_path || !special_path)
        return set_err(HUTK_E_ARG,
                       "Invalid arguments. Expected a string (vocab_file_path), a string "
                       "(special_file_path)");
    hutk_ctx* c = new (std::nothrow) hutk_ctx();
    if (!c) return set_err(HUTK_E_MEMORY, "out of memory");
    LoadError le = load_tables(vocab_path, special_path, prefix, is_byte_encoder != 0, merges_path, c->tab);
    if (le.code) {
        delete c;
        return set_err(le.code, le.msg);
    }
    if (device == -2) {  // host-only context: tables for inspection, no encode
        c->host_only = true;
        *out = c;
        return HUTK_OK;
    }
    const int rc = attach_device(c, device);
    if (rc) {
        std::string keep = g_err;
        destroy(c);
        g_err = keep;
        return rc;
    }
    *out = c;
    return HUTK_OK;
}

int hutk_ctx_add_device(hutk_ctx* c, int device) {
    if (!c) return set_err(HUTK_E_ARG, "ctx is NULL");
    if (c->host_only) return set_err(HUTK_E_DEVICE, "host-only context: no device to encode on");
    if (device < 0) return set_err(HUTK_E_ARG, "device ordinal out of range");
    std::lock_guard<std::recursive_mutex> lock(c->mu);
    hutk_ctx* p = new (std::nothrow) hutk_ctx();
    if (!p) return set_err(HUTK_E_MEMORY, "out of memory");
    p->tab = c->tab;  // the host tables as loaded: same files, same ids
    p->timing = false;
    const int rc = attach_device(p, device);  // (builds the whole-word table with the hand-written splitter, like the first device)
    if (rc) {
        std::string keep = g_err;
        destroy(p);
        g_err = keep;
        return rc;
    }
    p->pattern = c->pattern;
    c->peers.push_back(p);
    c->peer_ids.emplace_back();
    return HUTK_OK;
}

int hutk_ctx_device_count(const hutk_ctx* c) { return c && !c->host_only ? 1 + (int)c->peers.size() : 0; }

void hutk_ctx_destroy(hutk_ctx* ctx) { destroy(ctx); }

int64_t hutk_ids_capacity(const hutk_ctx* ctx, int64_t n_bytes, int64_t n_docs) {
    if (!ctx) return 0;
    // at most one id per unit: a unit per input byte, or max_units_per_item where a replacement has several (SURVEY 8 b)
    return n_bytes * (int64_t)ctx->tab.max_units_per_item + (int64_t)pad_per_doc(ctx) * n_docs + 1;
}

int64_t hutk_vocab_size(const hutk_ctx* ctx) { return ctx ? ctx->tab.n_keys : 0; }
int hutk_uses_merges(const hutk_ctx* ctx) { return ctx && ctx->tab.id_path ? 1 : 0; }
int64_t hutk_pair_table_entries(const hutk_ctx* ctx) { return ctx ? ctx->tab.n_pairs : 0; }
int hutk_device_ordinal(const hutk_ctx* ctx) { return ctx ? ctx->device : -1; }
int64_t hutk_debug_pairs_second(const hutk_ctx* ctx) { return ctx ? ctx->tab.n_pairs_second : 0; }
int64_t hutk_debug_long_words(const hutk_ctx* ctx) { return ctx ? ctx->n_wordl_entries : 0; }
int hutk_debug_seam(const hutk_ctx* ctx, uint32_t* out256) {
    if (!ctx || !out256) return set_err(HUTK_E_ARG, "bad argument");
    memcpy(out256, ctx->tab.seam_hi, sizeof ctx->tab.seam_hi);
    return ctx->tab.seam_on && !(getenv("HUTK_NO_SEAM") && atoi(getenv("HUTK_NO_SEAM"))) ? 1 : 0;
}
// second level: 1 when it is on and says that NO token can span the boundary between the three-byte characters a3 | b3
// (little-endian 24-bit values) -- asked where hutk_debug_seam's map says "may join"; 0 otherwise
int hutk_debug_seam2_cut(const hutk_ctx* ctx, uint32_t a3, uint32_t b3) {
    if (!ctx) return 0;
    const Tables& T = ctx->tab;
    if (!T.seam_on || !T.seam2_on || (getenv("HUTK_NO_SEAM") && atoi(getenv("HUTK_NO_SEAM"))) ||
        (getenv("HUTK_NO_SEAM2") && atoi(getenv("HUTK_NO_SEAM2"))))
        return 0;
    if (!seam2_char3(a3) || !seam2_char3(b3)) return 0;
    if ((T.seam2_part[(a3 >> 16) & 0xFFu] >> (b3 & 31u)) & 1u) return 0;
    const uint32_t kinds[5][2] = {{3, 3}, {3, 2}, {3, 1}, {1, 3}, {1, 2}};
    for (auto& kk : kinds) {
        if (!(T.seam2_cats & seam2_cat_bit(kk[0], kk[1]))) continue;
        const uint32_t a = kk[0] == 3 ? a3 : (a3 >> 16) & 0xFFu;
        const uint32_t b = kk[1] == 3 ? b3 : kk[1] == 2 ? (b3 & 0xFFFFu) : (b3 & 0xFFu);
        const uint32_t h = seam2_hash(a, b, kk[0], kk[1]) >> T.seam2_shift;
        if ((T.seam2_bits[h >> 5] >> (h & 31)) & 1u) return 0;
    }
    return 1;
}
void hutk_set_timing(hutk_ctx* ctx, int enabled) {
    if (ctx) ctx->timing = enabled != 0;
}

int hutk_table_stats(const hutk_ctx* ctx, int64_t* out8) {
    if (!ctx || !out8) return HUTK_E_ARG;
    const Tables& T = ctx->tab;
    out8[0] = T.n_keys;
    out8[1] = T.n_vocab_sym;
    out8[2] = T.n_sym;
    out8[3] = T.n_pairs;
    out8[4] = (int64_t)T.pair_slots.size();
    out8[5] = T.rank_is_sym;
    out8[6] = T.ident_ids;
    out8[7] = ctx->n_word_entries;
    return HUTK_OK;
}

// Diagnostic: per-phase clock64 stamps of k_tiles.  enable, run one batch, then read
// the mean cycles between consecutive stamps over the first n_tiles tiles.
int hutk_debug_profile(hutk_ctx* c, int enable) {
    if (!c) return HUTK_E_ARG;
    c->profile = enable != 0;
    return HUTK_OK;
}
int hutk_debug_tile_bytes(void) { return TILE_BYTES; }
int hutk_debug_profile_read(hutk_ctx* c, int64_t n_tiles, double* out10) {
    if (!c || !out10 || !c->w_prof.p || n_tiles <= 0) return set_err(HUTK_E_ARG, "no profile");
    std::vector<long long> h((size_t)n_tiles * 10);
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(h.data(), c->w_prof.p, h.size() * 8, hipMemcpyDeviceToHost));
    for (int k = 0; k < 10; k++) out10[k] = 0;
    for (int64_t t = 0; t < n_tiles; t++)
        for (int k = 1; k < 10; k++) out10[k] += (double)(h[t * 10 + k] - h[t * 10 + k - 1]);
    for (int k = 1; k < 10; k++) out10[k] /= (double)n_tiles;
    for (int k = 1; k < 10; k++) out10[0] += out10[k];
    return HUTK_OK;
}

// the stamps themselves, ten per tile
int hutk_debug_profile_raw(hutk_ctx* c, int64_t n_tiles, long long* out) {
    if (!c || !out || !c->w_prof.p || n_tiles <= 0) return set_err(HUTK_E_ARG, "no profile");
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out, c->w_prof.p, (size_t)n_tiles * 10 * 8, hipMemcpyDeviceToHost));
    return HUTK_OK;
}

static int encode_device_impl(hutk_ctx* c, const uint8_t* d_bytes, const int64_t* d_offsets, int64_t n_docs,
                              int64_t n_bytes, int32_t* d_ids_out, int64_t ids_cap, int64_t* d_out_offsets,
                              int32_t* d_status, int32_t* d_err, void* hip_stream, const uint32_t* d_word_bits,
                              const uint32_t* d_gap_bits, const uint32_t* d_first_bits = nullptr,
                              const uint32_t* d_alone_bits = nullptr);

static int regex_bitmaps(const std::string& pattern, const uint8_t* bytes, const int64_t* offsets, int64_t n_docs,
                         std::vector<uint32_t>& wbits, std::vector<uint32_t>& gbits, std::vector<uint8_t>& too_large,
                         std::vector<uint32_t>* fbits, std::vector<uint32_t>* abits);

int hutk_encode_batch_device(hutk_ctx* c, const uint8_t* d_bytes, const int64_t* d_offsets,
                             int64_t n_docs, int64_t n_bytes, int32_t* d_ids_out, int64_t ids_cap,
                             int64_t* d_out_offsets, int32_t* d_status, int32_t* d_err,
                             void* hip_stream) {
    std::string pattern;  // (a copy: hutk_ctx_set_pattern may change the context's under the same mutex while this call runs)
    if (c && !c->host_only) {
        std::lock_guard<std::recursive_mutex> lock(c->mu);
        pattern = c->pattern;
    }
    if (!pattern.empty()) {
        // The regex pre-token path splits with libc's regexec (core.c:350-378), which runs on the host: the bytes and offsets
        // come down once, the bitmaps of the matches go up, and the encode itself stays on the device buffers.  This form
        // of the call therefore SYNCHRONISES with the stream (the only one that does) -- on EVERY way out once a copy from
        // the vectors below has been queued: they are the copies' sources.
        if (n_docs < 0 || n_bytes < 0 || !d_offsets || (n_bytes > 0 && !d_bytes)) return set_err(HUTK_E_ARG, "bad argument");
        std::lock_guard<std::recursive_mutex> lock(c->mu);
        HIP_TRY(hipSetDevice(c->device));
        hipStream_t s = hip_stream ? (hipStream_t)hip_stream : c->stream;
        std::vector<uint8_t> hb((size_t)n_bytes + 1);
        std::vector<int64_t> ho((size_t)n_docs + 1);
        std::vector<uint32_t> wbits, gbits, fbits, abits;
        std::vector<uint8_t> too_large;
        struct Drain {  // declared behind the vectors: runs before they are freed
            hipStream_t s;
            ~Drain() { (void)hipStreamSynchronize(s); }
        } drain{s};
        if (n_bytes) HIP_TRY(hipMemcpyAsync(hb.data(), d_bytes, (size_t)n_bytes, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipMemcpyAsync(ho.data(), d_offsets, (size_t)(n_docs + 1) * 8, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        if (ho[0] != 0 || ho[(size_t)n_docs] != n_bytes) return set_err(HUTK_E_ARG, "n_bytes must equal offsets[n_docs]");
        for (int64_t i = 0; i < n_docs; i++)
            if (ho[(size_t)i + 1] < ho[(size_t)i]) return set_err(HUTK_E_ARG, "offsets must not decrease");
        if (memchr(hb.data(), 0, (size_t)n_bytes)) return set_err(HUTK_E_NUL_BYTE, "a document contains a 0x00 byte");
        const bool pfx = c->tab.has_prefix;
        int rc = regex_bitmaps(pattern, hb.data(), ho.data(), n_docs, wbits, gbits, too_large, pfx ? &fbits : nullptr,
                               pfx ? &abits : nullptr);
        if (rc) return set_err(rc, "Regex could not be compiled.");
        if (c->busy_valid) HIP_TRY(hipStreamWaitEvent(s, c->ev_busy, 0));  // (the bitmaps are the context's: the previous call may still read them)
        HIP_TRY(c->w_wbits.reserve(wbits.size()));
        HIP_TRY(c->w_gbits.reserve(gbits.size()));
        HIP_TRY(hipMemcpyAsync(c->w_wbits.p, wbits.data(), wbits.size() * 4, hipMemcpyHostToDevice, s));
        HIP_TRY(hipMemcpyAsync(c->w_gbits.p, gbits.data(), gbits.size() * 4, hipMemcpyHostToDevice, s));
        if (pfx) {
            HIP_TRY(c->w_fbits.reserve(fbits.size()));
            HIP_TRY(c->w_abits.reserve(abits.size()));
            HIP_TRY(hipMemcpyAsync(c->w_fbits.p, fbits.data(), fbits.size() * 4, hipMemcpyHostToDevice, s));
            HIP_TRY(hipMemcpyAsync(c->w_abits.p, abits.data(), abits.size() * 4, hipMemcpyHostToDevice, s));
        }
        rc = encode_device_impl(c, d_bytes, d_offsets, n_docs, n_bytes, d_ids_out, ids_cap, d_out_offsets, d_status, d_err,
                                hip_stream, c->w_wbits.p, c->w_gbits.p, pfx ? c->w_fbits.p : nullptr, pfx ? c->w_abits.p : nullptr);
        if (rc) return rc;
        if (d_status) {  // a match over the reference's limit ends its document (found by the host: core.c:402-407)
            static const int32_t st_too_large = HUTK_DOC_WORD_TOO_LARGE;
            for (int64_t d = 0; d < n_docs; d++)
                if (too_large[(size_t)d]) HIP_TRY(hipMemcpyAsync(d_status + d, &st_too_large, 4, hipMemcpyHostToDevice, s));
        }
        return HUTK_OK;  // (Drain: the stream is waited for here)
    }
    return encode_device_impl(c, d_bytes, d_offsets, n_docs, n_bytes, d_ids_out, ids_cap, d_out_offsets, d_status, d_err,
                              hip_stream, nullptr, nullptr);
}

// Which tile kernel for the batches both can take: HUTK_PTILES=1 / 0 the persistent one (hutk_ptiles.hip) / k_tiles;
// unset ("auto"): both are enqueued and the batch's own bytes decide on the device (Workspace::select): the persistent
// kernel for text dense in three- and four-byte characters (2.4x on CJK paragraphs), k_tiles for everything else.
static int ptiles_mode() {
    const char* e = getenv("HUTK_PTILES");
    return e ? (atoi(e) != 0 ? 1 : 0) : HUTK_PTILES_DEFAULT;
}

static int encode_device_impl(hutk_ctx* c, const uint8_t* d_bytes, const int64_t* d_offsets, int64_t n_docs,
                              int64_t n_bytes, int32_t* d_ids_out, int64_t ids_cap, int64_t* d_out_offsets,
                              int32_t* d_status, int32_t* d_err, void* hip_stream, const uint32_t* d_word_bits,
                              const uint32_t* d_gap_bits, const uint32_t* d_first_bits, const uint32_t* d_alone_bits) {
    if (!c) return set_err(HUTK_E_ARG, "ctx is NULL");
    if (c->host_only) return set_err(HUTK_E_DEVICE, "host-only context: no device to encode on");
    if (n_docs < 0 || n_bytes < 0 || !d_offsets || !d_out_offsets || (n_bytes > 0 && (!d_bytes || !d_ids_out)))
        return set_err(HUTK_E_ARG, "bad argument");
    if (((uintptr_t)d_bytes & 15u) != 0) return set_err(HUTK_E_ARG, "d_bytes must be 16-byte aligned");
    if (ids_cap < hutk_ids_capacity(c, n_bytes, n_docs) - 1)
        return set_err(HUTK_E_CAPACITY, "ids_cap is below hutk_ids_capacity()");
    std::lock_guard<std::recursive_mutex> lock(c->mu);
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t s = hip_stream ? (hipStream_t)hip_stream : c->stream;
    if (c->busy_valid) HIP_TRY(hipStreamWaitEvent(s, c->ev_busy, 0));  // the previous call still owns the workspace
    struct BusyMark {  // recorded on every way out, behind whatever this call enqueued
        hutk_ctx* c; hipStream_t s;
        ~BusyMark() { if (hipEventRecord(c->ev_busy, s) == hipSuccess) c->busy_valid = true; }
    } busy_mark{c, s};
    const int64_t n_tiles = (n_bytes + TILE_BYTES - 1) / TILE_BYTES;
    if (n_tiles > 0x7FFFFFFFll) return set_err(HUTK_E_ARG, "batch too large");
    Workspace W{};
    int rc = ensure_workspace(c, n_bytes, n_docs, n_tiles, W);
    if (rc) return rc;
    BatchArgs A{};
    A.bytes = d_bytes;
    A.offsets = d_offsets;
    A.n_docs = n_docs;
    A.n_bytes = n_bytes;
    A.n_tiles = n_tiles;
    A.ids_out = d_ids_out;
    A.ids_cap = ids_cap;
    A.out_offsets = d_out_offsets;
    A.status = d_status;
    A.err = d_err ? d_err : c->w_err.p;
    A.word_bits = d_word_bits;
    A.gap_bits = d_gap_bits;
    A.first_bits = d_first_bits;
    A.alone_bits = d_alone_bits;

    c->ev_valid = false;
    if (c->timing) HIP_TRY(hipEventRecord(c->ev[0], s));
    if (n_tiles == 0) {
        HIP_TRY(hipMemsetAsync(A.err, 0, 4, s));
        if (d_status && n_docs) HIP_TRY(hipMemsetAsync(d_status, 0, (size_t)n_docs * 4, s));
        HIP_TRY(hipMemsetAsync(d_out_offsets, 0, (size_t)(n_docs + 1) * 8, s));
        if (c->timing) {
            HIP_TRY(hipEventRecord(c->ev[1], s));
            HIP_TRY(hipEventRecord(c->ev[2], s));
            HIP_TRY(hipEventRecord(c->ev[3], s));
            c->ev_valid = true;
        }
        return HUTK_OK;
    }
    launch_pre(A, W, s);
    if (c->timing) HIP_TRY(hipEventRecord(c->ev[1], s));
    {
        const int mode = ptiles_takes(c->dt, A) ? ptiles_mode() : 0;
        W.select = mode == 2 && c->dt.seam_on ? 1 : 0;
        if (mode == 1 || W.select == 1) launch_ptiles(c->dt, A, W, s);
        if (mode != 1) {
            W.select = W.select ? 2 : 0;
            launch_tiles(c->dt, A, W, s);
        }
        W.select = 0;
    }
    if (c->timing) HIP_TRY(hipEventRecord(c->ev[2], s));
    if (small_tail(A)) {
        launch_tail_small(c->dt, A, W, s);
    } else {
        launch_exceptions(c->dt, A, W, s);
        launch_scan(A, W, s);
        launch_finish(c->dt, A, W, s);
        if (n_bytes > MAX_WORD_BYTES) launch_cut(c->dt, A, W, s);
    }
    HIP_TRY(hipGetLastError());
    if (c->timing) {
        HIP_TRY(hipEventRecord(c->ev[3], s));
        c->ev_valid = true;
    }
    return HUTK_OK;
}

// A batch of at most four tiles (hutk_encode(): a sentence) in ONE launch and without a stream synchronisation: every buffer of
// the call lives in the context's page-locked, device-mapped staging area; k_tiles<..., ONE> does the whole pipeline and
// raises *flag, which this thread polls (the reference answers such a call in ~20 us on one core, lib.c:668-720; three launches,
// two copies and a hipStreamSynchronize were 56-64 us).  flag 2: the batch has exception words, the tail is launched behind.
static int encode_one_shot(hutk_ctx* c, const uint8_t* m_bytes, const int64_t* m_offsets, int64_t n_docs, int64_t n_bytes,
                           int32_t* m_ids, int64_t ids_cap, int64_t* m_oo, int32_t* m_status, int32_t* m_err, int32_t* m_flag) {
    hipStream_t s = c->stream;
    if (c->busy_valid) HIP_TRY(hipStreamWaitEvent(s, c->ev_busy, 0));
    struct BusyMark {
        hutk_ctx* c; hipStream_t s;
        ~BusyMark() { if (hipEventRecord(c->ev_busy, s) == hipSuccess) c->busy_valid = true; }
    } busy_mark{c, s};
    const int64_t n_tiles = (n_bytes + TILE_BYTES - 1) / TILE_BYTES;
    Workspace W{};
    int rc = ensure_workspace(c, n_bytes, n_docs, n_tiles, W);
    if (rc) return rc;
    BatchArgs A{};
    A.bytes = m_bytes;
    A.offsets = m_offsets;
    A.n_docs = n_docs;
    A.n_bytes = n_bytes;
    A.n_tiles = n_tiles;
    A.ids_out = m_ids;
    A.ids_cap = ids_cap;
    A.out_offsets = m_oo;
    A.status = m_status;
    A.err = m_err;
    W.one_flag = m_flag;
    HIP_TRY(c->s_small_in.reserve((((size_t)n_docs + 1) * 8 + 15 + (size_t)n_bytes + 15 + 64)));
    W.one_in = c->s_small_in.p;
    c->ev_valid = false;
    __atomic_store_n(m_flag, 0, __ATOMIC_RELEASE);
    launch_one_shot(c->dt, A, W, s);
    HIP_TRY(hipGetLastError());
    int32_t f = 0;
    const auto t0 = std::chrono::steady_clock::now();
    for (uint64_t spins = 0; (f = __atomic_load_n(m_flag, __ATOMIC_ACQUIRE)) == 0; spins++) {
#if defined(__x86_64__)
        __builtin_ia32_pause();
#endif
        if ((spins & 0xFFFFu) == 0xFFFFu && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(5)) {
            HIP_TRY(hipStreamSynchronize(s));  // (a failed launch shows here; a kernel that ended without raising the flag cannot happen)
            f = __atomic_load_n(m_flag, __ATOMIC_ACQUIRE);
            if (f == 0) return set_err(HUTK_E_DEVICE, "the one-launch encode did not complete");
            break;
        }
    }
    if (f == 2) {  // exception words: their stages, the scan and the copy-out behind the tile kernel
        W.one_flag = nullptr;
        launch_tail_small(c->dt, A, W, s);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(s));
    }
    return HUTK_OK;
}

int hutk_last_timing(hutk_ctx* c, float* ms_tile_kernel, float* ms_total) {
    if (!c || !c->ev_valid) return set_err(HUTK_E_ARG, "no timed call yet");
    HIP_TRY(hipEventSynchronize(c->ev[3]));
    float a = 0, b = 0;
    HIP_TRY(hipEventElapsedTime(&a, c->ev[1], c->ev[2]));
    HIP_TRY(hipEventElapsedTime(&b, c->ev[0], c->ev[3]));
    if (ms_tile_kernel) *ms_tile_kernel = a;
    if (ms_total) *ms_total = b;
    return HUTK_OK;
}

static int encode_batch_simple(hutk_ctx* c, const uint8_t* bytes, const int64_t* offsets, int64_t n_docs,
                               int32_t* ids_out, int64_t ids_cap, int64_t* out_offsets, int32_t* status);
static int encode_batch_pipelined(hutk_ctx* c, const uint8_t* bytes, const int64_t* offsets, int64_t n_docs,
                                  int32_t* ids_out, int64_t ids_cap, int64_t* out_offsets, int32_t* status);

// Batches of at least this many bytes go through the chunked path that overlaps the H2D copy of chunk c+1,
// the kernels of chunk c and the D2H copy of chunk c-1 (worth it only when the copies dominate)
static const int64_t PIPE_MIN_BYTES = 48ll << 20;
static int64_t pipe_chunk_bytes(int64_t n_bytes) {  // an eighth of the batch, 16..64 MB; HUTK_PIPE_CHUNK_MB overrides
    const char* e = getenv("HUTK_PIPE_CHUNK_MB");
    if (e && atol(e) > 0) return (int64_t)atol(e) << 20;
    const int64_t lo = 16ll << 20, hi = 64ll << 20;
    return std::min(hi, std::max(lo, n_bytes / 8));
}

// Regex pre-token path (reference src/core.c:350-360, 372-378, 392-400, 498-500), host half.  The reference compiles the
// pattern (POSIX ERE) for every encode() call and takes, again and again, the LEFTMOST match at or after its cursor: the
// match is a word, what lies before it is dropped, an empty match moves the cursor one byte on (or ends the document at
// the end of the text).  This is libc's regexec in the process's locale, so it stays on the host -- with the same libc
// calls -- and yields two bitmaps over the batch's bytes: where a word or a dropped stretch begins, and which of those
// are dropped stretches.  Pretokenizer and merge loop then run on the GPU as for the hand-written splitter.
// A word over the reference's limit (core.c:402-407) ends its document: the rest becomes a dropped stretch.
// fbits / abits (a context with a prefix; else null): the first match of every document, and those of them whose document
// begins with a space (core.c:364-366: the prefix goes with the first match, in front of it or as a word of its own)
static int regex_bitmaps(const std::string& pattern, const uint8_t* bytes, const int64_t* offsets, int64_t n_docs,
                         std::vector<uint32_t>& wbits, std::vector<uint32_t>& gbits, std::vector<uint8_t>& too_large,
                         std::vector<uint32_t>* fbits, std::vector<uint32_t>* abits) {
    const int64_t n_bytes = offsets[n_docs];
    const size_t n_words = (size_t)(n_bytes / 32 + 40);  // (a tile reads the bits of its whole 1024-position window)
    wbits.assign(n_words, 0u);
    gbits.assign(n_words, 0u);
    if (fbits) { fbits->assign(n_words, 0u); abits->assign(n_words, 0u); }
    too_large.assign((size_t)(n_docs ? n_docs : 1), 0);
    auto set_bit = [](std::vector<uint32_t>& v, int64_t p) {
        __atomic_fetch_or(&v[(size_t)(p >> 5)], 1u << (p & 31), __ATOMIC_RELAXED);
    };
    set_bit(wbits, n_bytes);  // the end of the data closes the last word
    unsigned nt = std::thread::hardware_concurrency();
    nt = std::max(1u, std::min(nt ? nt : 1u, 64u));
    if ((int64_t)nt > n_docs) nt = (unsigned)std::max<int64_t>(1, n_docs);
    std::atomic<int> failed{0};
    auto work = [&](unsigned t) {
        regex_t re;  // one compiled pattern per thread: glibc serialises regexec() on a shared regex_t
        if (regcomp(&re, pattern.c_str(), REG_EXTENDED) != 0) { failed = 1; return; }
        std::string z;
        const int64_t d0 = n_docs * t / nt, d1 = n_docs * (t + 1) / nt;
        for (int64_t d = d0; d < d1; d++) {
            const int64_t base = offsets[d], len = offsets[d + 1] - base;
            if (len <= 0) continue;
            z.assign(reinterpret_cast<const char*>(bytes + base), (size_t)len);  // NUL-terminated copy
            int64_t pos = 0, covered = 0;
            bool first = true;
            auto gap_to = [&](int64_t upto) {  // [covered, upto) belongs to no word
                if (upto > covered) { set_bit(wbits, base + covered); set_bit(gbits, base + covered); }
            };
            while (pos < len) {
                regmatch_t m;
                if (regexec(&re, z.c_str() + pos, 1, &m, 0) != 0) break;
                const int64_t ws = pos + m.rm_so, wl = m.rm_eo - m.rm_so;
                if (wl == 0) {
                    if (ws >= len) break;
                    pos = ws + 1;
                    continue;
                }
                if (wl * 64 > 16ll * 1024 * 1024) { too_large[(size_t)d] = 1; break; }
                gap_to(ws);
                set_bit(wbits, base + ws);
                if (first && fbits) {
                    set_bit(*fbits, base + ws);
                    if (z[0] == ' ') set_bit(*abits, base + ws);
                }
                first = false;
                covered = pos = ws + wl;
            }
            gap_to(len);
        }
        regfree(&re);
    };
    std::vector<std::thread> th;
    for (unsigned t = 1; t < nt; t++) th.emplace_back(work, t);
    work(0);
    for (auto& x : th) x.join();
    return failed ? HUTK_E_VALUE : HUTK_OK;
}

static int encode_batch_regex(hutk_ctx* c, const uint8_t* bytes, const int64_t* offsets, int64_t n_docs,
                              int32_t* ids_out, int64_t ids_cap, int64_t* out_offsets, int32_t* status);

int hutk_ctx_set_pattern(hutk_ctx* c, const char* pattern) {
    if (!c) return set_err(HUTK_E_ARG, "ctx is NULL");
    // (encode calls on other threads read the pattern under the context's mutex, each peer under its own)
    auto assign = [&](const char* v) {
        {
            std::lock_guard<std::recursive_mutex> lock(c->mu);
            c->pattern = v;
        }
        for (hutk_ctx* p : c->peers) {
            std::lock_guard<std::recursive_mutex> lock(p->mu);
            p->pattern = v;
        }
    };
    if (!pattern) {
        assign("");
        return HUTK_OK;
    }
    regex_t re;
    if (!*pattern || regcomp(&re, pattern, REG_EXTENDED) != 0)
        return set_err(HUTK_E_VALUE, "Regex could not be compiled.");  // core.c:352-358
    regfree(&re);
    assign(pattern);
    return HUTK_OK;
}

// one context, one device
static int encode_batch_one(hutk_ctx* c, const uint8_t* bytes, const int64_t* offsets, int64_t n_docs,
                            int32_t* ids_out, int64_t ids_cap, int64_t* out_offsets, int32_t* status) {
    if (!c) return set_err(HUTK_E_ARG, "ctx is NULL");
    std::lock_guard<std::recursive_mutex> lock(c->mu);  // (the staging buffers are the context's too)
    if (!c->host_only && !c->pattern.empty())
        return encode_batch_regex(c, bytes, offsets, n_docs, ids_out, ids_cap, out_offsets, status);
    if (!c->host_only && offsets && out_offsets && n_docs > 0 && offsets[0] == 0 &&
        offsets[n_docs] >= PIPE_MIN_BYTES && !getenv("HUTK_NO_PIPELINE")) {
        return encode_batch_pipelined(c, bytes, offsets, n_docs, ids_out, ids_cap, out_offsets, status);
    }
    return encode_batch_simple(c, bytes, offsets, n_docs, ids_out, ids_cap, out_offsets, status);
}

// Several devices in one process (SURVEY section 8(b) `device_mask`; the reference's batch_encode spreads documents over
// host threads with a DP on their lengths, lib.c:779-794): runs of whole documents with about the same number of BYTES,
// one per device, encoded side by side; run k's ids land in page-locked memory of their own and are copied behind
// those of runs 0..k-1 once these are counted (run 0 writes in place).  Same ids, offsets and status as on one device.
static int64_t multi_min_bytes() {
    const char* e = getenv("HUTK_MULTI_MIN_BYTES");
    return e ? atoll(e) : (4ll << 20);  // below this a second device's copies cost more than they save
}

static int encode_batch_multi(hutk_ctx* c, const uint8_t* bytes, const int64_t* offsets, int64_t n_docs,
                              int32_t* ids_out, int64_t ids_cap, int64_t* out_offsets, int32_t* status) {
    const int n_dev = 1 + (int)c->peers.size();
    const int64_t n_bytes = offsets[n_docs];
    const int64_t need = hutk_ids_capacity(c, n_bytes, n_docs) - 1;
    if (ids_cap < need) return set_err(HUTK_E_CAPACITY, "ids_cap is below hutk_ids_capacity()");
    // cut k: the first document that starts at or after k/n of the bytes
    std::vector<int64_t> lo((size_t)n_dev + 1, n_docs);
    lo[0] = 0;
    for (int k = 1; k < n_dev; k++) {
        const int64_t target = n_bytes / n_dev * k;
        lo[k] = std::lower_bound(offsets, offsets + n_docs + 1, target) - offsets;
        if (lo[k] > n_docs) lo[k] = n_docs;
        if (lo[k] < lo[k - 1]) lo[k] = lo[k - 1];
    }
    struct Run {
        std::vector<int64_t> offs, oo;
        int32_t* ids = nullptr;
        int64_t cap = 0;
        int rc = HUTK_OK;
        std::string msg;
    };
    std::vector<Run> runs((size_t)n_dev);
    for (int k = 0; k < n_dev; k++) {
        Run& r = runs[(size_t)k];
        const int64_t a = lo[k], b = lo[k + 1], nd = b - a;
        r.offs.resize((size_t)nd + 1);
        for (int64_t i = 0; i <= nd; i++) r.offs[(size_t)i] = offsets[a + i] - offsets[a];
        r.oo.assign((size_t)nd + 1, 0);
        hutk_ctx* ck = k ? c->peers[(size_t)k - 1] : c;
        r.cap = hutk_ids_capacity(ck, r.offs[(size_t)nd], nd) - 1;
        if (k == 0) {
            r.ids = ids_out;
        } else {
            hutk_ctx::PeerBuf& pb = c->peer_ids[(size_t)k - 1];
            if (pb.cap < (size_t)r.cap + 1) {
                if (pb.p) (void)hipHostFree(pb.p);
                pb.p = nullptr;
                pb.cap = 0;
                const size_t want = (size_t)r.cap + (size_t)r.cap / 8 + 64;
                HIP_TRY(hipSetDevice(ck->device));
                HIP_TRY(hipHostMalloc((void**)&pb.p, want * 4, hipHostMallocPortable));
                pb.cap = want;
            }
            r.ids = pb.p;
        }
    }
    auto encode_run = [&](int k) {
        Run& r = runs[(size_t)k];
        hutk_ctx* ck = k ? c->peers[(size_t)k - 1] : c;
        const int64_t a = lo[k], nd = lo[k + 1] - a;
        r.rc = encode_batch_one(ck, bytes + offsets[a], r.offs.data(), nd, r.ids, r.cap, r.oo.data(),
                                status ? status + a : nullptr);
        if (r.rc) r.msg = g_err;  // (the message is the thread's)
    };
    {
        std::vector<std::thread> th;
        for (int k = 1; k < n_dev; k++) th.emplace_back(encode_run, k);
        encode_run(0);
        for (auto& t : th) t.join();
    }
    for (int k = 0; k < n_dev; k++)
        if (runs[(size_t)k].rc) return set_err(runs[(size_t)k].rc, runs[(size_t)k].msg);
    std::vector<int64_t> base((size_t)n_dev + 1, 0);
    for (int k = 0; k < n_dev; k++) base[(size_t)k + 1] = base[(size_t)k] + runs[(size_t)k].oo.back();
    if (base[(size_t)n_dev] > ids_cap) return set_err(HUTK_E_CAPACITY, "ids_cap too small");
    auto place_run = [&](int k) {
        const Run& r = runs[(size_t)k];
        const int64_t a = lo[k], nd = lo[k + 1] - a, b0 = base[(size_t)k];
        if (k && r.oo.back()) memcpy(ids_out + b0, r.ids, (size_t)r.oo.back() * 4);
        for (int64_t i = 0; i < nd; i++) out_offsets[a + i] = b0 + r.oo[(size_t)i];
    };
    {
        std::vector<std::thread> th;
        for (int k = 1; k < n_dev; k++) th.emplace_back(place_run, k);
        place_run(0);
        for (auto& t : th) t.join();
    }
    out_offsets[n_docs] = base[(size_t)n_dev];
    return HUTK_OK;
}

int hutk_encode_batch(hutk_ctx* c, const uint8_t* bytes, const int64_t* offsets, int64_t n_docs,
                      int32_t* ids_out, int64_t ids_cap, int64_t* out_offsets, int32_t* status) {
    if (!c) return set_err(HUTK_E_ARG, "ctx is NULL");
    std::lock_guard<std::recursive_mutex> lock(c->mu);
    if (!c->peers.empty() && offsets && out_offsets && bytes && ids_out && n_docs >= 1 + (int64_t)c->peers.size() &&
        offsets[0] == 0 && offsets[n_docs] >= multi_min_bytes()) {
        bool sane = true;
        for (int64_t i = 0; i < n_docs && sane; i++) sane = offsets[i + 1] >= offsets[i];
        if (sane) return encode_batch_multi(c, bytes, offsets, n_docs, ids_out, ids_cap, out_offsets, status);
    }
    return encode_batch_one(c, bytes, offsets, n_docs, ids_out, ids_cap, out_offsets, status);
}

int hutk_decode_batch_device(hutk_ctx* c, const int32_t* d_ids, const int64_t* d_id_offsets, int64_t n_docs,
                             int64_t n_ids, uint8_t* d_bytes_out, int64_t bytes_cap, int64_t* d_out_offsets,
                             int32_t* d_status, int32_t* d_err, void* hip_stream) {
    if (!c) return set_err(HUTK_E_ARG, "ctx is NULL");
    if (c->host_only) return set_err(HUTK_E_DEVICE, "host-only context: no device to decode on");
    if (n_docs < 0 || n_ids < 0 || !d_id_offsets || !d_out_offsets || (n_ids > 0 && !d_ids))
        return set_err(HUTK_E_ARG, "bad argument");
    std::lock_guard<std::recursive_mutex> lock(c->mu);
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t s = hip_stream ? (hipStream_t)hip_stream : c->stream;
    if (c->busy_valid) HIP_TRY(hipStreamWaitEvent(s, c->ev_busy, 0));
    struct BusyMark {
        hutk_ctx* c; hipStream_t s;
        ~BusyMark() { if (hipEventRecord(c->ev_busy, s) == hipSuccess) c->busy_valid = true; }
    } busy_mark{c, s};
    const int64_t tile = dec_tile_ids();
    const int64_t n_tiles = (n_ids + tile - 1) / tile;
    if (n_tiles > 0x7FFFFFFFll) return set_err(HUTK_E_ARG, "batch too large");
    HIP_TRY(c->dw_first.reserve((size_t)(n_ids / 32 + 4)));
    HIP_TRY(c->dw_state.reserve((size_t)n_tiles + 8));
    HIP_TRY(c->dw_tfd.reserve((size_t)n_tiles + 1));
    HIP_TRY(c->w_err.reserve(1));
    DecArgs D{};
    D.ids = d_ids;
    D.id_offsets = d_id_offsets;
    D.n_docs = n_docs;
    D.n_ids = n_ids;
    D.n_tiles = n_tiles;
    D.bytes_out = d_bytes_out;
    D.bytes_cap = bytes_cap;
    D.out_offsets = d_out_offsets;
    D.status = d_status;
    D.err = d_err ? d_err : c->w_err.p;
    D.first_bits = c->dw_first.p;
    D.tile_state = c->dw_state.p;
    D.tile_first_doc = c->dw_tfd.p;
    D.help_after = getenv("HUTK_DEC_HELP_AFTER") ? (uint32_t)atol(getenv("HUTK_DEC_HELP_AFTER")) : (1u << 14);  // (0: tests of the fallback)
    HIP_TRY(hipMemsetAsync(D.err, 0, 4, s));
    const bool strip = c->dec.sent != nullptr;  // the first-token bitmap is only needed to strip a prefix
    if (strip) HIP_TRY(hipMemsetAsync(D.first_bits, 0, (size_t)(n_ids / 32 + 4) * 4, s));
    else D.first_bits = nullptr;
    if (d_status && n_docs) HIP_TRY(hipMemsetAsync(d_status, 0, (size_t)n_docs * 4, s));
    if (n_tiles == 0) {
        HIP_TRY(hipMemsetAsync(d_out_offsets, 0, (size_t)(n_docs + 1) * 8, s));
        return HUTK_OK;
    }
    if (strip) launch_dec_mark(D, s);
    HIP_TRY(hipMemsetAsync(D.tile_state, 0, (size_t)n_tiles * 8, s));
    launch_dec(c->dec, D, s);
    HIP_TRY(hipGetLastError());
    return HUTK_OK;
}

int hutk_decode_batch(hutk_ctx* c, const int32_t* ids, const int64_t* id_offsets, int64_t n_docs, uint8_t* bytes_out,
                      int64_t bytes_cap, int64_t* out_offsets, int32_t* status) {
    if (!c) return set_err(HUTK_E_ARG, "ctx is NULL");
    std::lock_guard<std::recursive_mutex> lock(c->mu);
    if (c->host_only) return set_err(HUTK_E_DEVICE, "host-only context: no device to decode on");
    if (n_docs < 0 || !id_offsets || !out_offsets) return set_err(HUTK_E_ARG, "bad argument");
    if (id_offsets[0] != 0) return set_err(HUTK_E_ARG, "id_offsets[0] must be 0");
    for (int64_t i = 0; i < n_docs; i++)
        if (id_offsets[i + 1] < id_offsets[i]) return set_err(HUTK_E_ARG, "id_offsets must not decrease");
    const int64_t n_ids = id_offsets[n_docs];
    if (n_ids > 0 && !ids) return set_err(HUTK_E_ARG, "bad argument");
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t s = c->stream;
    HIP_TRY(c->ds_ids.reserve((size_t)n_ids + 16));
    HIP_TRY(c->ds_offs.reserve((size_t)n_docs + 1));
    HIP_TRY(c->ds_oo.reserve((size_t)n_docs + 1));
    HIP_TRY(c->ds_status.reserve((size_t)n_docs + 1));
    HIP_TRY(c->w_err.reserve(1));
    if (bytes_out && bytes_cap > 0) HIP_TRY(c->ds_bytes.reserve((size_t)bytes_cap + 16));
    if (n_ids) HIP_TRY(hipMemcpyAsync(c->ds_ids.p, ids, (size_t)n_ids * 4, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(c->ds_offs.p, id_offsets, (size_t)(n_docs + 1) * 8, hipMemcpyHostToDevice, s));
    int rc = hutk_decode_batch_device(c, c->ds_ids.p, c->ds_offs.p, n_docs, n_ids, bytes_out ? c->ds_bytes.p : nullptr,
                                      bytes_cap, c->ds_oo.p, c->ds_status.p, c->w_err.p, s);
    if (rc) return rc;
    int32_t err = 0;
    HIP_TRY(hipMemcpyAsync(out_offsets, c->ds_oo.p, (size_t)(n_docs + 1) * 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(&err, c->w_err.p, 4, hipMemcpyDeviceToHost, s));
    if (status && n_docs) HIP_TRY(hipMemcpyAsync(status, c->ds_status.p, (size_t)n_docs * 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    if (bytes_out && err == HUTK_OK && out_offsets[n_docs] > 0)
        HIP_TRY(hipMemcpy(bytes_out, c->ds_bytes.p, (size_t)out_offsets[n_docs], hipMemcpyDeviceToHost));
    switch (err) {
        case HUTK_OK: return HUTK_OK;
        case HUTK_E_VALUE: return set_err(err, "Element must be non-negative and less than vocab size.");
        case HUTK_E_UNSUPPORTED:
            return set_err(err, "a token cannot be decoded on its own (id without a unique key, or a token that ends "
                                "inside a special value or a character)");
        case HUTK_E_CAPACITY: return set_err(err, "bytes_cap too small");
        default: return set_err(err, "device-side failure");
    }
}

// NUMA node of the current HIP device (its PCI function's numa_node in sysfs), or -1
static int device_numa_node() {
    int dev = 0;
    char bus[32] = {0};
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetPCIBusId(bus, (int)sizeof bus, dev) != hipSuccess) {
        (void)hipGetLastError();
        return -1;
    }
    for (char* q = bus; *q; q++) *q = (char)tolower((unsigned char)*q);
    const std::string path = std::string("/sys/bus/pci/devices/") + bus + "/numa_node";
    FILE* f = fopen(path.c_str(), "r");
    if (!f) return -1;
    int node = -1;
    if (fscanf(f, "%d", &node) != 1) node = -1;
    fclose(f);
    return node;
}

// Page-locked memory ON THE GPU'S NUMA NODE: on a two-socket host a buffer on the far node sends both copy directions
// over the link between the sockets.  The calling thread's memory policy prefers the device's node while the pages are
// allocated and pinned -- only when the thread runs under the default policy: a process started under numactl --membind
// or --interleave keeps what it was given -- and the default policy is what is put back afterwards
// (get_mempolicy(2) / set_mempolicy(2); HUTK_HOST_ALLOC_NUMA=0: the policy is left alone).
void* hutk_host_alloc(size_t n_bytes) {
    void* p = nullptr;
    bool bound = false;
#if defined(__linux__) && defined(SYS_set_mempolicy) && defined(SYS_get_mempolicy)
    const char* e = getenv("HUTK_HOST_ALLOC_NUMA");
    const int node = (e && e[0] == '0') ? -1 : device_numa_node();
    if (node >= 0 && node < 64) {
        int mode = -1;
        unsigned long old_mask[16] = {0};
        const bool known = syscall(SYS_get_mempolicy, &mode, old_mask, 16ul * 8 * sizeof(unsigned long), nullptr, 0ul) == 0;
        if (known && mode == 0 /* MPOL_DEFAULT: nothing of the caller's to lose */) {
            unsigned long mask = 1ul << node;
            bound = syscall(SYS_set_mempolicy, 1 /* MPOL_PREFERRED */, &mask, 65ul) == 0;
        }
    }
#endif
    const hipError_t rc = hipHostMalloc(&p, n_bytes ? n_bytes : 1, hipHostMallocDefault);
#if defined(__linux__) && defined(SYS_set_mempolicy) && defined(SYS_get_mempolicy)
    if (bound) (void)syscall(SYS_set_mempolicy, 0 /* MPOL_DEFAULT: what it was */, nullptr, 0ul);
#endif
    if (rc != hipSuccess) return nullptr;
    return p;
}

void hutk_host_free(void* p) {
    if (p) (void)hipHostFree(p);
}

// Chunks of whole documents, double buffered.  Per chunk: bytes + offsets H2D on s_in; on the context's
// stream the offsets are rebased, the kernel sequence runs and the chunk's out_offsets are made absolute with
// a device-side running total; the out_offsets come back on s_out (their last entry places the ids in the
// caller's array), then the ids.  No per-document work on the host.
static int encode_batch_pipelined(hutk_ctx* c, const uint8_t* bytes, const int64_t* offsets, int64_t n_docs,
                                  int32_t* ids_out, int64_t ids_cap, int64_t* out_offsets, int32_t* status) {
    for (int64_t i = 0; i < n_docs; i++)
        if (offsets[i + 1] < offsets[i]) return set_err(HUTK_E_ARG, "offsets must not decrease");
    const int64_t n_bytes = offsets[n_docs];
    if (!bytes || !ids_out) return set_err(HUTK_E_ARG, "bad argument");
    if (ids_cap < hutk_ids_capacity(c, n_bytes, n_docs) - 1)
        return set_err(HUTK_E_CAPACITY, "ids_cap is below hutk_ids_capacity()");
    HIP_TRY(hipSetDevice(c->device));
    hutk_ctx::Pipe& P = c->pipe;
    if (!P.ready) {
        HIP_TRY(hipStreamCreateWithFlags(&P.s_in, hipStreamNonBlocking));
        HIP_TRY(hipStreamCreateWithFlags(&P.s_out, hipStreamNonBlocking));
        for (int b = 0; b < hutk_ctx::Pipe::NB; b++) {
            HIP_TRY(hipEventCreateWithFlags(&P.ev_in[b], hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&P.ev_comp[b], hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&P.ev_out[b], hipEventDisableTiming));
        }
        HIP_TRY(hipHostMalloc((void**)&P.h_small, 64, hipHostMallocDefault));
        P.ready = true;
    }
    // chunk boundaries (whole documents)
    const int64_t chunk_bytes = pipe_chunk_bytes(n_bytes);
    std::vector<int64_t> first;  // first document of each chunk, plus n_docs
    int64_t max_bytes = 0, max_docs = 0;
    for (int64_t d = 0; d < n_docs;) {
        first.push_back(d);
        int64_t e = d + 1;
        while (e < n_docs && offsets[e + 1] - offsets[d] <= chunk_bytes) e++;
        max_bytes = std::max(max_bytes, offsets[e] - offsets[d]);
        max_docs = std::max(max_docs, e - d);
        d = e;
    }
    first.push_back(n_docs);
    const int n_chunks = (int)first.size() - 1;
    const int64_t max_ids = hutk_ids_capacity(c, max_bytes, max_docs);
    constexpr int NB = hutk_ctx::Pipe::NB;
    for (int b = 0; b < NB; b++) {
        HIP_TRY(P.bytes[b].reserve((size_t)max_bytes + 64));
        HIP_TRY(P.offs[b].reserve((size_t)max_docs + 1));
        HIP_TRY(P.offs_abs[b].reserve((size_t)max_docs + 1));
        HIP_TRY(P.oo[b].reserve((size_t)max_docs + 1));
        HIP_TRY(P.ids[b].reserve((size_t)max_ids + 1));
        HIP_TRY(P.status[b].reserve((size_t)max_docs + 1));
        HIP_TRY(P.err[b].reserve(1));
    }
    HIP_TRY(P.base.reserve(1));
    {  // the workspace is grown once, for the largest chunk: growing it later would synchronise the device
        Workspace W{};
        const int rc = ensure_workspace(c, max_bytes, max_docs, (max_bytes + TILE_BYTES - 1) / TILE_BYTES, W);
        if (rc) return rc;
    }
    hipStream_t sc = c->stream;
    HIP_TRY(hipMemsetAsync(P.base.p, 0, 8, sc));
    int64_t base = 0;  // ids of the chunks finalised so far
    int dev_err = 0;
    const bool trace = getenv("HUTK_PIPE_TRACE") != nullptr;  // diagnostic: host-side time stamps per chunk on stderr
    const auto t_start = std::chrono::steady_clock::now();
    auto now_ms = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count(); };
    std::vector<double> tr;
    auto finalize = [&](int ch) -> int {  // chunk ch: results to the caller's arrays
        const int b = ch % NB;
        const int64_t d0 = first[ch], nd = first[ch + 1] - d0;
        HIP_TRY(hipStreamWaitEvent(P.s_out, P.ev_comp[b], 0));
        // first the two numbers the host needs (the chunk's error word and where its ids end), to page-locked memory;
        // the out_offsets themselves follow with the ids
        HIP_TRY(hipMemcpyAsync(P.h_small, P.err[b].p, 4, hipMemcpyDeviceToHost, P.s_out));
        HIP_TRY(hipMemcpyAsync(P.h_small + 1, P.oo[b].p + nd, 8, hipMemcpyDeviceToHost, P.s_out));
        if (trace) tr.push_back(now_ms());
        HIP_TRY(hipStreamSynchronize(P.s_out));
        if (trace) tr.push_back(now_ms());
        const int32_t err = (int32_t)P.h_small[0];
        const int64_t total = P.h_small[1] - base;  // the chunk's offsets are absolute already
        HIP_TRY(hipMemcpyAsync(out_offsets + d0, P.oo[b].p, (size_t)(nd + 1) * 8, hipMemcpyDeviceToHost, P.s_out));
        if (err && err != HUTK_E_WORD_TOO_LARGE && !dev_err) dev_err = err;  // (an over-long word is a note: k_cut has cut its document)
        if (base + total > ids_cap) return set_err(HUTK_E_CAPACITY, "ids_cap too small");
        if (total)
            HIP_TRY(hipMemcpyAsync(ids_out + base, P.ids[b].p, (size_t)total * 4, hipMemcpyDeviceToHost, P.s_out));
        if (status && nd)
            HIP_TRY(hipMemcpyAsync(status + d0, P.status[b].p, (size_t)nd * 4, hipMemcpyDeviceToHost, P.s_out));
        HIP_TRY(hipEventRecord(P.ev_out[b], P.s_out));
        base += total;
        return HUTK_OK;
    };
    for (int ch = 0; ch < n_chunks; ch++) {
        const int b = ch % NB;
        const int64_t d0 = first[ch], nd = first[ch + 1] - d0;
        const int64_t b0 = offsets[d0], nb = offsets[d0 + nd] - b0;
        if (trace) tr.push_back(now_ms());
        if (ch >= NB) HIP_TRY(hipEventSynchronize(P.ev_out[b]));  // buffers b are free again
        if (trace) tr.push_back(now_ms());
        if (nb) HIP_TRY(hipMemcpyAsync(P.bytes[b].p, bytes + b0, (size_t)nb, hipMemcpyHostToDevice, P.s_in));
        HIP_TRY(hipMemcpyAsync(P.offs_abs[b].p, offsets + d0, (size_t)(nd + 1) * 8, hipMemcpyHostToDevice, P.s_in));
        HIP_TRY(hipEventRecord(P.ev_in[b], P.s_in));
        HIP_TRY(hipStreamWaitEvent(sc, P.ev_in[b], 0));
        launch_rebase_offsets(P.offs_abs[b].p, P.offs[b].p, nd + 1, sc);  // relative to the chunk's first byte
        const int rc = hutk_encode_batch_device(c, P.bytes[b].p, P.offs[b].p, nd, nb, P.ids[b].p, max_ids, P.oo[b].p,
                                                P.status[b].p, P.err[b].p, sc);
        if (rc) {
            (void)hipDeviceSynchronize();
            return rc;
        }
        launch_add_base(P.oo[b].p, nd + 1, P.base.p, sc);  // out_offsets absolute; base moves on
        HIP_TRY(hipEventRecord(P.ev_comp[b], sc));
        if (ch >= 1) {
            const int frc = finalize(ch - 1);
            if (frc) {
                (void)hipDeviceSynchronize();
                return frc;
            }
        }
    }
    {
        const int frc = finalize(n_chunks - 1);
        if (frc) {
            (void)hipDeviceSynchronize();
            return frc;
        }
    }
    HIP_TRY(hipStreamSynchronize(P.s_out));
    if (trace) {
        fprintf(stderr, "pipe trace: %d chunks, end %.2f ms;", n_chunks, now_ms());
        for (double v : tr) fprintf(stderr, " %.2f", v);
        fprintf(stderr, "\n");
    }
    out_offsets[n_docs] = base;
    switch (dev_err) {
        case HUTK_OK: return HUTK_OK;
        case HUTK_E_NUL_BYTE: return set_err(dev_err, "a document contains a 0x00 byte");
        case HUTK_E_INVALID_UTF8: return set_err(dev_err, "text is not valid UTF-8 (non-byte-encoder mode)");
        case HUTK_E_CAPACITY: return set_err(dev_err, "ids_cap too small");
        default: return set_err(dev_err, "device-side failure");
    }
}

static int encode_batch_host(hutk_ctx* c, const uint8_t* bytes, const int64_t* offsets, int64_t n_docs,
                             int32_t* ids_out, int64_t ids_cap, int64_t* out_offsets, int32_t* status,
                             const std::vector<uint32_t>* wbits, const std::vector<uint32_t>* gbits,
                             const std::vector<uint32_t>* fbits = nullptr, const std::vector<uint32_t>* abits = nullptr);

static int encode_batch_simple(hutk_ctx* c, const uint8_t* bytes, const int64_t* offsets, int64_t n_docs,
                               int32_t* ids_out, int64_t ids_cap, int64_t* out_offsets, int32_t* status) {
    return encode_batch_host(c, bytes, offsets, n_docs, ids_out, ids_cap, out_offsets, status, nullptr, nullptr);
}

static int encode_batch_regex(hutk_ctx* c, const uint8_t* bytes, const int64_t* offsets, int64_t n_docs,
                              int32_t* ids_out, int64_t ids_cap, int64_t* out_offsets, int32_t* status) {
    if (n_docs < 0 || !offsets || !out_offsets) return set_err(HUTK_E_ARG, "bad argument");
    if (offsets[0] != 0) return set_err(HUTK_E_ARG, "offsets[0] must be 0");
    for (int64_t i = 0; i < n_docs; i++)
        if (offsets[i + 1] < offsets[i]) return set_err(HUTK_E_ARG, "offsets must not decrease");
    if (offsets[n_docs] > 0 && !bytes) return set_err(HUTK_E_ARG, "bad argument");
    for (int64_t i = 0; i < offsets[n_docs]; i++)  // (regexec would stop there; the packed interface refuses it anyway)
        if (!bytes[i]) return set_err(HUTK_E_NUL_BYTE, "a document contains a 0x00 byte");
    std::vector<uint32_t> wbits, gbits, fbits, abits;
    std::vector<uint8_t> too_large;
    const bool pfx = c->tab.has_prefix;
    int rc = regex_bitmaps(c->pattern, bytes, offsets, n_docs, wbits, gbits, too_large, pfx ? &fbits : nullptr, pfx ? &abits : nullptr);
    if (rc) return set_err(rc, "Regex could not be compiled.");
    rc = encode_batch_host(c, bytes, offsets, n_docs, ids_out, ids_cap, out_offsets, status, &wbits, &gbits,
                           pfx ? &fbits : nullptr, pfx ? &abits : nullptr);
    if (rc == HUTK_OK && status)
        for (int64_t d = 0; d < n_docs; d++)
            if (too_large[(size_t)d]) status[d] = HUTK_DOC_WORD_TOO_LARGE;
    return rc;
}

static int encode_batch_host(hutk_ctx* c, const uint8_t* bytes, const int64_t* offsets, int64_t n_docs,
                             int32_t* ids_out, int64_t ids_cap, int64_t* out_offsets, int32_t* status,
                             const std::vector<uint32_t>* wbits, const std::vector<uint32_t>* gbits,
                             const std::vector<uint32_t>* fbits, const std::vector<uint32_t>* abits) {
    if (!c) return set_err(HUTK_E_ARG, "ctx is NULL");
    if (c->host_only) return set_err(HUTK_E_DEVICE, "host-only context: no device to encode on");
    if (n_docs < 0 || !offsets || !out_offsets) return set_err(HUTK_E_ARG, "bad argument");
    if (offsets[0] != 0) return set_err(HUTK_E_ARG, "offsets[0] must be 0");
    for (int64_t i = 0; i < n_docs; i++)
        if (offsets[i + 1] < offsets[i]) return set_err(HUTK_E_ARG, "offsets must not decrease");
    const int64_t n_bytes = offsets[n_docs];
    if (n_bytes > 0 && (!bytes || !ids_out)) return set_err(HUTK_E_ARG, "bad argument");
    const int64_t need = hutk_ids_capacity(c, n_bytes, n_docs) - 1;
    if (ids_cap < need) return set_err(HUTK_E_CAPACITY, "ids_cap is below hutk_ids_capacity()");
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t s = c->stream;
    // Small batches (one sentence, a few documents) are bound by the number of copies and launches, not by their size:
    // offsets and bytes go up as ONE copy from a page-locked staging buffer, and out_offsets, error word, status and
    // ids come back as ONE copy, instead of two up and four down.
    if (!wbits && n_bytes > 0 && n_bytes <= SMALL_BYTES && n_docs <= SMALL_DOCS) {
        const size_t in_offs = 0, in_bytes = (((size_t)n_docs + 1) * 8 + 15) & ~(size_t)15;
        const size_t in_size = in_bytes + (size_t)n_bytes;
        const size_t o_oo = 0, o_err = ((size_t)n_docs + 1) * 8, o_st = o_err + 8, o_ids = (o_st + (size_t)n_docs * 4 + 15) & ~(size_t)15;
        const size_t out_size = o_ids + (size_t)need * 4;
        if (!c->small_host) {  // page-locked, mapped into the device's address space, coherent (the one-launch path polls a word of it)
            if (hipHostMalloc(&c->small_host, SMALL_HOST_BYTES, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) {
                (void)hipGetLastError();
                if (hipHostMalloc(&c->small_host, SMALL_HOST_BYTES, hipHostMallocDefault) != hipSuccess) c->small_host = nullptr;
            }
        }
        // at most four tiles: everything in one launch, input read from and results written to this buffer by the kernel itself
        static const bool one_shot_on = !(getenv("HUTK_ONE_SHOT") && atoi(getenv("HUTK_ONE_SHOT")) == 0);
        {
            BatchArgs probe{};
            probe.n_tiles = (n_bytes + TILE_BYTES - 1) / TILE_BYTES;
            constexpr size_t ONE_OUT = 64 * 1024, ONE_FLAG = SMALL_HOST_BYTES - 64;
            if (one_shot_on && c->small_host && one_shot_takes(c->dt, probe) && in_size <= ONE_OUT && ONE_OUT + out_size <= ONE_FLAG) {
                uint8_t* h = static_cast<uint8_t*>(c->small_host);
                memcpy(h + in_offs, offsets, ((size_t)n_docs + 1) * 8);
                memcpy(h + in_bytes, bytes, (size_t)n_bytes);
                uint8_t* ho = h + ONE_OUT;
                int rc = encode_one_shot(c, h + in_bytes, reinterpret_cast<const int64_t*>(h + in_offs), n_docs, n_bytes,
                                         reinterpret_cast<int32_t*>(ho + o_ids), need, reinterpret_cast<int64_t*>(ho + o_oo),
                                         reinterpret_cast<int32_t*>(ho + o_st), reinterpret_cast<int32_t*>(ho + o_err),
                                         reinterpret_cast<int32_t*>(h + ONE_FLAG));
                if (rc) return rc;
                int32_t err = 0;
                memcpy(&err, ho + o_err, 4);
                if (err == HUTK_OK) {
                    memcpy(out_offsets, ho + o_oo, ((size_t)n_docs + 1) * 8);
                    const int64_t total = out_offsets[n_docs];
                    if (total > ids_cap) return set_err(HUTK_E_CAPACITY, "ids_cap too small");
                    if (total) memcpy(ids_out, ho + o_ids, (size_t)total * 4);
                    if (status && n_docs) memcpy(status, ho + o_st, (size_t)n_docs * 4);
                    return HUTK_OK;
                }
                // (an error: the general path below reports it)
            }
        }
        if (c->small_host && in_size <= SMALL_HOST_BYTES && out_size <= SMALL_HOST_BYTES) {
            HIP_TRY(c->s_small_in.reserve(in_size + 64));
            HIP_TRY(c->s_small_out.reserve(out_size + 64));
            uint8_t* h = static_cast<uint8_t*>(c->small_host);
            memcpy(h + in_offs, offsets, ((size_t)n_docs + 1) * 8);
            memcpy(h + in_bytes, bytes, (size_t)n_bytes);
            HIP_TRY(hipMemcpyAsync(c->s_small_in.p, h, in_size, hipMemcpyHostToDevice, s));
            uint8_t* di = c->s_small_in.p;
            uint8_t* dout = c->s_small_out.p;
            int rc = encode_device_impl(c, di + in_bytes, reinterpret_cast<const int64_t*>(di + in_offs), n_docs, n_bytes,
                                        reinterpret_cast<int32_t*>(dout + o_ids), need, reinterpret_cast<int64_t*>(dout + o_oo),
                                        reinterpret_cast<int32_t*>(dout + o_st), reinterpret_cast<int32_t*>(dout + o_err), s,
                                        nullptr, nullptr);
            if (rc) return rc;
            HIP_TRY(hipMemcpyAsync(h, dout, out_size, hipMemcpyDeviceToHost, s));
            HIP_TRY(hipStreamSynchronize(s));
            int32_t err = 0;
            memcpy(&err, h + o_err, 4);
            if (err == HUTK_OK) {
                memcpy(out_offsets, h + o_oo, ((size_t)n_docs + 1) * 8);
                const int64_t total = out_offsets[n_docs];
                if (total > ids_cap) return set_err(HUTK_E_CAPACITY, "ids_cap too small");
                if (total) memcpy(ids_out, h + o_ids, (size_t)total * 4);
                if (status && n_docs) memcpy(status, h + o_st, (size_t)n_docs * 4);
                return HUTK_OK;
            }
            // (an error or a cut document: the general path below reports it)
        }
    }
    HIP_TRY(c->s_bytes.reserve((size_t)n_bytes + 64));
    HIP_TRY(c->s_offsets.reserve((size_t)n_docs + 1));
    HIP_TRY(c->s_out_offsets.reserve((size_t)n_docs + 1));
    HIP_TRY(c->s_ids.reserve((size_t)need + 1));
    HIP_TRY(c->s_status.reserve((size_t)n_docs + 1));
    HIP_TRY(c->w_err.reserve(1));
    if (n_bytes) HIP_TRY(hipMemcpyAsync(c->s_bytes.p, bytes, (size_t)n_bytes, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(c->s_offsets.p, offsets, (size_t)(n_docs + 1) * 8, hipMemcpyHostToDevice, s));
    const uint32_t *d_wb = nullptr, *d_gb = nullptr, *d_fb = nullptr, *d_ab = nullptr;
    if (wbits) {
        HIP_TRY(c->w_wbits.reserve(wbits->size()));
        HIP_TRY(c->w_gbits.reserve(gbits->size()));
        HIP_TRY(hipMemcpyAsync(c->w_wbits.p, wbits->data(), wbits->size() * 4, hipMemcpyHostToDevice, s));
        HIP_TRY(hipMemcpyAsync(c->w_gbits.p, gbits->data(), gbits->size() * 4, hipMemcpyHostToDevice, s));
        d_wb = c->w_wbits.p;
        d_gb = c->w_gbits.p;
        if (fbits) {
            HIP_TRY(c->w_fbits.reserve(fbits->size()));
            HIP_TRY(c->w_abits.reserve(abits->size()));
            HIP_TRY(hipMemcpyAsync(c->w_fbits.p, fbits->data(), fbits->size() * 4, hipMemcpyHostToDevice, s));
            HIP_TRY(hipMemcpyAsync(c->w_abits.p, abits->data(), abits->size() * 4, hipMemcpyHostToDevice, s));
            d_fb = c->w_fbits.p;
            d_ab = c->w_abits.p;
        }
    }
    int rc = encode_device_impl(c, c->s_bytes.p, c->s_offsets.p, n_docs, n_bytes, c->s_ids.p, need,
                                c->s_out_offsets.p, c->s_status.p, c->w_err.p, s, d_wb, d_gb, d_fb, d_ab);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(out_offsets, c->s_out_offsets.p, (size_t)(n_docs + 1) * 8, hipMemcpyDeviceToHost, s));
    int32_t err = 0;
    HIP_TRY(hipMemcpyAsync(&err, c->w_err.p, 4, hipMemcpyDeviceToHost, s));
    if (status && n_docs)
        HIP_TRY(hipMemcpyAsync(status, c->s_status.p, (size_t)n_docs * 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    int64_t total = out_offsets[n_docs];
    if (total > ids_cap) return set_err(HUTK_E_CAPACITY, "ids_cap too small");
    if (total) HIP_TRY(hipMemcpy(ids_out, c->s_ids.p, (size_t)total * 4, hipMemcpyDeviceToHost));
    // (HUTK_E_WORD_TOO_LARGE is a note: the reference ends a document at a word longer than 262144 bytes and reports
    // nothing, core.c:402-407, 503; k_cut has done that on the device and set the document's status)
    if (err == HUTK_E_WORD_TOO_LARGE) return HUTK_OK;
    switch (err) {
        case HUTK_OK: return HUTK_OK;
        case HUTK_E_NUL_BYTE: return set_err(err, "a document contains a 0x00 byte");
        case HUTK_E_INVALID_UTF8: return set_err(err, "text is not valid UTF-8 (non-byte-encoder mode)");
        case HUTK_E_CAPACITY: return set_err(err, "ids_cap too small");
        default: return set_err(err, "device-side failure");
    }
}

int hutk_encode(hutk_ctx* c, const uint8_t* text, int64_t len, int32_t* ids_out, int64_t ids_cap,
                int64_t* n_ids, int32_t* status) {
    if (!n_ids || len < 0) return set_err(HUTK_E_ARG, "bad argument");
    int64_t offsets[2] = {0, len};
    int64_t oo[2] = {0, 0};
    int32_t st = 0;
    const int rc = hutk_encode_batch(c, text, offsets, 1, ids_out, ids_cap, oo, &st);
    *n_ids = oo[1];
    if (status) *status = st;
    return rc;
}

}  // extern "C"
