// hutk_device.h -- device-side structures and the launch interface between
// hutk_api.cpp (host orchestration) and hutk_kernels.hip.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "hutk_internal.h"

namespace hutk {

// ---- geometry of the tile kernel -----------------------------------------
constexpr int TILE_BYTES = 960;    // input bytes owned by one wavefront
constexpr int TILE_THREADS = 64;   // 1 wavefront: no inter-wave barriers in the hot kernel
constexpr int LOOKBACK = 16;       // bytes staged before the tile (classification needs <= 8)
constexpr int HALO = 64;           // positions classified after the tile (a word may end there)
constexpr int TAIL = 16;           // look-ahead for the last halo positions
constexpr int WINDOW = LOOKBACK + TILE_BYTES + HALO + TAIL;  // 1056 staged bytes
constexpr int LANE_MAX_UNITS = 32;   // longest word a single lane merges (32-bit live/candidate masks)
constexpr int LANE_MAX_BYTES = 63;   // and its byte length (its end must be within 63 positions)
// ids of one tile's words sit at run[tile * RUN_STRIDE + (first word start) + k]: at most one id per byte of
// the words that START in the tile (the last may overhang by LANE_MAX_BYTES), plus up to RUN_EXTRA ids of
// prefix units / prefix-alone ids granted to document-first words (beyond that budget: exception path)
constexpr int RUN_EXTRA = 64;
constexpr int RUN_STRIDE = TILE_BYTES + 64 + RUN_EXTRA;
constexpr int EXC_LDS_UNITS = 1024;  // exception words up to this many units merge in LDS

// per-position codes produced by the classifier (parser.c:24-183 restated as a
// function of a +-3 byte neighbourhood)
enum : uint8_t { C_INTERIOR = 0, C_ALPHA = 1, C_DIGIT = 2, C_OTHER = 3, C_SPACE = 4, C_WS = 5, C_BAD = 6 };

// Slot of the whole-word table: ONE 16-byte load.  Key bytes as little-endian dwords, zero padded, and the symbol in
// what the key leaves free: 16-bit symbols (vocabularies below 65520 symbols): 14 key bytes, symbol in the top half of
// k[3]; otherwise 12 key bytes, symbol = k[3].  (Round 2 began with 20-byte slots for 16-byte keys: two L1 accesses per
// slot, and the kernel is bound by the number of gather accesses)
struct WordSlot { uint32_t k[4]; };
// ... and of its companion for the words that do not fit (15..28 bytes; 13..28 with 32-bit symbols): 28 key bytes, then
// the symbol: two 16-byte loads, which a lane that holds such a word issues INSTEAD of the two slot loads of the main
// table (same registers, same instructions: the companion sits behind the main table in one allocation and a lane only
// picks other offsets).
struct WordSlotLong { uint32_t k[7]; uint32_t sym; };
constexpr int WORD_KEY_BYTES_16 = 14, WORD_KEY_BYTES_32 = 12;  // (the companion: WORDL_KEY_BYTES, hutk_internal.h)

struct DevTables {
    const uint4* pair_buckets;  // two entries per bucket {w0, w1, w0, w1}, see hutk_internal.h
    uint32_t pair_shift;        // bucket = hash >> pair_shift
    const int32_t* sym_id;
    uint32_t n_vocab_sym, n_sym;
    const uint32_t* item_sym;    // [256]
    const uint8_t* item_direct;  // [256]
    const uint64_t* char_slots;
    uint32_t char_mask, char_shift;
    const uint32_t* prefix_syms;
    int32_t n_prefix;
    const int32_t* prefix_alone_ids;      // the prefix encoded as a word of its own: ids ...
    const uint32_t* prefix_alone_syms;    // ... and the same tokens as symbols
    int32_t n_prefix_alone;
    int32_t is_byte_encoder, has_prefix, rank_is_sym, ident_ids;
    // byte-encoder mode: merged symbol of the initial pair (byte b1, byte b2) at
    // [b1 << 8 | b2], stored as uint16 when sym16 else uint32 (SYM_NONE when unranked)
    const void* bytepair;
    int32_t sym16;  // every symbol < 0xFFF0: LDS arrays hold 16-bit symbols
    // whole-word table: raw word bytes of 2..14 (12) bytes, zero padded -> symbol of the one token the word encodes to.
    // Two-choice cuckoo, 16-byte slots (WordSlot), an empty slot is all zero (a key's first bytes never are); slot 1 = hash & mask,
    // slot 2 = word_slot2(hash, mask) (hutk_internal.h).  word_mask == 0: no table.
    const WordSlot* word_tab;
    uint32_t word_mask;
    uint32_t wordl_off;   // the companion for longer words (WordSlotLong) begins at word_tab[wordl_off]: ONE slot per word,
    uint32_t wordl_mask;  // word_hash_long & wordl_mask; wordl_mask == 0: none
    // the word splitter as an automaton (hutk_classify.h, namespace dfa): dfa::TABLE_BYTES of transition table, then
    // the 256-byte byte-class table; the same for every vocabulary, staged in LDS by k_tiles
    const uint4* split_dfa;
    // seam map (Tables::seam_hi): bit (y - 0xE0) of seam_hi[x] clear = no token can span the input bytes x | y, so a word
    // of its own starts at y (k_tiles, exc_word_end).  seam_on == 0: never split (regex path, HUTK_NO_SEAM=1)
    const uint32_t* seam_hi;
    int32_t seam_on;
    // second level (Tables::seam2_*): whole characters A | B where seam_hi says "may join"
    const uint32_t* seam2_bits;
    const uint32_t* seam2_part;
    uint32_t seam2_shift;
    uint32_t seam2_cats;
    int32_t seam2_on;
    // items with a replacement of several units, or of none (Tables::multi_bits): bit b of multi_bits[8]; the units of
    // item b are item_units[item_units_off[b] .. item_units_off[b + 1]).  has_multi == 0: every item is one unit.
    // unit_scale = most units one input item can become: exception words own unit_scale slots per byte.
    const uint32_t* multi_bits;
    const uint32_t* item_units_off;
    const uint32_t* item_units;
    int32_t has_multi, unit_scale;
};

// one word the tile kernel hands to the exception kernel
struct ExcRec {
    int64_t ws;        // global byte offset of the word
    int64_t tok_base;  // its ids start at exc_tok[tok_base] (written by the exception kernel;
                       // negative: the word was too large, -(offset in document)-1)
    int64_t out_pos;   // position of the word's first id in ids_out (written by the gather kernel)
    int32_t len;       // byte length, or -1 when the end lies beyond the staged window
    uint32_t wpos;     // ids the tile emitted before this word (tile-local)
    uint32_t cnt;      // ids of this word (written by the exception kernel)
    uint32_t tile;
};

struct Workspace {
    uint32_t* run;        // SYMBOLS of tile t's lane-path words at run[t*RUN_STRIDE + first_word + k]; k_gather maps to ids
    int32_t* exc_tok;     // [cap_bytes + pad] ids of exception words at their own byte offset (+ doc padding)
    uint32_t* exc_sym;    // [cap_bytes + pad] symbol array of exception words too long for LDS
    uint32_t* exc_mrg;    // [cap_bytes + pad] their pair array
    uint32_t* tile_count;      // [n_tiles] ids per tile (dense run + exception words)
    uint32_t* tile_dense;      // [n_tiles] ids in the dense run only
    uint32_t* tile_run_start;  // [n_tiles] tile-local offset of the run (= first word start)
    uint32_t* tile_exc_first;  // [n_tiles] index of the tile's first ExcRec
    uint32_t* tile_nexc;       // [n_tiles]
    int64_t* tile_first_doc;   // [n_tiles] first document whose offset is >= tile start - LOOKBACK
    int64_t* tile_base;        // [n_tiles + 1] exclusive scan of tile_count
    unsigned long long* scan_state;  // [n_scan_blocks] k_scan's decoupled look-back: flag (2 bits) | total or inclusive prefix
    int64_t n_scan_blocks;
    uint32_t* doc_tile_pos;    // [n_docs + 1] ids the owning tile emits before the document start
    ExcRec* exc;               // [cap_exc]
    uint32_t* exc_quad;        // [cap_exc] exception words of at most 128 units from the front (count: counters[4]), of 129..256 from the back (counters[11]): d_exc_group_fast<2>, <4> (or d_exc_quad, both as one list)
    uint32_t* exc_mid;         // [cap_exc] ... of 257..512 units from the front (counters[12]), of 513..1024 from the back (counters[13]): d_exc_group_fast<8>, <16>
    uint32_t* exc_wave;        // [cap_exc] ... the rest, one wavefront each in d_exc (count: counters[5])
    uint32_t* counters;        // [0] exception total, [1] tiles with exceptions (one 64-bit atomic claims both), [2] d_exc's work cursor, [3] d_exc_lane_fast<1>'s (k_exc_a), [8] [9] [14] [15] d_exc_group_fast<2>'s .. <16>'s (k_exc_b), [4] / [5] entries of exc_quad / exc_wave, [6] tiles without a start of the reference's own, [7] k_scan's ticket
    uint32_t* exc_tiles;       // [n_tiles] those tiles, in no particular order
    uint32_t* noreal_bits;     // [n_tiles / 32 + 1] bit t: tile t holds no word start of the reference's own (k_cut)
    uint32_t* tile_first_start;  // [n_tiles] first word start (seams and document starts included) among the tile's 1024 classified positions, 0xFFFF: none -- d_exc_ends finds the end of a word its tile could not see here
    uint32_t* tile_lastreal;   // [n_tiles] position of the tile's last such start | ids before it << 16 (written when none follows in the halo)
    int64_t cap_exc;
    int32_t pad_per_doc;       // extra exc_* slots per document (prefix units + prefix-alone ids)
    long long* prof;           // diagnostic: [n_tiles][10] clock64 stamps of k_tiles, or null
    // Which tile kernel encodes the batch when BOTH are enqueued (hutk_api.cpp, "auto"): k_pre samples the batch's bytes
    // (counters[10] = bytes >= 0xE0 among the SELECT_SAMPLE bytes it looks at in one tile of SELECT_BLOCK_STRIDE); a batch in which they are at
    // least one byte in SELECT_DENSE_DIV is "dense".  select: 0 = run; 1 = run only for a dense batch (k_ptiles); 2 = run
    // only for one that is not (k_tiles).  The kernel that is not chosen returns at once.
    int32_t select;
    uint8_t* one_in;    // k_tiles<..., ONE>: device memory for the batch's offsets and bytes (copied from the caller's host memory once)
    int32_t* one_flag;  // k_tiles<..., ONE>: page-locked host memory; 1 = the batch's ids and offsets are in the caller's buffers, 2 = exception words: the tail is still to run
};
constexpr int SELECT_SAMPLE = 16, SELECT_DENSE_DIV = 8, SELECT_BLOCK_STRIDE = 8;  // (k_pre: 256 tiles per workgroup)
__device__ __forceinline__ bool select_skips(const Workspace& W, int64_t n_tiles) {
    if (W.select == 0) return false;
    // (about one tile in SELECT_BLOCK_STRIDE is sampled; the batches both kernels take have thousands of tiles)
    const bool dense = (int64_t)W.counters[10] * SELECT_DENSE_DIV * SELECT_BLOCK_STRIDE >= n_tiles * SELECT_SAMPLE;
    return (W.select == 1) != dense;
}

struct BatchArgs {
    const uint8_t* bytes;
    const int64_t* offsets;
    int64_t n_docs, n_bytes, n_tiles;
    int32_t* ids_out;
    int64_t ids_cap;
    int64_t* out_offsets;
    int32_t* status;  // may be null
    int32_t* err;     // never null (workspace word when the caller passes none)
    // regex pre-token path (a pattern was given to initialize): bit p of word_bits = a word or a dropped stretch starts
    // at byte p (bits up to and including n_bytes, padded with zero words); bit p of gap_bits = what starts at p is a
    // stretch no match covers (no ids).  Both null: the hand-written splitter (the automaton in k_tiles).
    const uint32_t* word_bits;
    const uint32_t* gap_bits;
    // ... together with a prefix (core.c:364-366, 421-451: the prefix goes with the FIRST MATCH of a document): bit p of
    // first_bits = the word at byte p is its document's first match; of alone_bits = ... and the document begins with a
    // space, so the prefix is encoded as a word of its own in front of it.  Both null: the first word is the one at the
    // document's first byte.
    const uint32_t* first_bits;
    const uint32_t* alone_bits;
};

// ---- decode direction (hutk_decode.hip) ----
// One 8-byte entry per id.  Low byte of x = tag: 0..7 the token's output is that many bytes, held in the
// entry itself (bytes 1..7); DEC_TAG_LONG: x >> 8 = length, y = offset into the blob; DEC_TAG_BAD: the token
// cannot be decoded on its own (no key, several keys, or context dependent).
constexpr uint32_t DEC_TAG_LONG = 0x80u, DEC_TAG_BAD = 0xFFu, DEC_INLINE_MAX = 7u;
struct DecTables {
    const uint2* ent;     // [n] entry of a token that is not the first of its document
    const uint2* sent;    // [n] entry of a document's first token (prefix stripped), or null (no prefix: use ent)
    const uint8_t* blob;  // output bytes
    int64_t n;            // ids 0..n-1 are in range (number of vocabulary lines, lib.c:377)
};
struct DecArgs {
    const int32_t* ids;
    const int64_t* id_offsets;  // [n_docs + 1]
    int64_t n_docs, n_ids, n_tiles;
    uint8_t* bytes_out;         // null: sizes and offsets only
    int64_t bytes_cap;
    int64_t* out_offsets;       // [n_docs + 1]
    int32_t* status;            // may be null
    int32_t* err;
    uint32_t* first_bits;       // [n_ids / 32 + 2] bit i: token i is the first of a document
    unsigned long long* tile_state;  // [n_tiles] decoupled look-back: flag (2 bits) | total or inclusive prefix
    int64_t* tile_first_doc;    // [n_tiles] first document whose first token is at or after the tile's
    uint32_t help_after;        // look-back: polls of a predecessor's state before a tile adds that tile's bytes up itself
};
int64_t dec_tile_ids();
void launch_dec_mark(const DecArgs& d, hipStream_t s);
void launch_dec(const DecTables& t, const DecArgs& d, hipStream_t s);

// hutk_kernels.hip
void launch_pre(const BatchArgs& a, const Workspace& w, hipStream_t s);
void launch_rebase_offsets(const int64_t* in, int64_t* out, int64_t n, hipStream_t s);
void launch_add_base(int64_t* v, int64_t n, int64_t* base, hipStream_t s);  // v[i] += *base; *base = v[n-1]
void launch_tiles(const DevTables& t, const BatchArgs& a, const Workspace& w, hipStream_t s);
// hutk_ptiles.hip: the persistent form of the tile kernel (one workgroup per compute unit, no workgroup barrier; same
// outputs).  ptiles_takes: is this vocabulary shape / batch one it handles?
bool ptiles_takes(const DevTables& t, const BatchArgs& a);
void launch_ptiles(const DevTables& t, const BatchArgs& a, const Workspace& w, hipStream_t s);
void launch_exceptions(const DevTables& t, const BatchArgs& a, const Workspace& w, hipStream_t s);
void launch_scan(const BatchArgs& a, const Workspace& w, hipStream_t s);
int64_t scan_blocks(int64_t n_tiles);
// a batch of at most four tiles: the whole pipeline in ONE launch (Workspace::one_flag tells the host when and how it ended)
bool one_shot_takes(const DevTables& t, const BatchArgs& a);
void launch_one_shot(const DevTables& t, const BatchArgs& a, const Workspace& w, hipStream_t s);
// a batch of a few tiles: exception stages, scan and copy-out as one single-wavefront launch
bool small_tail(const BatchArgs& a);
void launch_tail_small(const DevTables& t, const BatchArgs& a, const Workspace& w, hipStream_t s);
// tile runs (+ exception words) -> ids_out, out_offsets: one launch
void launch_finish(const DevTables& t, const BatchArgs& a, const Workspace& w, hipStream_t s);
// documents with a word of more than MAX_WORD_BYTES bytes end in front of it (core.c:402-407): one workgroup, which
// returns at once unless k_tiles counted enough tiles without a word start
void launch_cut(const DevTables& t, const BatchArgs& a, const Workspace& w, hipStream_t s);
// one-off at context creation: merge a symbol sequence on the device, return ids
void launch_bpe_symbols(const DevTables& t, uint32_t* d_syms, int n, int32_t* d_ids_out,
                        int32_t* d_n_out, hipStream_t s);

}  // namespace hutk
