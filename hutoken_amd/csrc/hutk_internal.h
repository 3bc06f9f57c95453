// hutk_internal.h -- shared between the host loader, the C-ABI and the kernels.
#pragma once
#include "hutk_seam2.h"
#include <cstdint>
#include <string>
#include <vector>

#include "hutoken_amd.h"

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define HUTK_HD __host__ __device__ __forceinline__
#else
#define HUTK_HD inline
#endif

namespace hutk {

// ---- symbol space -------------------------------------------------------
// Every byte string that can be a live token during a merge is a "symbol":
// the distinct vocabulary keys (numbered in ascending id order) followed by the
// initial units that are not keys ("pseudo" symbols, id -1).  Symbols are 20-bit.
constexpr uint32_t SYM_BITS = 20;
constexpr uint32_t SYM_UNK = (1u << SYM_BITS) - 1;  // unit that is no symbol: never merges, id -1
constexpr uint32_t SYM_NONE = 0xFFFFFFFFu;          // "this pair has no rank"
constexpr uint64_t SLOT_EMPTY = ~0ull;
constexpr int WORDL_KEY_BYTES = 28;  // longest raw word the whole-word tables hold (the companion's key, hutk_device.h)

// Pair table: BUCKETS of two 8-byte entries, one 16-byte load per lookup.  Entry (two dwords):
//   w0 = left | (right & 0xFFF) << 20        w1 = right >> 12 | merged << 8 | filter nibble << 28
// so a probe compares 32-bit words only.  An empty entry is w0 = all ones, w1 = 0x0FFFFFFF (left = SYM_UNK is
// never a key; its merged field reads PAIR_ABSENT).  A pair lives in its first bucket, or -- when that was full
// when the table was built -- in its second one; then bit (t & 7) of the first bucket's 8-bit filter (the two
// nibbles) is set, and only lookups that miss the first bucket AND find their filter bit set go on to the
// second (a fraction of a per cent of the lookups that find nothing).  Pairs are placed in ascending order of
// the merged symbol, i.e. the frequent merges of a trained vocabulary get their first bucket.
constexpr uint32_t PAIR_ABSENT = 0xFFFFFu;  // merged field of an empty entry
static inline uint64_t pair_slot(uint32_t l, uint32_t r, uint32_t m) {
    const uint32_t w0 = l | ((r & 0xFFFu) << 20), w1 = (r >> 12) | (m << 8);
    return (uint64_t)w0 | ((uint64_t)w1 << 32);
}
constexpr uint64_t PAIR_EMPTY = 0x0FFFFFFFFFFFFFFFull;

// The same mixing on host (table build) and device (lookup): 24-bit multiplies (full rate on CDNA) fold
// (left, right) into one 32-bit word t.  First bucket = the top bits of t, filter bit = its low three bits,
// second bucket (rarely computed) = the top bits of t times an odd constant, made different from the first.
HUTK_HD uint32_t pair_mix(uint32_t l, uint32_t r) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __umul24(l, 0x9E3779u) ^ (__umul24(r, 0x85EBCBu) + 0x165667B1u);
#else
    return (uint32_t)((uint64_t)(l & 0xFFFFFFu) * 0x9E3779u) ^
           ((uint32_t)((uint64_t)(r & 0xFFFFFFu) * 0x85EBCBu) + 0x165667B1u);
#endif
}
HUTK_HD uint32_t pair_bucket1(uint32_t t, uint32_t shift) { return t >> shift; }
HUTK_HD uint32_t pair_bucket2(uint32_t t, uint32_t shift) {
    const uint32_t b = (t * 0x9E3779B1u) >> shift;
    return b == (t >> shift) ? b ^ 1u : b;
}
// whole-word table: 16 raw bytes (zero padded) as four dwords
HUTK_HD uint32_t word_hash(uint32_t k0, uint32_t k1, uint32_t k2, uint32_t k3) {
    uint32_t x = k0 ^ ((k1 << 13) | (k1 >> 19)) ^ ((k2 << 7) | (k2 >> 25)) ^ ((k3 << 21) | (k3 >> 11));
    x *= 0x9E3779B1u;
    return x ^ (x >> 15);
}
// ... and of a longer word (28 raw bytes, zero padded, as seven dwords): the same function with the upper dwords folded in
HUTK_HD uint32_t word_hash_long(uint32_t k0, uint32_t k1, uint32_t k2, uint32_t k3, uint32_t k4, uint32_t k5, uint32_t k6) {
    return word_hash(k0 ^ ((k4 << 5) | (k4 >> 27)), k1 ^ ((k5 << 11) | (k5 >> 21)), k2 ^ ((k6 << 17) | (k6 >> 15)), k3);
}
// second candidate slot from the same hash: an odd multiple of its upper bits away from the first (never the same slot)
HUTK_HD uint32_t word_slot2(uint32_t h, uint32_t mask) {
#if defined(__HIP_DEVICE_COMPILE__)
    return ((h & mask) ^ __umul24((h >> 15) | 1u, 0x5BD1u)) & mask;
#else
    return ((h & mask) ^ (((h >> 15) | 1u) * 0x5BD1u)) & mask;
#endif
}

// Two-choice cuckoo placement of n keys into `cap` (power of two) single-entry slots.  h1/h2 give the two
// slots of key i; on return where[i] is the slot it got.  False when some key could not be placed
// (the caller grows the table or, for an optional table, drops the key).
template <class H1, class H2>
static inline bool cuckoo_place(size_t n, uint32_t cap, H1 h1, H2 h2, std::vector<uint32_t>& where,
                                std::vector<uint32_t>* failed = nullptr) {
    const uint32_t NONE = 0xFFFFFFFFu;
    std::vector<uint32_t> owner(cap, NONE);
    where.assign(n, NONE);
    bool all = true;
    uint64_t rng = 0x9E3779B97F4A7C15ull;
    for (size_t i = 0; i < n; i++) {
        uint32_t cur = (uint32_t)i;
        uint32_t slot = h1(cur) & (cap - 1);
        bool placed = false;
        for (int kick = 0; kick < 2000; kick++) {
            const uint32_t a = h1(cur) & (cap - 1), b = h2(cur) & (cap - 1);
            if (owner[a] == NONE) slot = a;
            else if (owner[b] == NONE) slot = b;
            else {
                // evict: take the slot we did not just come from, or a random one
                rng = rng * 6364136223846793005ull + 1442695040888963407ull;
                slot = (slot == a) ? b : (slot == b) ? a : ((rng >> 33) & 1u ? a : b);
            }
            const uint32_t prev = owner[slot];
            owner[slot] = cur;
            where[cur] = slot;
            if (prev == NONE) { placed = true; break; }
            where[prev] = NONE;
            cur = prev;
        }
        if (!placed) {
            // `cur` is homeless: the walk did not converge
            all = false;
            if (!failed) return false;
            failed->push_back(cur);
        }
    }
    return all;
}
HUTK_HD uint32_t char_hash(uint32_t packed) {
    uint32_t h = packed * 0x9E3779B1u;
    h ^= h >> 16;
    return h * 0x85EBCA6Bu;
}
// the two candidate slots of a character: the top bits of its hash, and an odd multiple of its low bits away from them
HUTK_HD uint32_t char_slot1(uint32_t h, uint32_t shift) { return h >> shift; }
HUTK_HD uint32_t char_slot2(uint32_t h, uint32_t mask) { return ((h >> 7) ^ ((h & 0x7Fu) * 0x2F1Bu) ^ 0x5A5Au) & mask; }

// reference limit: 64 * word length must fit the 16 MiB arena (core.c:27-28, 402-407)
constexpr int64_t MAX_WORD_BYTES = 262144;

struct Tables {
    // vocabulary facts
    int64_t n_keys = 0;         // distinct keys loaded (duplicates collapsed, last id wins)
    uint32_t n_vocab_sym = 0;   // keys with id != -1
    uint32_t n_sym = 0;         // + pseudo symbols
    bool rank_is_sym = false;   // ids strictly increase with the symbol index
    bool ident_ids = false;     // id(sym) == sym for every vocabulary symbol
    std::vector<int32_t> sym_id;  // [n_sym]

    // (left, right) -> merged symbol: buckets of two entries (see above); pair_slots.size() == 2 << (32 - pair_shift)
    std::vector<uint64_t> pair_slots;
    uint32_t pair_shift = 0;   // bucket = hash >> pair_shift
    int64_t n_pairs = 0;
    int64_t n_pairs_second = 0;  // pairs that live in their second bucket

    // initial symbol of a source item
    //   byte-encoder mode: item = input byte
    //   otherwise:         item = UTF-8 character, indexed by its lead byte when
    //                      the lead byte has a replacement or is ASCII
    uint32_t item_sym[256];
    uint8_t item_direct[256];  // 1: item_sym valid for this (lead) byte
    // A special-character replacement of several units (or of none) makes its item "multi": bit b of multi_bits; the
    // units of every item are item_units[item_units_off[b] .. item_units_off[b + 1]) (one entry, item_sym[b], for an
    // ordinary item).  A word with a multi item is an exception word: d_exc expands it.  max_units_per_item scales the
    // id capacity and the exception arrays (SURVEY section 8 b).
    uint32_t multi_bits[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    uint32_t item_units_off[257] = {0};
    std::vector<uint32_t> item_units;
    uint32_t max_units_per_item = 1;
    bool has_multi = false;
    // non-byte mode: multi-byte character (packed little-endian) -> symbol
    std::vector<uint64_t> char_slots;  // [packed:32][sym:32], SLOT_EMPTY
    uint32_t char_mask = 0;
    uint32_t char_shift = 0;

    // byte-encoder mode: merged symbol of the initial pair (byte b1, byte b2) at
    // [b1 << 8 | b2]; the 16-bit form is used when every symbol is < 0xFFF0
    bool sym16 = false;
    std::vector<uint32_t> bytepair32;
    std::vector<uint16_t> bytepair16;

    // byte-encoder mode without prefix: vocabulary keys expressed in raw input bytes
    // (<= 16), candidates for the whole-word table.  A candidate enters the table only
    // after the device pipeline itself has encoded it to exactly its own id.
    std::vector<uint8_t> cand_bytes;   // concatenated
    std::vector<uint32_t> cand_off;    // [n+1]
    std::vector<uint32_t> cand_sym;    // [n]

    // Seam map: bit (y - 0xE0) of seam_hi[x] = some merge can join a token that ends with input byte x to one that
    // begins with input byte y (y >= 0xE0: the lead bytes of three- and four-byte characters).  A clear bit is a place
    // where no token can ever span x | y: the word's encoding is the concatenation of the encodings of its two sides, and
    // the tile kernel starts a word of its own at y (hutk_loader.cpp, seam_from_pairs).
    uint32_t seam_hi[256] = {0};
    bool seam_on = false;
    // Second level, consulted where seam_hi says "may join" and both sides of the boundary are whole three-byte characters A | B
    // (hutk_loader.cpp, seam2_build): a merge can join across A | B only if seam2_part[last byte of A] has B's lead byte's bit
    // (entries of which no more than those two bytes is known) or the hashed set seam2_bits holds one of the keys (A, B),
    // (A, B's first two bytes), (A, B's lead byte), (A's last byte, B), (A's last byte, B's first two bytes) -- hutk_seam2.h:
    // entries whose left side ends with a whole character, whose right side begins with one or IS the two-byte prefix of
    // one (byte-level BPE on CJK text learns such tokens: a character + the prefix that sixty-four others share).  A false
    // positive costs a cut, nothing else.  For vocabularies whose merges cover every (last byte, lead byte) pair but
    // join only the character pairs of their frequent words (trained on CJK text).
    std::vector<uint32_t> seam2_bits;
    uint32_t seam2_part[256] = {0};
    uint32_t seam2_shift = 0;
    uint32_t seam2_cats = 0;  // kinds of keys the set holds, seam2_cat_bit(kind of the left part, kind of the right part)
    bool seam2_on = false;

    bool is_byte_encoder = false;
    bool has_prefix = false;
    std::vector<uint32_t> prefix_syms;        // units of the prefix when it is prepended to a word
    std::vector<uint32_t> prefix_alone_syms;  // units of the prefix encoded as its own word
    // id-keyed merge path (a merges file was loaded): tables built by build_id_tables; the prefix encoded as
    // a word of its own is then already merged on the host (it stays on the string path, core.c:421-446)
    bool id_path = false;
    bool prefix_alone_final = false;
    std::vector<int32_t> prefix_alone_ids;

    // ---- decode direction (core.c:513-581, pretokenizer.c:197-296) ----
    // Per id < dec_n (the number of vocabulary lines, lib.c:377): its output bytes when the token is decoded
    // on its own.  That is exact in any context when the scan of pretokenizer_decode, started at the token's
    // first byte, ends exactly at its last one and never stops on a proper prefix of a longer special value.
    // Tokens for which this cannot be promised carry a flag and make their document fail loudly.
    int64_t dec_n = 0;
    std::vector<uint8_t> dec_blob;    // output bytes of all tokens, then (prefix mode) of their stripped forms
    std::vector<uint32_t> dec_off;    // [dec_n] into dec_blob
    std::vector<uint16_t> dec_len;    // [dec_n] output length; DEC_BAD for ids that cannot be decoded
    std::vector<uint32_t> dec_soff;   // prefix mode: the same with the prefix removed from the token's front
    std::vector<uint16_t> dec_slen;   //   DEC_NOSTRIP: the token does not start with the prefix
    std::vector<uint8_t> dec_flag;    // DEC_F_* bits
};
constexpr uint16_t DEC_BAD = 0xFFFFu, DEC_NOSTRIP = 0xFFFEu;
enum : uint8_t {
    DEC_F_HOLE = 1,        // no key has this id (undefined behaviour in the reference)
    DEC_F_AMBIGUOUS = 2,   // several keys have this id (the reference keeps whichever its hash map yields last)
    DEC_F_CONTEXT = 4,     // decoding depends on the neighbouring tokens (see above)
    DEC_F_PFX_PARTIAL = 8  // the token is a proper prefix of the prefix string: stripping may span tokens
};

struct LoadError {
    int code = HUTK_OK;
    std::string msg;
};

// hutk_loader.cpp: parse both files with the reference's quirks and build Tables.
// merges_path: optional merges file (id-keyed merge path, reference lib.c:573-663)
LoadError load_tables(const char* vocab_path, const char* special_path, const char* prefix,
                      bool is_byte_encoder, const char* merges_path, Tables& out);

}  // namespace hutk
