// hutk_loader.cpp -- host side of hutk_ctx_create: reads huToken's vocab and
// special-character files with the reference's observable quirks and turns them
// into the device tables of hutk_internal.h.
//
// Reference behaviour followed here (paths into the reference tree):
//   vocab file      src/lib.c:243-388, src/helper.c:82-128
//   special file    src/lib.c:460-571
//   unit rule       src/core.c:35-55 (literal "<0x..>" or one UTF-8 character,
//                   src/pretokenizer.c:14-28)
//   pretokenizer    src/pretokenizer.c:102-168
//   rank of a pair  src/core.c:700-722: the vocabulary id of the concatenation
#include <algorithm>
#include <functional>
#include <cerrno>
#include <climits>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <unordered_map>
#include <vector>

#include "hutk_internal.h"

namespace hutk {
namespace {

struct FileCloser {
    void operator()(FILE* f) const {
        if (f) fclose(f);
    }
};
using File = std::unique_ptr<FILE, FileCloser>;

LoadError fail(int code, const char* msg) { return LoadError{code, msg}; }

// helper.c:82-128: "0x" + two characters handed to strtol(.., 16); anything else
// is skipped.  false when more than 2047 bytes would be produced.
bool hex_field_to_bytes(const char* s, const char* end, std::string& out) {
    out.clear();
    while (s < end) {
        if (s[0] == '0' && s + 1 < end && s[1] == 'x') {
            s += 2;
            if (s < end && s + 1 < end) {
                const char two[3] = {s[0], s[1], 0};
                if (out.size() >= 2047) return false;
                out.push_back((char)(unsigned)strtol(two, nullptr, 16));
            }
            s += 2;
        } else {
            ++s;
        }
    }
    return true;
}

LoadError read_vocab(const char* path, std::unordered_map<std::string, int32_t>& vocab, int64_t* n_lines) {
    File f(fopen(path, "rb"));
    if (!f) return fail(HUTK_E_FILE_NOT_FOUND, "Could not open vocab file.");
    std::string data;
    char buf[1 << 16];
    size_t got;
    while ((got = fread(buf, 1, sizeof buf, f.get())) > 0) data.append(buf, got);
    size_t pos = 0, lines = 0;
    std::string key;
    while (pos < data.size()) {
        const void* nl = memchr(data.data() + pos, '\n', data.size() - pos);
        if (!nl) break;  // lib.c:264-289: an unterminated final line is dropped silently
        size_t eol = (const char*)nl - data.data();
        // C-string view of the line: it ends at an embedded NUL if there is one
        std::string line(data.data() + pos, eol + 1 - pos);
        line.resize(strlen(line.c_str()));
        pos = eol + 1;
        const char* sep = strstr(line.c_str(), " == ");
        if (!sep) return fail(HUTK_E_VALUE, "Invalid format in vocab file.");
        const char* vs = sep + 4;
        char* endp = nullptr;
        errno = 0;
        long v = strtol(vs, &endp, 10);
        if (endp == vs)
            return fail(HUTK_E_VALUE, "Invalid vocab format: could not parse integer value.");
        if (errno == ERANGE || v > INT_MAX || v < INT_MIN)
            return fail(HUTK_E_VALUE, "Integer value in vocab file is out of range.");
        bool ok = hex_field_to_bytes(line.c_str(), sep, key);
        // strdup(): the key ends at its first 0x00; an empty key is an error (lib.c:345-357)
        if (ok) key.resize(strlen(key.c_str()));
        if (!ok || key.empty()) return fail(HUTK_E_VALUE, "Failed to convert hex string to ASCII.");
        vocab[key] = (int32_t)v;  // a repeated key keeps its last id (hashmap.c:208-214)
        ++lines;
    }
    if (lines == 0) return fail(HUTK_E_VALUE, "Vocab file is empty.");
    *n_lines = (int64_t)lines;
    return {};
}

LoadError read_special(const char* path, std::string special[256], bool has[256], int seq[256]) {
    int n_loaded = 0;
    File f(fopen(path, "r"));
    if (!f) return fail(HUTK_E_FILE_NOT_FOUND, "Could not open special characters file.");
    char buf[32];  // lib.c:483: records are 31-character fgets() chunks
    while (fgets(buf, sizeof buf, f.get())) {
        const char* sep = strstr(buf, " == ");
        if (!sep) return fail(HUTK_E_VALUE, "Invalid format in special character file.");
        char* endp = nullptr;
        errno = 0;
        long idx = strtol(buf, &endp, 10);
        if (endp == buf)
            return fail(HUTK_E_VALUE, "Invalid vocab format: could not parse integer value.");
        // lib.c:516 lets 256 through and then writes out of bounds; rejected here
        if (errno == ERANGE || idx > 255 || idx < 0)
            return fail(HUTK_E_VALUE, "Integer value in vocab file is out of range.");
        const char* vs = sep + 4;
        size_t vlen = strlen(vs);
        // the chunk's last character is dropped whatever it is (lib.c:527-531)
        if (vlen < 2) return fail(HUTK_E_VALUE, "Failed to convert hex string to ASCII.");
        special[idx].assign(vs, vlen - 1);
        has[idx] = true;
        seq[idx] = ++n_loaded;  // decode: among equal values the one loaded last names the byte
    }
    return {};
}

int lead_len(unsigned char b) {  // pretokenizer.c:14-28
    if ((b & 0x80) == 0x00) return 1;
    if ((b & 0xE0) == 0xC0) return 2;
    if ((b & 0xF0) == 0xE0) return 3;
    if ((b & 0xF8) == 0xF0) return 4;
    return 1;
}

bool is_hex(unsigned char c) {
    return (c >= '0' && c <= '9') || (c >= 'a' && c <= 'f') || (c >= 'A' && c <= 'F');
}

// core.c:35-47 on a bounded string.  Returns the literal's length, 0 when s does
// not start one, and sets *dangling when s ran out while still matching.
size_t hex_literal_len(const std::string& s, size_t p, bool* dangling) {
    *dangling = false;
    static const char pat0[] = "<0";
    for (size_t k = 0; k < 2; k++) {
        if (p + k >= s.size()) { *dangling = true; return 0; }
        if (s[p + k] != pat0[k]) return 0;
    }
    if (p + 2 >= s.size()) { *dangling = true; return 0; }
    if (s[p + 2] != 'x' && s[p + 2] != 'X') return 0;
    size_t q = p + 3;
    while (q < s.size() && is_hex((unsigned char)s[q])) q++;
    if (q >= s.size()) { *dangling = true; return 0; }
    return s[q] == '>' ? q - p + 1 : 0;
}

// Splits s into units.  false when a unit would run past the end of s or when s
// ends in a proper prefix of a "<0x..>" literal: in both cases the reference's
// split of the word would depend on the NEXT item, which the per-item device
// tables cannot express.
bool split_units(const std::string& s, bool hex_units, std::vector<std::string>& out) {
    out.clear();
    size_t p = 0;
    while (p < s.size()) {
        size_t l = 0;
        if (hex_units) {
            bool dangling = false;
            l = hex_literal_len(s, p, &dangling);
            if (dangling) return false;
        }
        if (!l) l = (size_t)lead_len((unsigned char)s[p]);
        if (p + l > s.size()) return false;
        out.push_back(s.substr(p, l));
        p += l;
    }
    // a '<' later in s must not leave a dangling literal prefix either
    if (hex_units)
        for (size_t q = 0; q < s.size(); q++)
            if (s[q] == '<') {
                bool dangling = false;
                (void)hex_literal_len(s, q, &dangling);
                if (dangling) return false;
            }
    return true;
}

bool unit_shaped(const std::string& s) {
    if (s.empty()) return false;
    bool dangling = false;
    size_t l = hex_literal_len(s, 0, &dangling);
    if (l == s.size()) return true;
    return (size_t)lead_len((unsigned char)s[0]) == s.size();
}

uint32_t pack_char(const std::string& s) {
    uint32_t v = 0;
    for (size_t i = 0; i < s.size() && i < 4; i++) v |= (uint32_t)(unsigned char)s[i] << (8 * i);
    return v;
}

uint32_t pow2_at_least(uint64_t n) {
    uint32_t c = 16;
    while (c < n) c <<= 1;
    return c;
}


// item -> its units (Tables::item_units): one unit = item_sym[b] unless the replacement has several, or none
void finish_item_units(Tables& T, const std::vector<uint32_t> multi[256]) {
    T.item_units.clear();
    T.max_units_per_item = 1;
    for (int b = 0; b < 256; b++) {
        T.item_units_off[b] = (uint32_t)T.item_units.size();
        if ((T.multi_bits[b >> 5] >> (b & 31)) & 1u) {
            T.item_units.insert(T.item_units.end(), multi[b].begin(), multi[b].end());
            T.max_units_per_item = std::max<uint32_t>(T.max_units_per_item, (uint32_t)multi[b].size());
            T.has_multi = true;
        } else {
            T.item_units.push_back(T.item_sym[b]);
        }
    }
    T.item_units_off[256] = (uint32_t)T.item_units.size();
}

// (left, right) -> merged entries into the bucketed pair table (hutk_internal.h)
LoadError finish_pair_table(Tables& T, const std::vector<uint64_t>& entries_in) {
    T.n_sym = (uint32_t)T.sym_id.size();
    if (T.n_sym >= SYM_UNK)
        return fail(HUTK_E_UNSUPPORTED, "vocabulary too large for 20-bit symbols");
    T.n_pairs = (int64_t)entries_in.size();
    auto left_of = [](uint64_t e) { return (uint32_t)e & 0xFFFFFu; };
    auto right_of = [](uint64_t e) { return ((uint32_t)e >> 20) | (((uint32_t)(e >> 32) & 0xFFu) << 12); };
    // ascending merged symbol: the low ranks (frequent merges) are placed first and get their first bucket
    std::vector<uint64_t> entries(entries_in);
    std::stable_sort(entries.begin(), entries.end(),
                     [](uint64_t x, uint64_t y) { return (x >> 40) < (y >> 40); });
    uint32_t n_buckets = pow2_at_least(entries.size() * 5 / 4 + 16);  // two entries each: load <= 0.4
    if (const char* e = getenv("HUTK_PAIR_BUCKETS_LOG2")) n_buckets = 1u << atoi(e);  // measurement: footprint against probes
    for (int attempt = 0;; attempt++) {
        int64_t n_second_extra = 0;
        uint32_t shift = 32;
        while ((1ull << (32 - shift)) < n_buckets) shift--;
        std::vector<uint64_t> slots((size_t)n_buckets * 2, PAIR_EMPTY);
        auto put = [&](uint32_t b, uint64_t e) -> bool {
            for (int k = 0; k < 2; k++)
                if (slots[2 * (size_t)b + k] == PAIR_EMPTY) { slots[2 * (size_t)b + k] = e; return true; }
            return false;
        };
        std::vector<uint64_t> second;
        auto mix_of = [&](uint64_t e) { return pair_mix(left_of(e), right_of(e)); };
        for (uint64_t e : entries)
            if (!put(pair_bucket1(mix_of(e), shift), e)) second.push_back(e);
        // A pair that found its first bucket full goes to its second one.  When that is full too, a resident moves
        // to the other of ITS two buckets, and so on for a few steps (a bucket never loses an entry on the way, so
        // the first bucket of a pair that lives in its second one stays full).  Every pair stays within its two
        // buckets: a lookup never reads more than two.
        std::function<bool(uint32_t, uint64_t, int)> into = [&](uint32_t b, uint64_t e, int depth) -> bool {
            if (put(b, e)) return true;
            if (depth == 0) return false;
            for (int k = 0; k < 2; k++) {
                const uint64_t x = slots[2 * (size_t)b + k];
                const uint32_t tx = mix_of(x), x1 = pair_bucket1(tx, shift), x2 = pair_bucket2(tx, shift);
                const uint32_t alt = x1 == b ? x2 : x1;
                slots[2 * (size_t)b + k] = e;
                if (into(alt, x, depth - 1)) return true;
                slots[2 * (size_t)b + k] = x;
            }
            return false;
        };
        bool ok = true;
        for (uint64_t e : second) {
            const uint32_t t = mix_of(e);
            if (!into(pair_bucket2(t, shift), e, 5) && !into(pair_bucket1(t, shift), e, 5)) { ok = false; break; }
        }
        if (ok) {  // filter bits: one per pair that ended up in its second bucket, in the top nibbles of its first bucket's entries
            for (size_t i = 0; i < slots.size(); i++) {
                const uint64_t x = slots[i];
                if (x == PAIR_EMPTY) continue;
                const uint32_t tx = mix_of(x & 0x0FFFFFFFFFFFFFFFull), x1 = pair_bucket1(tx, shift);
                if ((uint32_t)(i >> 1) == x1) continue;
                const uint32_t f = tx & 7u;
                slots[2 * (size_t)x1 + (f >> 2)] |= (uint64_t)(1u << (f & 3u)) << 60;
                n_second_extra++;
            }
        }
        if (ok) {
            T.pair_shift = shift;
            T.pair_slots.swap(slots);
            T.n_pairs_second = n_second_extra;
            break;
        }
        if (attempt == 5) return fail(HUTK_E_UNSUPPORTED, "pair table could not be built");
        n_buckets *= 2;
    }
    T.sym16 = T.n_sym < 0xFFF0u;
    return {};
}

uint32_t host_pair_lookup(const Tables& T, uint32_t l, uint32_t r) {
    const uint32_t k0 = l | ((r & 0xFFFu) << 20), k1 = r >> 12;
    const uint32_t t = pair_mix(l, r);
    for (uint32_t b : {pair_bucket1(t, T.pair_shift), pair_bucket2(t, T.pair_shift)})
        for (int k = 0; k < 2; k++) {
            const uint64_t sl = T.pair_slots[2 * (size_t)b + k];
            if ((uint32_t)sl == k0 && (((uint32_t)(sl >> 32)) & 0xFFu) == k1) {
                const uint32_t m = (uint32_t)(sl >> 40) & 0xFFFFFu;
                return m == PAIR_ABSENT ? SYM_NONE : m;
            }
        }
    return SYM_NONE;
}

// byte-encoder mode: direct table for the initial (byte, byte) pairs
void finish_bytepair(Tables& T) {
    if (!T.is_byte_encoder) return;
    if (T.sym16) T.bytepair16.assign(65536, 0xFFFFu); else T.bytepair32.assign(65536, SYM_NONE);
    for (int b1 = 1; b1 < 256; b1++)
        for (int b2 = 1; b2 < 256; b2++) {
            const uint32_t m = host_pair_lookup(T, T.item_sym[b1], T.item_sym[b2]);
            if (m == SYM_NONE) continue;
            if (T.sym16) T.bytepair16[(b1 << 8) | b2] = (uint16_t)m; else T.bytepair32[(b1 << 8) | b2] = m;
        }
}

// non-byte mode: multi-byte character (packed) -> symbol
void finish_char_table(Tables& T, std::vector<std::pair<uint32_t, uint32_t>>& chars) {
    if (T.is_byte_encoder) {
        T.char_slots.assign(16, SLOT_EMPTY);
        T.char_mask = 15;
        T.char_shift = 28;
        return;
    }
    // two-choice cuckoo like the other tables: slot char_slot1(h) or char_slot2(h), both loaded together by a lookup, no
    // dependent probe sequence (a wavefront pays for the slowest of its 64 lanes)
    uint32_t cap = pow2_at_least(chars.size() * 5 / 2 + 16);
    for (int attempt = 0;; attempt++) {
        uint32_t lg = 0;
        while ((1u << lg) < cap) lg++;
        T.char_mask = cap - 1;
        T.char_shift = 32 - lg;
        std::vector<uint32_t> where;
        const bool ok = cuckoo_place(chars.size(), cap,
                                     [&](uint32_t i) { return char_slot1(char_hash(chars[i].first), T.char_shift); },
                                     [&](uint32_t i) { return char_slot2(char_hash(chars[i].first), T.char_mask); }, where);
        if (ok || attempt == 6) {
            T.char_slots.assign(cap, SLOT_EMPTY);
            for (size_t i = 0; i < chars.size(); i++)
                if (where[i] != 0xFFFFFFFFu) T.char_slots[where[i]] = ((uint64_t)chars[i].first << 32) | chars[i].second;
            break;  // (a character left out after seven doublings would read as unknown: cannot happen at this load)
        }
        cap *= 2;
    }
}


// ------------------------------------------------------------------------
// Seam map (Tables::seam_hi).  A merge joins two tokens L and R only if (L, R) is an entry of the pair table
// (core.c:700-722: the concatenation is a key; core.c:724-736: the pair is a rule), and every token is a run of whole
// units as the pretokenizer makes them from the input items.  So two adjacent input bytes x | y with a unit boundary
// between them can end up inside one token only if some pair entry has an L whose last unit can come from an input
// item ending in x and an R whose first unit can come from an item beginning with y.  Where no entry does, the word's
// encoding is the concatenation of the encodings of its two sides -- whatever the rest of the word looks like -- and
// k_tiles may treat y as the start of a word of its own.  Kept for y >= 0xE0 only (the lead bytes of three- and
// four-byte characters: in text, runs of CJK characters and emoji); everything else is "may merge".
// INVARIANT the analysis rests on: a unit is made from ONE input item (byte, or character outside byte-encoder mode) or is
// one unit of that item's replacement string -- a "<0xNN>" literal is a unit only as (part of) a replacement, never
// assembled from raw '<', '0', 'x' items: the hand-written splitter never leaves those three in one word
// (SURVEY section 8 a-6), and the regex path, whose words are whatever the pattern says, runs without seams (seam_on
// is cleared there).  tests/test_seam_cpu.py fuzzes the map against the oracle, raw "<0xNN>" text beside CJK included.
// ------------------------------------------------------------------------
struct RawEnds {
    uint64_t last[4] = {0, 0, 0, 0};  // input bytes the last unit can come from (its item's last byte)
    uint32_t first_hi = 0;            // input bytes >= 0xE0 the first unit's item can begin with, bit y - 0xE0
    void all() { last[0] = last[1] = last[2] = last[3] = ~0ull; first_hi = ~0u; }
    bool join(const RawEnds& o, bool take_first, bool take_last) {
        bool ch = false;
        if (take_first && (first_hi | o.first_hi) != first_hi) { first_hi |= o.first_hi; ch = true; }
        if (take_last)
            for (int q = 0; q < 4; q++)
                if ((last[q] | o.last[q]) != last[q]) { last[q] |= o.last[q]; ch = true; }
        return ch;
    }
};
struct UnitEnds {
    std::unordered_map<std::string, RawEnds> of;  // unit string -> the input items it can come from
    bool is_byte_encoder = false;
    const bool* has_special = nullptr;
    void add_item(int b, const std::string& unit, bool whole_char_replaced) {
        RawEnds& e = of[unit];
        if (b >= 0xE0) e.first_hi |= 1u << (b - 0xE0);
        // (outside byte-encoder mode) the item is a whole character led by b.  Well-formed, it ends in a continuation byte --
        // but the reference's pretokenizer takes utf8_char_length(b) bytes whatever they are (pretokenizer.c:130-153), so the
        // byte in front of the seam can be anything when the text is malformed: every byte counts as a possible last one.
        // (The kernels raise HUTK_E_INVALID_UTF8 for such text anyway; the seam map does not lean on that.)
        if (whole_char_replaced) e.last[0] = e.last[1] = e.last[2] = e.last[3] = ~0ull;
        else e.last[b >> 6] |= 1ull << (b & 63);
    }
    RawEnds get(const std::string& u) const {
        RawEnds e;
        auto it = of.find(u);
        if (it != of.end()) e = it->second;
        // outside byte-encoder mode a multi-byte character without replacement is its own unit
        if (!is_byte_encoder && u.size() >= 2 && (size_t)lead_len((unsigned char)u[0]) == u.size() &&
            !has_special[(unsigned char)u[0]]) {
            const unsigned b0 = (unsigned char)u[0], bl = (unsigned char)u.back();
            if (b0 >= 0xE0) e.first_hi |= 1u << (b0 - 0xE0);
            e.last[bl >> 6] |= 1ull << (bl & 63);
        }
        return e;
    }
};
UnitEnds unit_ends_of_items(const std::string special[256], const bool has_special[256], bool is_byte_encoder,
                            bool hex_units) {
    UnitEnds U;
    U.is_byte_encoder = is_byte_encoder;
    U.has_special = has_special;
    std::vector<std::string> us;
    for (int b = 1; b < 256; b++) {
        if (!is_byte_encoder && ((b >= 0x80 && b < 0xC0) || b >= 0xF8)) continue;
        std::string s;
        if (has_special[b]) s = special[b];
        else if (is_byte_encoder && b >= 0x80) {
            s.push_back((char)(0xC0 | (b >> 6)));
            s.push_back((char)(0x80 | (b & 0x3F)));
        } else if (b < 0x80) s.push_back((char)b);
        else continue;
        // (a replacement of several units: the seam sees its first and its last one)
        if (!split_units(s, hex_units, us) || us.empty()) { us.assign(1, s); }
        const bool whole = !is_byte_encoder && b >= 0xC0;
        if (us.size() == 1) U.add_item(b, us[0], whole);
        else {
            RawEnds& f = U.of[us.front()];
            if (b >= 0xE0) f.first_hi |= 1u << (b - 0xE0);
            RawEnds& l = U.of[us.back()];
            if (whole) l.last[0] = l.last[1] = l.last[2] = l.last[3] = ~0ull;  // (as in add_item)
            else l.last[b >> 6] |= 1ull << (b & 63);
        }
    }
    return U;
}
void seam_from_pairs(Tables& T, const std::vector<uint64_t>& entries, const std::vector<RawEnds>& ends) {
    for (int x = 0; x < 256; x++) T.seam_hi[x] = 0;
    for (uint64_t e : entries) {
        const uint32_t l = (uint32_t)e & 0xFFFFFu, r = ((uint32_t)e >> 20) | (((uint32_t)(e >> 32) & 0xFFu) << 12);
        if (l >= ends.size() || r >= ends.size()) continue;
        const uint32_t f = ends[r].first_hi;
        if (!f) continue;
        for (int q = 0; q < 4; q++)
            for (uint64_t m = ends[l].last[q]; m; m &= m - 1) T.seam_hi[64 * q + __builtin_ctzll(m)] |= f;
    }
    T.seam_on = true;
}

// Second level (Tables::seam2_*), byte-encoder mode, string-keyed path.  raw[s]: the input bytes of vocabulary symbol s when
// every unit of its key is produced by exactly one input byte (empty: not known).  An entry (L, R) whose L ends with a whole
// three-byte character A and whose R begins with one, B, can join two input bytes x | y only where the text reads A | B
// there: it goes into the hashed set of character pairs.  Every other entry stays what it is in seam_hi: a (last byte,
// lead byte) pair, in seam2_part.  So where the text reads A | B, both well-formed, a token across the boundary needs
// seam2_part[A's last byte] bit (B's lead byte) or the pair (A, B) in the set.
void seam2_build(Tables& T, const std::vector<uint64_t>& entries, const std::vector<RawEnds>& ends,
                 const std::vector<std::string>& raw) {
    T.seam2_on = false;
    for (int x = 0; x < 256; x++) T.seam2_part[x] = 0;
    T.seam2_bits.clear();
    if (!T.seam_on || !T.is_byte_encoder || T.has_multi || T.has_prefix) return;
    auto tail3 = [](const std::string& r, uint32_t* out) -> bool {
        if (r.size() < 3) return false;
        const size_t n = r.size();
        const uint32_t c = (uint32_t)(unsigned char)r[n - 3] | ((uint32_t)(unsigned char)r[n - 2] << 8) | ((uint32_t)(unsigned char)r[n - 1] << 16);
        *out = c;
        return seam2_char3(c);
    };
    auto head3 = [](const std::string& r, uint32_t* out) -> bool {
        if (r.size() < 3) return false;
        const uint32_t c = (uint32_t)(unsigned char)r[0] | ((uint32_t)(unsigned char)r[1] << 8) | ((uint32_t)(unsigned char)r[2] << 16);
        *out = c;
        return seam2_char3(c);
    };
    struct Key { uint32_t a, b, ka, kb; };
    std::vector<Key> keys;
    size_t n_full = 0;
    for (uint64_t e : entries) {
        const uint32_t l = (uint32_t)e & 0xFFFFFu, r = ((uint32_t)e >> 20) | (((uint32_t)(e >> 32) & 0xFFu) << 12);
        if (l >= ends.size() || r >= ends.size()) continue;
        const uint32_t f = ends[r].first_hi;
        if (!f) continue;  // (R never begins with a byte >= 0xE0: no seam is asked about)
        uint32_t a3 = 0, b3 = 0;
        const bool la = l < raw.size() && tail3(raw[l], &a3);   // L ends with the whole character a3
        uint32_t kb = 1;                                        // what is known of R's beginning
        if (r < raw.size() && head3(raw[r], &b3)) kb = 3;
        else if (r < raw.size() && raw[r].size() == 2 && ((unsigned char)raw[r][0] & 0xF0u) == 0xE0u && ((unsigned char)raw[r][1] & 0xC0u) == 0x80u) {
            kb = 2;
            b3 = (uint32_t)(unsigned char)raw[r][0] | ((uint32_t)(unsigned char)raw[r][1] << 8);
        }
        if (!la && kb == 1) {  // two bytes: the first level's kind of entry
            for (int q = 0; q < 4; q++)
                for (uint64_t m = ends[l].last[q]; m; m &= m - 1) T.seam2_part[64 * q + __builtin_ctzll(m)] |= f;
            if (getenv("HUTK_DEBUG_SEAM2") && atoi(getenv("HUTK_DEBUG_SEAM2")) > 1) {
                fprintf(stderr, "  part entry: L =");
                if (l < raw.size()) for (unsigned char ch : raw[l]) fprintf(stderr, " %02x", ch);
                fprintf(stderr, " | R =");
                if (r < raw.size()) for (unsigned char ch : raw[r]) fprintf(stderr, " %02x", ch);
                fprintf(stderr, "  (l %u r %u first_hi %x)\n", l, r, f);
            }
            continue;
        }
        n_full += la && kb == 3;
        // the left parts: the character, or every byte L can end with; the right parts: the character / prefix, or every lead byte
        std::vector<uint32_t> as, bs;
        if (la) as.push_back(a3);
        else
            for (int q = 0; q < 4; q++)
                for (uint64_t m = ends[l].last[q]; m; m &= m - 1) as.push_back((uint32_t)(64 * q + __builtin_ctzll(m)));
        if (kb != 1) bs.push_back(b3);
        else
            for (uint32_t m = f; m; m &= m - 1) bs.push_back(0xE0u + (uint32_t)__builtin_ctz(m));
        for (uint32_t a : as)
            for (uint32_t b : bs) keys.push_back({a, b, la ? 3u : 1u, kb});
    }
    if (getenv("HUTK_DEBUG_SEAM2")) {
        size_t known = 0, part_bits = 0;
        for (auto& r : raw) known += !r.empty();
        for (int x = 0; x < 256; x++) part_bits += __builtin_popcount(T.seam2_part[x]);
        fprintf(stderr, "seam2: %zu entries, %zu of whole characters, %zu keys, %zu of %zu symbols with known bytes, %zu bits in the part map\n",
                entries.size(), n_full, keys.size(), known, raw.size(), part_bits);
    }
    if (keys.empty()) return;  // (nothing the first level does not say already)
    uint32_t lg = 16;
    while (lg < 24 && ((size_t)1 << lg) < keys.size() * 32) lg++;
    T.seam2_shift = 32 - lg;
    T.seam2_bits.assign(((size_t)1 << lg) / 32, 0u);
    T.seam2_cats = 0;
    for (auto& k : keys) {
        const uint32_t h = seam2_hash(k.a, k.b, k.ka, k.kb) >> T.seam2_shift;
        T.seam2_bits[h >> 5] |= 1u << (h & 31);
        T.seam2_cats |= seam2_cat_bit(k.ka, k.kb);
    }
    T.seam2_on = true;
}

// ------------------------------------------------------------------------
// id-keyed merge path: a merges file was given (src/lib.c:573-663, src/core.c:211-337, 457-477)
// ------------------------------------------------------------------------
struct MergeRule {
    int32_t l, r, rank, m;
};

// lib.c:573-663 with its quirks: lines come through fgets with a 10000-byte buffer (a longer line arrives in
// pieces); a line counts when it does not start with '#' and holds a space; strtok(" ") takes the first two
// space-separated fields; a rule whose left, right or concatenation is no vocabulary key is skipped and takes
// no rank; an equal (left id, right id) replaces the earlier rule, rank included.  *has_merges tells whether
// the reference would have created its merges map at all (it switches paths on that, core.c:457).
LoadError read_merges(const char* path, const std::unordered_map<std::string, int32_t>& vocab, bool* has_merges,
                      std::vector<MergeRule>& alive) {
    File f(fopen(path, "r"));
    if (!f) return fail(HUTK_E_FILE_NOT_FOUND, "Could not open merges file.");
    std::vector<char> line(10000);  // MAX_LINE_LENGTH, lib.c:71
    size_t line_count = 0;
    while (fgets(line.data(), (int)line.size(), f.get()))
        if (line[0] != '#' && strchr(line.data(), ' ') != nullptr) line_count++;
    *has_merges = line_count > 0;
    alive.clear();
    if (!line_count) return {};
    rewind(f.get());
    std::unordered_map<uint64_t, size_t> at;  // (left id, right id) -> index into alive
    size_t idx = 0;
    int32_t rank = 0;
    while (fgets(line.data(), (int)line.size(), f.get()) && idx < line_count) {
        if (line[0] == '#') continue;
        char* p = line.data();
        p[strcspn(p, "\r\n")] = 0;
        auto next_field = [&](char*& q) -> char* {  // strtok(.., " ")
            while (*q == ' ') q++;
            if (!*q) return nullptr;
            char* start = q;
            while (*q && *q != ' ') q++;
            if (*q) *q++ = 0;
            return start;
        };
        char* left = next_field(p);
        char* right = left ? next_field(p) : nullptr;
        if (!left || !right) continue;
        const std::string ls(left), rs(right);
        auto li = vocab.find(ls), ri = vocab.find(rs), mi = vocab.find(ls + rs);
        if (li == vocab.end() || ri == vocab.end() || mi == vocab.end()) continue;
        MergeRule nr{li->second, ri->second, rank++, mi->second};
        idx++;
        const uint64_t key = ((uint64_t)(uint32_t)nr.l << 32) | (uint32_t)nr.r;
        auto it = at.find(key);
        if (it == at.end()) {
            at.emplace(key, alive.size());
            alive.push_back(nr);
        } else {
            alive[it->second] = nr;
        }
    }
    std::sort(alive.begin(), alive.end(), [](const MergeRule& a, const MergeRule& b) { return a.rank < b.rank; });
    return {};
}

// The string-keyed merge of a few units on the host (core.c:66-209): only used for the prefix encoded as a
// word of its own, which stays on the string path even when a merges file is loaded (core.c:421-446).
std::vector<int32_t> host_string_bpe(std::vector<std::string> u, const std::unordered_map<std::string, int32_t>& vocab) {
    for (;;) {
        int64_t best = INT64_MAX;
        size_t at = 0;
        for (size_t i = 0; i + 1 < u.size(); i++) {
            auto it = vocab.find(u[i] + u[i + 1]);
            if (it == vocab.end() || it->second == -1) continue;
            if ((int64_t)it->second < best) {
                best = it->second;
                at = i;
            }
        }
        if (best == INT64_MAX) break;
        u[at] += u[at + 1];
        u.erase(u.begin() + (long)at + 1);
    }
    std::vector<int32_t> ids;
    for (auto& x : u) {
        auto it = vocab.find(x);
        ids.push_back(it == vocab.end() ? -1 : it->second);
    }
    return ids;
}

// Device tables for the id-keyed path.  Tokens are ids here, so a symbol stands for an id: one symbol per
// (id, rule that produces it), numbered in rule-rank order so that "smaller symbol" is "smaller rank" among
// merge results exactly as on the string path, plus one "base" symbol for an id that can be an initial unit
// or that no rule produces.  The merge kernels are the same; only the tables differ.
LoadError build_id_tables(const std::unordered_map<std::string, int32_t>& vocab, const std::string special[256],
                          const bool has_special[256], const char* prefix, bool is_byte_encoder,
                          const std::vector<MergeRule>& alive, Tables& T) {
    T.id_path = true;
    auto single_char = [](const std::string& k) {
        return !k.empty() && (size_t)lead_len((unsigned char)k[0]) == k.size();
    };
    // ---- symbols ----
    std::unordered_map<int32_t, std::vector<uint32_t>> produced;  // id -> indices into alive
    for (size_t k = 0; k < alive.size(); k++) produced[alive[k].m].push_back((uint32_t)k);
    std::unordered_map<int32_t, bool> unit_id;  // ids a single character can start as (core.c:460-474)
    std::vector<int32_t> ids;
    {
        std::unordered_map<int32_t, bool> seen;
        for (auto& kv : vocab) {
            if (kv.second == -1) continue;
            if (single_char(kv.first)) unit_id[kv.second] = true;
            if (!seen[kv.second]) {
                seen[kv.second] = true;
                ids.push_back(kv.second);
            }
        }
        std::sort(ids.begin(), ids.end());
    }
    std::unordered_map<int32_t, uint32_t> base_sym;  // id -> its base symbol
    std::vector<uint32_t> rule_sym(alive.size());    // rule (in rank order) -> symbol of its result
    // The usual case -- ids 0..N-1, every id either a unit or the result of exactly one rule, results
    // numbered in rule order (GPT-2 style files) -- lets the symbol BE the id, which saves the symbol -> id
    // lookup when the ids are written.  Otherwise: base symbols first, then the rule results in rank order.
    bool identity = !ids.empty() && ids.front() == 0 && ids.back() == (int32_t)ids.size() - 1;
    for (size_t k = 0; k < alive.size() && identity; k++) {
        if (k && alive[k].m <= alive[k - 1].m) identity = false;
        if (unit_id.count(alive[k].m) || produced[alive[k].m].size() != 1) identity = false;
    }
    if (identity) {
        T.sym_id.resize(ids.size());
        for (int32_t id : ids) {
            T.sym_id[(size_t)id] = id;
            if (!produced.count(id)) base_sym[id] = (uint32_t)id;
        }
        for (size_t k = 0; k < alive.size(); k++) rule_sym[k] = (uint32_t)alive[k].m;
    } else {
        for (int32_t id : ids)
            if (unit_id.count(id) || !produced.count(id)) {
                base_sym[id] = (uint32_t)T.sym_id.size();
                T.sym_id.push_back(id);
            }
        for (size_t k = 0; k < alive.size(); k++) {
            rule_sym[k] = (uint32_t)T.sym_id.size();
            T.sym_id.push_back(alive[k].m);
        }
    }
    T.n_vocab_sym = (uint32_t)T.sym_id.size();
    T.rank_is_sym = true;  // by construction: rule results are numbered in rank order
    T.ident_ids = identity;
    // symbols an id can be LIVE as inside a word: produced by a rule, or an initial unit
    auto live_variants = [&](int32_t id, std::vector<uint32_t>& out) {
        out.clear();
        if (unit_id.count(id)) out.push_back(base_sym[id]);
        auto it = produced.find(id);
        if (it != produced.end())
            for (uint32_t k : it->second) out.push_back(rule_sym[k]);
    };
    auto any_symbol = [&](int32_t id) -> uint32_t {  // for output only
        if (id == -1) return SYM_UNK;
        auto b = base_sym.find(id);
        if (b != base_sym.end()) return b->second;
        auto it = produced.find(id);
        if (it != produced.end()) return rule_sym[it->second[0]];
        return SYM_UNK;
    };
    auto unit_symbol = [&](const std::string& u) -> uint32_t {  // initial unit: vocabulary lookup of one character
        auto it = vocab.find(u);
        if (it == vocab.end() || it->second == -1) return SYM_UNK;
        return base_sym[it->second];
    };

    // ---- items -> initial symbols: a replacement must be ONE character (the unit rule here is the UTF-8
    // length alone, core.c:460-474, so a longer replacement is several units per input item) ----
    std::string item_str[256];
    std::vector<uint32_t> multi[256];
    for (int b = 1; b < 256; b++) {
        T.item_sym[b] = SYM_UNK;
        T.item_direct[b] = 0;
        const bool never_leads = !is_byte_encoder && ((b >= 0x80 && b < 0xC0) || b >= 0xF8);
        if (never_leads) continue;
        std::string s;
        if (has_special[b]) {
            s = special[b];
            if (!single_char(s)) {
                // several characters (Llama-style "<0x0A>"): on this path the units are the characters, each looked up in
                // the vocabulary (core.c:460-474 splits by UTF-8 length only)
                for (size_t i = 0; i < s.size();) {
                    const size_t cl = (size_t)lead_len((unsigned char)s[i]);
                    if (i + cl > s.size())
                        return fail(HUTK_E_UNSUPPORTED, "special-character replacement is not a whole number of units");
                    multi[b].push_back(unit_symbol(s.substr(i, cl)));
                    i += cl;
                }
                item_str[b] = s;
                T.item_sym[b] = multi[b].empty() ? SYM_UNK : multi[b][0];
                T.item_direct[b] = 1;
                T.multi_bits[b >> 5] |= 1u << (b & 31);
                continue;
            }
        } else if (is_byte_encoder && b >= 0x80) {
            s.push_back((char)(0xC0 | (b >> 6)));
            s.push_back((char)(0x80 | (b & 0x3F)));
        } else if (b < 0x80) {
            s.push_back((char)b);
        } else {
            continue;  // multi-byte character without replacement: char table
        }
        item_str[b] = s;
        T.item_sym[b] = unit_symbol(s);
        T.item_direct[b] = 1;
    }
    T.item_sym[0] = SYM_UNK;
    T.item_direct[0] = 1;
    finish_item_units(T, multi);

    // ---- prefix ----
    if (prefix && prefix[0]) {
        T.has_prefix = true;
        const std::string p(prefix);
        for (size_t i = 0; i < p.size();) {  // prepended raw to the first word, then split by UTF-8 length
            const size_t cl = (size_t)lead_len((unsigned char)p[i]);
            if (i + cl > p.size()) return fail(HUTK_E_UNSUPPORTED, "prefix is not valid UTF-8");
            T.prefix_syms.push_back(unit_symbol(p.substr(i, cl)));
            i += cl;
        }
        // encoded as a word of its own: through the pretokenizer, UTF-8 units, STRING-keyed merges (core.c:421-446)
        std::string enc;
        for (size_t i = 0; i < p.size();) {
            const unsigned char b = (unsigned char)p[i];
            const size_t cl = is_byte_encoder ? 1 : (size_t)lead_len(b);
            if (i + cl > p.size()) return fail(HUTK_E_UNSUPPORTED, "prefix is not valid UTF-8");
            if (has_special[b]) enc += special[b];
            else if (is_byte_encoder && b >= 0x80) {
                enc.push_back((char)(0xC0 | (b >> 6)));
                enc.push_back((char)(0x80 | (b & 0x3F)));
            } else enc.append(p, i, cl);
            i += cl;
        }
        std::vector<std::string> units;
        if (!split_units(enc, false, units))
            return fail(HUTK_E_UNSUPPORTED, "encoded prefix is not a whole number of units");
        T.prefix_alone_final = true;
        T.prefix_alone_ids = host_string_bpe(units, vocab);
        for (int32_t id : T.prefix_alone_ids) T.prefix_alone_syms.push_back(any_symbol(id));
    }

    // ---- pairs: one entry per rule and per pair of live variants of its two ids ----
    std::vector<uint64_t> entries;
    entries.reserve(alive.size() + 16);
    std::vector<uint32_t> lv, rv;
    for (size_t k = 0; k < alive.size(); k++) {
        live_variants(alive[k].l, lv);
        live_variants(alive[k].r, rv);
        for (uint32_t a : lv)
            for (uint32_t b : rv) entries.push_back(pair_slot(a, b, rule_sym[k]));
        if (entries.size() > (size_t)8 << 20)
            return fail(HUTK_E_UNSUPPORTED, "merges file with too many rules per token id");
    }
    { LoadError pe = finish_pair_table(T, entries); if (pe.code) return pe; }
    finish_bytepair(T);
    {   // seam map: a base symbol ends like its characters, a rule's result begins like its left and ends like its right
        const UnitEnds U = unit_ends_of_items(special, has_special, is_byte_encoder, false);
        std::vector<RawEnds> ends(T.sym_id.size());
        for (auto& kv : vocab)
            if (kv.second != -1 && single_char(kv.first)) {
                auto b = base_sym.find(kv.second);
                if (b != base_sym.end() && b->second < ends.size()) ends[b->second].join(U.get(kv.first), true, true);
            }
        for (bool changed = true; changed;) {
            changed = false;
            for (size_t k = 0; k < alive.size(); k++) {
                live_variants(alive[k].l, lv);
                live_variants(alive[k].r, rv);
                for (uint32_t a : lv) changed |= ends[rule_sym[k]].join(ends[a], true, false);
                for (uint32_t b : rv) changed |= ends[rule_sym[k]].join(ends[b], false, true);
            }
        }
        seam_from_pairs(T, entries, ends);
    }

    // ---- whole-word table candidates: keys as raw input bytes ----
    {
        std::unordered_map<std::string, int> byte_of_unit;
        bool ambiguous = false;
        for (int b = 1; b < 256; b++) {
            if (!T.item_direct[b] || ((T.multi_bits[b >> 5] >> (b & 31)) & 1u)) continue;
            auto ins = byte_of_unit.emplace(item_str[b], b);
            if (!ins.second) ambiguous = true;
        }
        T.cand_off.push_back(0);
        if (!ambiguous)
            for (auto& kv : vocab) {
                if (kv.second == -1) continue;
                const std::string& k = kv.first;
                std::string raw;
                bool ok = true;
                size_t n_units = 0;
                for (size_t i = 0; i < k.size() && ok;) {
                    const size_t cl = (size_t)lead_len((unsigned char)k[i]);
                    if (i + cl > k.size()) { ok = false; break; }
                    const std::string u = k.substr(i, cl);
                    auto bt = byte_of_unit.find(u);
                    if (bt != byte_of_unit.end()) raw.push_back((char)bt->second);
                    else if (!is_byte_encoder && cl >= 2 && !has_special[(unsigned char)u[0]]) raw += u;
                    else ok = false;
                    i += cl;
                    n_units++;
                }
                if (!ok || n_units > 16 || raw.size() < 2 || raw.size() > 16) continue;
                const uint32_t sym = any_symbol(kv.second);
                if (sym == SYM_UNK) continue;
                T.cand_bytes.insert(T.cand_bytes.end(), raw.begin(), raw.end());
                T.cand_off.push_back((uint32_t)T.cand_bytes.size());
                T.cand_sym.push_back(sym);
            }
    }

    // ---- non-byte mode: multi-byte character -> symbol ----
    {
        std::vector<std::pair<uint32_t, uint32_t>> chars;
        if (!is_byte_encoder)
            for (auto& kv : vocab) {
                const std::string& c = kv.first;
                if (kv.second != -1 && c.size() >= 2 && c.size() <= 4 && single_char(c))
                    chars.push_back({pack_char(c), base_sym[kv.second]});
            }
        finish_char_table(T, chars);
    }
    return {};
}


// ------------------------------------------------------------------------
// decode direction: per-token output bytes (see Tables::dec_*)
// ------------------------------------------------------------------------
void build_decode_tables(const std::unordered_map<std::string, int32_t>& vocab, int64_t n_lines,
                         const std::string special[256], const bool has_special[256], const int seq[256],
                         const char* prefix, bool is_byte_encoder, Tables& T) {
    T.dec_n = n_lines;
    const size_t N = (size_t)n_lines;
    std::vector<const std::string*> key(N, nullptr);
    T.dec_flag.assign(N, DEC_F_HOLE);
    for (auto& kv : vocab) {
        if (kv.second < 0 || (int64_t)kv.second >= n_lines) continue;
        uint8_t& f = T.dec_flag[(size_t)kv.second];
        if (key[(size_t)kv.second]) f |= DEC_F_AMBIGUOUS;
        key[(size_t)kv.second] = &kv.first;
        f &= (uint8_t)~DEC_F_HOLE;
    }
    // one scan step of pretokenizer_decode at s[p]: *adv bytes consumed, output appended to out.  *context:
    // the step would depend on bytes after the end of s (a longer special value could still match, or the
    // character is cut off).
    auto step = [&](const std::string& s, size_t p, std::string& out, size_t* adv, bool* context) {
        size_t best_len = 0;
        int best_idx = -1, best_seq = -1;
        const size_t rest = s.size() - p;
        for (int i = 0; i < 256; i++) {
            if (!has_special[i] || special[i].empty()) continue;
            const std::string& v = special[i];
            if (v.size() <= rest) {
                if (s.compare(p, v.size(), v) != 0) continue;
                if (v.size() > best_len || (v.size() == best_len && seq[i] > best_seq)) {
                    best_len = v.size();
                    best_idx = i;
                    best_seq = seq[i];
                }
            } else if (v.compare(0, rest, s, p, rest) == 0) {
                *context = true;  // the rest of the token is a proper prefix of this value
            }
        }
        if (best_idx >= 0) {
            out.push_back((char)best_idx);
            *adv = best_len;
            return;
        }
        const unsigned char b = (unsigned char)s[p];
        size_t l = (size_t)lead_len(b);
        const bool bad_lead = b >= 0x80 && l == 1;
        if (!bad_lead && p + l > s.size()) {
            *context = true;
            l = s.size() - p;
        }
        if (is_byte_encoder) {  // pretokenizer.c:236-246
            // utf8_to_codepoint (pretokenizer.c:175-195): plain bit arithmetic, no validation
            auto at = [&](size_t q) -> uint32_t { return q < s.size() ? (unsigned char)s[q] : 0u; };
            uint32_t cp = 0xFFFD;
            if (b < 0x80) cp = b;
            else if ((b & 0xE0) == 0xC0) cp = ((b & 0x1Fu) << 6) | (at(p + 1) & 0x3Fu);
            else if ((b & 0xF0) == 0xE0) cp = ((b & 0x0Fu) << 12) | ((at(p + 1) & 0x3Fu) << 6) | (at(p + 2) & 0x3Fu);
            else if ((b & 0xF8) == 0xF0)
                cp = ((b & 0x07u) << 18) | ((at(p + 1) & 0x3Fu) << 12) | ((at(p + 2) & 0x3Fu) << 6) | (at(p + 3) & 0x3Fu);
            out.push_back(cp < 256 ? (char)cp : '?');
        } else {
            out.append(s, p, l);
        }
        *adv = l;
    };
    auto decode_alone = [&](const std::string& s, size_t from, std::string& out) -> bool {
        bool context = false;
        out.clear();
        for (size_t p = from; p < s.size();) {
            size_t adv = 1;
            step(s, p, out, &adv, &context);
            p += adv ? adv : 1;
        }
        return context;
    };
    const std::string pfx = (prefix && prefix[0]) ? std::string(prefix) : std::string();
    T.dec_off.assign(N, 0);
    T.dec_len.assign(N, DEC_BAD);
    if (!pfx.empty()) {
        T.dec_soff.assign(N, 0);
        T.dec_slen.assign(N, DEC_NOSTRIP);
    }
    std::string out;
    for (size_t id = 0; id < N; id++) {
        if (!key[id] || (T.dec_flag[id] & DEC_F_AMBIGUOUS)) continue;
        const std::string& k = *key[id];
        if (decode_alone(k, 0, out)) T.dec_flag[id] |= DEC_F_CONTEXT;
        if (out.size() >= DEC_NOSTRIP) continue;  // cannot happen with 2047-byte keys; stays DEC_BAD
        T.dec_blob.resize((T.dec_blob.size() + 3) & ~(size_t)3, 0);  // entries start on 4-byte boundaries
        T.dec_off[id] = (uint32_t)T.dec_blob.size();
        T.dec_len[id] = (uint16_t)out.size();
        T.dec_blob.insert(T.dec_blob.end(), out.begin(), out.end());
        if (!pfx.empty()) {
            if (k.size() >= pfx.size() && k.compare(0, pfx.size(), pfx) == 0) {
                if (decode_alone(k, pfx.size(), out)) T.dec_flag[id] |= DEC_F_CONTEXT;
                T.dec_blob.resize((T.dec_blob.size() + 3) & ~(size_t)3, 0);
                T.dec_soff[id] = (uint32_t)T.dec_blob.size();
                T.dec_slen[id] = (uint16_t)out.size();
                T.dec_blob.insert(T.dec_blob.end(), out.begin(), out.end());
            } else if (k.size() < pfx.size() && pfx.compare(0, k.size(), k) == 0) {
                T.dec_flag[id] |= DEC_F_PFX_PARTIAL;
            }
        }
    }
    T.dec_blob.resize(((T.dec_blob.size() + 3) & ~(size_t)3) + 16, 0);  // 16-byte reads at a token's offset stay inside
}

}  // namespace

LoadError load_tables(const char* vocab_path, const char* special_path, const char* prefix,
                      bool is_byte_encoder, const char* merges_path, Tables& T) {
    std::unordered_map<std::string, int32_t> vocab;
    int64_t n_lines = 0;
    LoadError e = read_vocab(vocab_path, vocab, &n_lines);
    if (e.code) return e;
    std::string special[256];
    bool has_special[256] = {false};
    int special_seq[256] = {0};
    e = read_special(special_path, special, has_special, special_seq);
    if (e.code) return e;

    T = Tables();
    T.is_byte_encoder = is_byte_encoder;
    T.n_keys = (int64_t)vocab.size();
    build_decode_tables(vocab, n_lines, special, has_special, special_seq, prefix, is_byte_encoder, T);
    if (merges_path) {
        // a merges file switches the reference to its id-keyed merge loop -- unless the file has no countable
        // line, in which case no merges map exists and the string path below stays in force (lib.c:592, core.c:457)
        bool has_merges = false;
        std::vector<MergeRule> alive;
        e = read_merges(merges_path, vocab, &has_merges, alive);
        if (e.code) return e;
        if (has_merges) return build_id_tables(vocab, special, has_special, prefix, is_byte_encoder, alive, T);
    }

    // ---- symbols: keys in ascending id order (an id of -1 is "absent",
    // core.c:100,155,168,205-207) ----
    std::vector<std::pair<int32_t, const std::string*>> order;
    order.reserve(vocab.size());
    for (auto& kv : vocab)
        if (kv.second != -1) order.push_back({kv.second, &kv.first});
    std::sort(order.begin(), order.end(), [](auto& a, auto& b) {
        return a.first != b.first ? a.first < b.first : *a.second < *b.second;
    });
    std::unordered_map<std::string, uint32_t> sym_of;
    sym_of.reserve(order.size() * 2);
    T.sym_id.reserve(order.size() + 512);
    T.rank_is_sym = true;
    T.ident_ids = true;
    for (size_t i = 0; i < order.size(); i++) {
        sym_of.emplace(*order[i].second, (uint32_t)i);
        T.sym_id.push_back(order[i].first);
        if (i && order[i].first == order[i - 1].first) T.rank_is_sym = false;
        if (order[i].first != (int32_t)i) T.ident_ids = false;
    }
    T.n_vocab_sym = (uint32_t)order.size();
    auto pseudo = [&](const std::string& s) -> uint32_t {
        auto it = sym_of.find(s);
        if (it != sym_of.end()) return it->second;
        uint32_t id = (uint32_t)T.sym_id.size();
        sym_of.emplace(s, id);
        T.sym_id.push_back(-1);
        return id;
    };

    // ---- items -> initial symbols ----
    std::vector<std::string> units;
    std::vector<uint32_t> multi[256];
    const bool raw_lt_possible = !has_special[(unsigned char)'<'];
    for (int b = 1; b < 256; b++) {
        T.item_sym[b] = SYM_UNK;
        T.item_direct[b] = 0;
        // outside byte-encoder mode an item is a whole character indexed by its lead
        // byte; 0x80-0xBF and 0xF8-0xFF never lead a valid character, and text that
        // is not valid UTF-8 is rejected there, so their entries are inert
        const bool never_leads = !is_byte_encoder && ((b >= 0x80 && b < 0xC0) || b >= 0xF8);
        if (never_leads) continue;
        std::string s;
        if (has_special[b]) {
            s = special[b];
            if (!split_units(s, true, units))
                return fail(HUTK_E_UNSUPPORTED,
                            "special-character replacement is not a whole number of units");
            if (raw_lt_possible && s[0] == '0')
                return fail(HUTK_E_UNSUPPORTED,
                            "special-character replacement starting with '0' could complete a "
                            "\"<0x..>\" literal begun by a raw '<'");
            if (units.size() != 1) {
                // a replacement of several units (pretokenizer.c:102-168 emits any string; tests/test_pretokenizer.c:38-41
                // 'a' -> "Alpha"), or of none: the item's units are listed apart, and a word that holds such an item is
                // encoded by the exception kernels, which expand it (k_tiles, d_exc)
                multi[b].reserve(units.size());
                for (auto& u : units) multi[b].push_back(pseudo(u));
                T.item_sym[b] = units.empty() ? SYM_UNK : multi[b][0];
                T.item_direct[b] = 1;
                T.multi_bits[b >> 5] |= 1u << (b & 31);
                continue;
            }
        } else if (is_byte_encoder && b >= 0x80) {  // pretokenizer.c:138-141
            s.push_back((char)(0xC0 | (b >> 6)));
            s.push_back((char)(0x80 | (b & 0x3F)));
        } else if (b < 0x80) {
            s.push_back((char)b);
        } else {
            continue;  // multi-byte character without replacement: char table
        }
        // a unit that is no key still gets a symbol of its own (id -1): it can be
        // one half of a longer key (core.c:700-722 looks the concatenation up)
        T.item_sym[b] = pseudo(s);
        T.item_direct[b] = 1;
    }
    finish_item_units(T, multi);
    T.item_sym[0] = SYM_UNK;
    T.item_direct[0] = 1;

    // ---- prefix (core.c:421-451) ----
    if (prefix && prefix[0]) {
        T.has_prefix = true;
        std::string p(prefix);
        if (!split_units(p, true, units))
            return fail(HUTK_E_UNSUPPORTED, "prefix is not a whole number of units");
        for (auto& u : units) T.prefix_syms.push_back(pseudo(u));
        // the prefix encoded as a word of its own goes through the pretokenizer
        // (special characters apply) and is split by UTF-8 length only
        std::string enc;
        for (size_t i = 0; i < p.size();) {
            unsigned char b = (unsigned char)p[i];
            size_t cl = is_byte_encoder ? 1 : (size_t)lead_len(b);
            if (i + cl > p.size()) return fail(HUTK_E_UNSUPPORTED, "prefix is not valid UTF-8");
            if (has_special[b]) enc += special[b];
            else if (is_byte_encoder && b >= 0x80) {
                enc.push_back((char)(0xC0 | (b >> 6)));
                enc.push_back((char)(0x80 | (b & 0x3F)));
            } else enc.append(p, i, cl);
            i += cl;
        }
        if (!split_units(enc, false, units))
            return fail(HUTK_E_UNSUPPORTED, "encoded prefix is not a whole number of units");
        for (auto& u : units) T.prefix_alone_syms.push_back(pseudo(u));
    }

    // ---- pairs: every split of every key whose halves are symbols ----
    std::vector<uint64_t> entries;
    entries.reserve(order.size() * 3);
    for (size_t i = 0; i < order.size(); i++) {
        const std::string& k = *order[i].second;
        for (size_t s = 1; s < k.size(); s++) {
            std::string l = k.substr(0, s), r = k.substr(s);
            uint32_t ls, rs;
            // outside byte-encoder mode a unit that is no key (an unknown character)
            // can still be one half of a key: give it a symbol
            auto li = sym_of.find(l);
            if (li != sym_of.end()) ls = li->second;
            else if (!is_byte_encoder && unit_shaped(l)) ls = pseudo(l);
            else continue;
            auto ri = sym_of.find(r);
            if (ri != sym_of.end()) rs = ri->second;
            else if (!is_byte_encoder && unit_shaped(r)) rs = pseudo(r);
            else continue;
            entries.push_back(pair_slot(ls, rs, (uint32_t)i));
        }
    }
    { LoadError pe = finish_pair_table(T, entries); if (pe.code) return pe; }
    finish_bytepair(T);
    {   // seam map: a symbol begins like its first unit and ends like its last one
        const UnitEnds U = unit_ends_of_items(special, has_special, is_byte_encoder, true);
        std::vector<RawEnds> ends(T.sym_id.size());
        std::vector<std::string> us;
        for (auto& kv : sym_of) {
            if (kv.second >= ends.size()) continue;
            RawEnds& e = ends[kv.second];
            if (!split_units(kv.first, true, us) || us.empty()) { e.all(); continue; }  // (no run of whole units: cannot be a token; stay on the safe side)
            e.join(U.get(us.front()), true, false);
            e.join(U.get(us.back()), false, true);
        }
        seam_from_pairs(T, entries, ends);
        // the symbols' input bytes, where every unit of the key has exactly ONE input byte that makes it
        std::vector<std::string> raw(order.size());
        if (is_byte_encoder && !T.has_multi) {
            std::unordered_map<uint32_t, std::pair<int, int>> makers;  // unit symbol -> (input bytes that make it, one of them)
            for (int b = 1; b < 256; b++) {
                auto& m = makers[T.item_sym[b]];
                m.first++;
                m.second = b;
            }
            for (size_t i = 0; i < order.size(); i++) {
                if (!split_units(*order[i].second, true, us) || us.empty()) continue;
                std::string r;
                for (auto& u : us) {
                    auto it = sym_of.find(u);
                    auto mk = it == sym_of.end() ? makers.end() : makers.find(it->second);
                    if (mk == makers.end() || mk->second.first != 1) { r.clear(); break; }
                    r.push_back((char)mk->second.second);
                }
                raw[i] = r;
            }
        }
        seam2_build(T, entries, ends, raw);
    }

    // ---- keys as raw input bytes (whole-word table candidates) ----
    // A key is a candidate when every unit of it is produced by exactly one input: the single byte whose
    // pretokenizer output is that unit, or (non-byte mode) the multi-byte character itself.  Whether the
    // raw bytes really are one word that encodes to the key's id is decided on the device when the context
    // is created (hutk_api.cpp build_word_table); a wrong guess here only costs a dropped candidate.
    {
        std::unordered_map<uint32_t, int> byte_of_sym;
        bool ambiguous = false;
        for (int b = 1; b < 256; b++) {
            if (!T.item_direct[b]) continue;  // lead byte of a character looked up in the char table
            if ((T.multi_bits[b >> 5] >> (b & 31)) & 1u) continue;  // (its words never reach the table)
            auto ins = byte_of_sym.emplace(T.item_sym[b], b);
            if (!ins.second) ambiguous = true;
        }
        T.cand_off.push_back(0);
        if (!ambiguous) {
            std::vector<std::string> us;
            for (size_t i = 0; i < order.size(); i++) {
                const std::string& k = *order[i].second;
                if (!split_units(k, true, us) || us.empty() || us.size() > (size_t)WORDL_KEY_BYTES) continue;
                std::string raw;
                bool ok = true;
                for (auto& u : us) {
                    auto it = sym_of.find(u);
                    if (it == sym_of.end()) { ok = false; break; }
                    auto bt = byte_of_sym.find(it->second);
                    if (bt != byte_of_sym.end()) raw.push_back((char)bt->second);
                    else if (!is_byte_encoder && u.size() >= 2 && u.size() <= 4 &&
                             (size_t)lead_len((unsigned char)u[0]) == u.size()) raw += u;
                    else { ok = false; break; }
                }
                if (!ok || raw.size() < 2 || raw.size() > (size_t)WORDL_KEY_BYTES) continue;
                T.cand_bytes.insert(T.cand_bytes.end(), raw.begin(), raw.end());
                T.cand_off.push_back((uint32_t)T.cand_bytes.size());
                T.cand_sym.push_back((uint32_t)i);
            }
        }
    }

    // ---- non-byte mode: multi-byte character -> symbol ----
    {
        std::vector<std::pair<uint32_t, uint32_t>> chars;
        if (!is_byte_encoder)
            for (auto& kv : sym_of) {
                const std::string& s = kv.first;
                if (s.size() >= 2 && s.size() <= 4 && (size_t)lead_len((unsigned char)s[0]) == s.size())
                    chars.push_back({pack_char(s), kv.second});
            }
        finish_char_table(T, chars);
    }
    return {};
}

}  // namespace hutk
