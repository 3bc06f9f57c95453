// Seam map, second level (Tables::seam2_*, DevTables::seam2_*): what the loader and the kernels must agree on.
#pragma once
#include <cstdint>
#if defined(__HIPCC__)
#define HUTK_HD __host__ __device__
#else
#define HUTK_HD
#endif
namespace hutk {
// The set's keys: what stands in front of the boundary -- a whole three-byte character A (kind 3) or only its last byte x
// (kind 1) -- and what stands behind it -- a whole character B (3), its first two bytes (2: a token that IS such a prefix)
// or only its lead byte y (1) --, little-endian values, and the two kinds as a tag.  (1, 1) is not in the set: seam2_part.
HUTK_HD inline uint32_t seam2_hash(uint32_t a, uint32_t b, uint32_t kind_a, uint32_t kind_b) {
    const uint64_t k = ((uint64_t)a | ((uint64_t)b << 24) | ((uint64_t)(kind_a * 4u + kind_b) << 48)) * 0x9E3779B97F4A7C15ull;
    return (uint32_t)(k >> 32);
}
HUTK_HD inline uint32_t seam2_cat_bit(uint32_t kind_a, uint32_t kind_b) { return 1u << (kind_a * 4u + kind_b); }
// a well-formed three-byte character's bytes: E0..EF, 80..BF, 80..BF
HUTK_HD inline bool seam2_char3(uint32_t c3) {
    return (c3 & 0xF0u) == 0xE0u && (c3 & 0xC000u) == 0x8000u && (c3 & 0xC00000u) == 0x800000u;
}
}  // namespace hutk
