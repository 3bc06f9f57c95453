// Seam map, second level (Tables::seam2_*, DevTables::seam2_*): what the loader and the kernels must agree on.
#pragma once
#include <cstdint>
#if defined(__HIPCC__)
#define HUTK_HD __host__ __device__
#else
#define HUTK_HD
#endif
namespace hutk {
// hash of the boundary's two three-byte characters (little-endian 24-bit values); the set's bit is its top bits
HUTK_HD inline uint32_t seam2_hash(uint32_t a3, uint32_t b3) {
    const uint64_t k = ((uint64_t)a3 | ((uint64_t)b3 << 24)) * 0x9E3779B97F4A7C15ull;
    return (uint32_t)(k >> 32);
}
// a well-formed three-byte character's bytes: E0..EF, 80..BF, 80..BF
HUTK_HD inline bool seam2_char3(uint32_t c3) {
    return (c3 & 0xF0u) == 0xE0u && (c3 & 0xC000u) == 0x8000u && (c3 & 0xC00000u) == 0x800000u;
}
}  // namespace hutk
