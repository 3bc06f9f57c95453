// hutk_decode.hip -- decode direction on the device: ids -> text.
//
// The reference concatenates the tokens' vocabulary strings, strips the prefix from the front of the text and
// maps the result back to input bytes with a left-to-right scan (src/core.c:513-581,
// src/pretokenizer.c:197-296).  hutk_loader.cpp proves per token that this scan, started at the token's first
// byte, ends at its last one whatever follows, and stores the token's output bytes; tokens for which it
// cannot (and ids without a unique key) make their document fail loudly.  What is left for the device is a
// segmented gather: lengths -> exclusive scan -> copy.  HBM-bound: 4 bytes read per id, the text written once.
//
//   k_dec_mark    first-token bitmap of the documents (prefix stripping, document offsets)
//   k_dec_pre     first document per tile of DEC_TILE ids
//   k_dec_tiles   ONE pass: lengths of the tile's tokens, block scan, the tile's offset in the text by
//                 decoupled look-back over the earlier tiles' totals, the text staged in LDS and stored in
//                 16-byte chunks, out_offsets of the documents that start in the tile
//   k_dec_tail    out_offsets of the (empty) documents at the very end
#include <hip/hip_runtime.h>

#include "hutk_device.h"

namespace hutk {

__device__ __forceinline__ void dec_raise(int32_t* err, int32_t code) { atomicCAS(err, 0, code); }

__global__ void k_dec_mark(DecArgs D) {
    const int64_t d = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= D.n_docs) return;
    const int64_t i = D.id_offsets[d];
    if (i < D.n_ids && i < D.id_offsets[d + 1]) atomicOr(&D.first_bits[i >> 5], 1u << (i & 31));
}

// document holding token index i: last d with id_offsets[d] <= i and id_offsets[d + 1] > i
__device__ __forceinline__ int64_t dec_doc_of(const DecArgs& D, int64_t i) {
    int64_t lo = 0, hi = D.n_docs;  // offsets[lo] <= i < offsets[hi]
    while (hi - lo > 1) {
        const int64_t mid = (lo + hi) >> 1;
        if (D.id_offsets[mid] <= i) lo = mid; else hi = mid;
    }
    return lo;
}

#ifndef HUTK_DEC_PER_THREAD
#define HUTK_DEC_PER_THREAD 8
#endif
constexpr int DEC_THREADS = 256, DEC_PER_THREAD = HUTK_DEC_PER_THREAD, DEC_TILE = DEC_THREADS * DEC_PER_THREAD;
static_assert(DEC_PER_THREAD % 4 == 0 && 32 % DEC_PER_THREAD == 0, "16-byte id loads; a thread's first-token bits sit in one word");

// first document whose first token is at or after the tile's first token (binary search, once per tile)
__global__ void k_dec_pre(DecArgs D) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= D.n_tiles) return;
    const int64_t t0 = t * DEC_TILE;
    int64_t lo = 0, hi = D.n_docs + 1;  // first d in [0, n_docs] with id_offsets[d] >= t0
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (D.id_offsets[mid] < t0) lo = mid + 1; else hi = mid;
    }
    D.tile_first_doc[t] = lo;
}

// inclusive prefix sum over the 64 lanes of a wavefront (DPP row shifts and broadcasts)
__device__ __forceinline__ uint32_t dec_wave_incl(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);
    return v;
}
constexpr unsigned long long DEC_ST_MASK = 3ull << 62, DEC_ST_TOTAL = 1ull << 62, DEC_ST_PREFIX = 2ull << 62;
constexpr int DEC_LOOK = 1;  // look-back window = 64 * DEC_LOOK tiles per round trip (4: measured 9 % slower)
constexpr int DEC_LDS_BYTES = 3 * 1024 * DEC_PER_THREAD;  // text of one tile staged for coalesced stores (mean a third of it)

// token i of the batch: its table entry (see DecTables); errors are reported here and leave an empty token
__device__ __forceinline__ uint2 dec_entry(const DecTables& T, const DecArgs& D, int64_t i, int32_t id, bool first) {
    if (id < 0 || (int64_t)id >= T.n) {
        dec_raise(D.err, HUTK_E_VALUE);  // "Element must be non-negative and less than vocab size."
        if (D.status) D.status[dec_doc_of(D, i)] = HUTK_DOC_ID_OUT_OF_RANGE;
        return make_uint2(0, 0);
    }
    const uint2 e = (first && T.sent) ? T.sent[id] : T.ent[id];
    if ((e.x & 0xFFu) == DEC_TAG_BAD) {
        dec_raise(D.err, HUTK_E_UNSUPPORTED);
        if (D.status) D.status[dec_doc_of(D, i)] = HUTK_DOC_ID_UNDECODABLE;
        return make_uint2(0, 0);
    }
    return e;
}
__device__ __forceinline__ uint32_t dec_len_of(uint2 e) { return (e.x & DEC_TAG_LONG) ? e.x >> 8 : (e.x & 0xFFu); }

// Bytes of tile p's text, computed by ONE lane: what a workgroup falls back on when a tile in front of its own has not
// published its total for a long time (see the look-back in k_dec_tiles).  Errors are reported by the tile's own
// workgroup as well; reporting them twice is harmless.
__device__ __forceinline__ uint64_t dec_tile_total(const DecTables& T, const DecArgs& D, int64_t p) {
    const int64_t a = p * DEC_TILE, b = (a + DEC_TILE < D.n_ids) ? a + DEC_TILE : D.n_ids;
    uint64_t sum = 0;
    for (int64_t i = a; i < b; i++) {
        const bool first = D.first_bits && ((D.first_bits[i >> 5] >> (i & 31)) & 1u);
        sum += dec_len_of(dec_entry(T, D, i, D.ids[i], first));
    }
    return sum;
}

template <bool WRITE>
__global__ __launch_bounds__(DEC_THREADS) __attribute__((amdgpu_waves_per_eu(5))) void k_dec_tiles(DecTables T, DecArgs D) {  // (96 VGPRs: with the look-back's fallback inlined the compiler takes 98 and loses a wavefront per SIMD)
    __shared__ uint32_t s_part[DEC_THREADS / 64];
    __shared__ uint16_t s_pref[DEC_TILE];  // bytes of the tile before each of its tokens
    __shared__ __attribute__((aligned(16))) uint8_t s_text[WRITE ? DEC_LDS_BYTES + 16 : 16];
    const int tid = threadIdx.x;
    const int64_t tile = blockIdx.x;
    const int64_t i0 = tile * DEC_TILE + (int64_t)tid * DEC_PER_THREAD;
    uint2 ent[DEC_PER_THREAD];
    uint32_t len[DEC_PER_THREAD];
    uint32_t firsts = 0;  // bit k: token i0 + k starts a document
    {
        const uint32_t w = (D.first_bits && i0 < D.n_ids) ? D.first_bits[i0 >> 5] : 0u;  // DEC_PER_THREAD = 8 divides 32
        firsts = (w >> (i0 & 31)) & ((1u << DEC_PER_THREAD) - 1u);
    }
    int32_t id[DEC_PER_THREAD];
    if (i0 + DEC_PER_THREAD <= D.n_ids && (reinterpret_cast<uintptr_t>(D.ids) & 15) == 0) {  // two 16-byte loads
#pragma unroll
        for (int g = 0; g < DEC_PER_THREAD; g += 4) {
            const int4 a = *reinterpret_cast<const int4*>(D.ids + i0 + g);
            id[g] = a.x; id[g + 1] = a.y; id[g + 2] = a.z; id[g + 3] = a.w;
        }
    } else {
#pragma unroll
        for (int k = 0; k < DEC_PER_THREAD; k++) id[k] = (i0 + k < D.n_ids) ? D.ids[i0 + k] : 0;
    }
    uint32_t mine = 0, any_long = 0;
#pragma unroll
    for (int k = 0; k < DEC_PER_THREAD; k++) {
        ent[k] = make_uint2(0, 0);
        if (i0 + k < D.n_ids) ent[k] = dec_entry(T, D, i0 + k, id[k], (firsts >> k) & 1u);
        len[k] = dec_len_of(ent[k]);
        any_long |= ent[k].x & DEC_TAG_LONG;
        mine += len[k];
    }
    // block exclusive scan of the per-thread byte counts: DPP scan per wavefront, then the four wave totals
    const uint32_t incl = dec_wave_incl(mine);
    if ((tid & 63) == 63) s_part[tid >> 6] = incl;
    __syncthreads();
    uint32_t wave_base = 0, total = 0;
#pragma unroll
    for (int w = 0; w < DEC_THREADS / 64; w++) {
        const uint32_t t = s_part[w];
        if (w < (tid >> 6)) wave_base += t;
        total += t;
    }
    uint32_t before = wave_base + incl - mine;
    // Byte offset of the tile's text = total of all earlier tiles, by decoupled look-back: every tile
    // publishes its own total at once and its inclusive prefix as soon as it knows it; a tile adds up the totals
    // of its predecessors back to the nearest published prefix.  One pass over the ids instead of sizes + scan +
    // write.  Workgroups start in index order on this hardware, so a predecessor is running or done and the wait
    // is short; HIP promises no such order, though, and nothing here depends on it: a workgroup that has waited
    // for a predecessor for milliseconds adds that tile's bytes up ITSELF (dec_tile_total) and goes on, so every
    // look-back ends whatever the dispatch order -- slowly in that case, but without a hang or a wrong offset.
    // (Tiles handed out by ticket instead -- a returning atomic per workgroup -- cost 12 % of the kernel.)
    // Flag and value share one 64-bit word, so relaxed agent-scope atomics suffice: nothing else is
    // communicated, and acquire/release at agent scope would write back / invalidate the XCD's L2 per tile.
    unsigned long long* st = D.tile_state;
    if (tid == 0 && tile > 0)
        __hip_atomic_store(&st[tile], DEC_ST_TOTAL | (unsigned long long)total, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    // While the predecessors get there: bytes before each token of the tile, and the tile's text staged in LDS
    // from index 0 (its alignment in the output is not known yet).
    {
        uint32_t pos = before;
#pragma unroll
        for (int k = 0; k < DEC_PER_THREAD; k++) {
            s_pref[tid * DEC_PER_THREAD + k] = (uint16_t)pos;
            pos += len[k];
        }
    }
    const bool write = WRITE && D.bytes_out != nullptr;
    const bool staged = write && total <= (uint32_t)DEC_LDS_BYTES;  // (always < 65536: s_pref holds 16-bit positions)
    if (staged) {
        // short tokens (almost all) carry their bytes in the entry; long ones are copied from the blob
        // (OR-ing shifted dwords into a zeroed area with LDS atomics instead of plain stores: measured 5 % slower)
        uint32_t pos = before;
#pragma unroll
        for (int k = 0; k < DEC_PER_THREAD; k++) {
            const uint64_t bits = (((uint64_t)ent[k].y << 32) | ent[k].x) >> 8;
            const uint32_t n_in = (ent[k].x & DEC_TAG_LONG) ? 0u : len[k];
            // exactly n_in bytes as 4 + 2 + 1 (gfx950 LDS takes unaligned stores): three predicated stores, not seven
            uint64_t v = bits;
            uint32_t p = pos;
            if (n_in & 4u) {
                const uint32_t w = (uint32_t)v;
                __builtin_memcpy(s_text + p, &w, 4);
                v >>= 32;
                p += 4;
            }
            if (n_in & 2u) {
                const uint16_t w = (uint16_t)v;
                __builtin_memcpy(s_text + p, &w, 2);
                v >>= 16;
                p += 2;
            }
            if (n_in & 1u) s_text[p] = (uint8_t)v;
            pos += len[k];
        }
        if (any_long) {
            pos = before;
#pragma unroll
            for (int k = 0; k < DEC_PER_THREAD; k++) {
                if (ent[k].x & DEC_TAG_LONG) {
                    const uint8_t* src = T.blob + ent[k].y;
                    for (uint32_t j = 0; j < len[k]; j++) s_text[pos + j] = src[j];
                }
                pos += len[k];
            }
        }
    }
    __shared__ int64_t s_g0;
    if (tid < 64) {  // wavefront 0 looks back
        const int lane = tid;
        int64_t excl = 0;
        if (tile > 0) {
            // DEC_LOOK * 64 predecessors per round trip (hi - lane - 64 j), nearest first
            bool found = false;
            for (int64_t hi = tile - 1; !found; hi -= 64 * DEC_LOOK) {
                unsigned long long v[DEC_LOOK];
                uint32_t spins = 0;
                for (;;) {
                    bool ready = true;
#pragma unroll
                    for (int j = 0; j < DEC_LOOK; j++) {
                        const int64_t p = hi - lane - 64 * j;
                        v[j] = DEC_ST_PREFIX;  // before tile 0: an empty prefix
                        if (p >= 0) v[j] = __hip_atomic_load(&st[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
#pragma unroll
                    for (int j = 0; j < DEC_LOOK; j++) ready = ready && (v[j] & DEC_ST_MASK) != 0;
                    if (__all(ready)) break;
                    if (++spins > D.help_after) {  // (tens of milliseconds by default) not dispatched yet, perhaps never before we leave
#pragma unroll
                        for (int j = 0; j < DEC_LOOK; j++)
                            if ((v[j] & DEC_ST_MASK) == 0)
                                v[j] = DEC_ST_TOTAL | (unsigned long long)dec_tile_total(T, D, hi - lane - 64 * j);
                        break;
                    }
                }
#pragma unroll
                for (int j = 0; j < DEC_LOOK; j++) {
                    if (found) continue;
                    const unsigned long long has_prefix = __ballot((v[j] & DEC_ST_MASK) == DEC_ST_PREFIX);
                    // lanes up to and including the nearest one with a prefix contribute
                    const int stop = has_prefix ? __builtin_ctzll(has_prefix) : 63;
                    int64_t part = (lane <= stop) ? (int64_t)(v[j] & ~DEC_ST_MASK) : 0;
                    for (int o = 32; o; o >>= 1) part += __shfl_xor(part, o, 64);
                    excl += part;
                    found = has_prefix != 0;
                }
            }
        }
        if (lane == 0) {
            __hip_atomic_store(&st[tile], DEC_ST_PREFIX | (unsigned long long)(excl + total), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
            s_g0 = excl;
        }
    }
    __syncthreads();
    const int64_t g0 = s_g0;  // byte offset of the tile's text in the output
    // out_offsets of the documents whose first token is in the tile (consecutive documents from tile_first_doc
    // on, empty ones included; coalesced reads of id_offsets)
    {
        const int64_t t0 = tile * DEC_TILE, t1 = t0 + DEC_TILE;
        for (int64_t d = D.tile_first_doc[tile] + tid; d < D.n_docs; d += DEC_THREADS) {
            const int64_t i = D.id_offsets[d];
            if (i >= t1 || i >= D.n_ids) break;
            D.out_offsets[d] = g0 + s_pref[i - t0];
        }
    }
    if (!write) return;
    if (g0 + (int64_t)total > D.bytes_cap) {
        if (tid == 0) dec_raise(D.err, HUTK_E_CAPACITY);
        return;
    }
    if (staged) {
        // 16-byte aligned stores: chunk c of the output holds text bytes [c - shift, c - shift + 16), read from
        // LDS as five dwords and realigned
        const uint32_t shift = (uint32_t)(g0 & 15);
        uint8_t* gbase = D.bytes_out + (g0 - shift);  // 16-byte aligned
        const uint32_t end = total + shift;
        const uint32_t* text32 = reinterpret_cast<const uint32_t*>(s_text);
        for (uint32_t c = (uint32_t)tid * 16; c < end; c += DEC_THREADS * 16) {
            if (c >= shift && c + 16 <= end) {
                const uint32_t t = c - shift, r = t & 3u;
                const uint32_t* q = text32 + (t >> 2);
                const uint32_t w0 = q[0], w1 = q[1], w2 = q[2], w3 = q[3], w4 = q[4];
                uint4 o;
                o.x = __builtin_amdgcn_alignbyte(w1, w0, r);
                o.y = __builtin_amdgcn_alignbyte(w2, w1, r);
                o.z = __builtin_amdgcn_alignbyte(w3, w2, r);
                o.w = __builtin_amdgcn_alignbyte(w4, w3, r);
                *reinterpret_cast<uint4*>(gbase + c) = o;
            } else {  // first and last chunk: only the bytes that belong to this tile
                for (uint32_t j = (c < shift ? shift : c); j < c + 16 && j < end; j++) gbase[j] = s_text[j - shift];
            }
        }
    } else {  // a tile of unusually long tokens: straight to memory
        uint8_t* dst = D.bytes_out + g0 + before;
#pragma unroll
        for (int k = 0; k < DEC_PER_THREAD; k++) {
            if (ent[k].x & DEC_TAG_LONG) {
                const uint8_t* src = T.blob + ent[k].y;
                for (uint32_t j = 0; j < len[k]; j++) dst[j] = src[j];
            } else {
                const uint64_t bits = (((uint64_t)ent[k].y << 32) | ent[k].x) >> 8;
                for (uint32_t j = 0; j < len[k]; j++) dst[j] = (uint8_t)(bits >> (8 * j));
            }
            dst += len[k];
        }
    }
}

__global__ void k_dec_tail(DecArgs D) {
    const int64_t d = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (d > D.n_docs) return;
    // documents without a first token of their own (empty ones) that are not followed by a non-empty one
    // inside the batch, and the end marker: everything decoded so far = the grand total
    if (d == D.n_docs || D.id_offsets[d] >= D.n_ids)
        D.out_offsets[d] = (int64_t)(D.tile_state[D.n_tiles - 1] & ~DEC_ST_MASK);  // inclusive prefix of the last tile
}

int64_t dec_tile_ids() { return DEC_TILE; }

void launch_dec_mark(const DecArgs& d, hipStream_t s) {
    hipLaunchKernelGGL(k_dec_mark, dim3((unsigned)((d.n_docs + 255) / 256)), dim3(256), 0, s, d);
}
void launch_dec(const DecTables& t, const DecArgs& d, hipStream_t s) {
    hipLaunchKernelGGL(k_dec_pre, dim3((unsigned)((d.n_tiles + 255) / 256)), dim3(256), 0, s, d);
    if (d.bytes_out) hipLaunchKernelGGL(k_dec_tiles<true>, dim3((unsigned)d.n_tiles), dim3(DEC_THREADS), 0, s, t, d);
    else hipLaunchKernelGGL(k_dec_tiles<false>, dim3((unsigned)d.n_tiles), dim3(DEC_THREADS), 0, s, t, d);
    hipLaunchKernelGGL(k_dec_tail, dim3((unsigned)((d.n_docs + 1 + 255) / 256)), dim3(256), 0, s, d);
}

}  // namespace hutk
