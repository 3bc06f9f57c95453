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
//   k_dec_tiles   <false>: bytes per tile of DEC_TILE ids; <true>: after the scan of the tile totals (the
//                 k_scan_* kernels of the encode direction), stage the tile's text in LDS, store it in
//                 16-byte chunks, write out_offsets of the documents that start in the tile
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

constexpr int DEC_THREADS = 256, DEC_PER_THREAD = 8, DEC_TILE = DEC_THREADS * DEC_PER_THREAD;

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
constexpr int DEC_LDS_BYTES = 24 * 1024;  // text of one tile staged for coalesced stores (mean ~8 KB)

// token i of the batch: (offset into the blob, output length); errors are reported here
__device__ __forceinline__ uint2 dec_entry(const DecTables& T, const DecArgs& D, int64_t i, int32_t id, bool first) {
    if (id < 0 || (int64_t)id >= T.n) {
        dec_raise(D.err, HUTK_E_VALUE);  // "Element must be non-negative and less than vocab size."
        if (D.status) D.status[dec_doc_of(D, i)] = HUTK_DOC_ID_OUT_OF_RANGE;
        return make_uint2(0, 0);
    }
    uint2 e = T.ent[id];  // x: offset, y: length | flags << 16
    uint32_t flags = e.y >> 16;
    uint32_t len = e.y & 0xFFFFu;
    if (first && T.sent) {  // first token of its document and a prefix is configured
        const uint2 s = T.sent[id];
        if ((s.y & 0xFFFFu) != DEC_NOSTRIP_DEV) {
            e.x = s.x;
            len = s.y & 0xFFFFu;
        }
    } else {
        flags &= ~(uint32_t)DEC_FD_PFX_PARTIAL;  // only matters at the front of a document
    }
    if (len == DEC_BAD_DEV || flags) {
        dec_raise(D.err, HUTK_E_UNSUPPORTED);
        if (D.status) D.status[dec_doc_of(D, i)] = HUTK_DOC_ID_UNDECODABLE;
        return make_uint2(0, 0);
    }
    return make_uint2(e.x, len);
}

template <bool WRITE>
__global__ __launch_bounds__(DEC_THREADS) void k_dec_tiles(DecTables T, DecArgs D) {
    __shared__ uint32_t s_part[DEC_THREADS / 64];
    __shared__ uint16_t s_pref[WRITE ? DEC_TILE : 2];  // bytes of the tile before each of its tokens
    __shared__ __attribute__((aligned(16))) uint8_t s_text[WRITE ? DEC_LDS_BYTES : 16];
    const int tid = threadIdx.x;
    const int64_t tile = blockIdx.x;
    const int64_t i0 = tile * DEC_TILE + (int64_t)tid * DEC_PER_THREAD;
    uint32_t off[DEC_PER_THREAD], len[DEC_PER_THREAD];
    uint32_t firsts = 0;  // bit k: token i0 + k starts a document
    {
        const uint32_t w = (D.first_bits && i0 < D.n_ids) ? D.first_bits[i0 >> 5] : 0u;  // DEC_PER_THREAD = 8 divides 32
        firsts = (w >> (i0 & 31)) & 0xFFu;
    }
    uint32_t mine = 0;
#pragma unroll
    for (int k = 0; k < DEC_PER_THREAD; k++) {
        const int64_t i = i0 + k;
        off[k] = 0;
        len[k] = 0;
        if (i < D.n_ids) {
            const uint2 e = dec_entry(T, D, i, D.ids[i], (firsts >> k) & 1u);
            off[k] = e.x;
            len[k] = e.y;
        }
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
    if (!WRITE) {
        if (tid == 0) D.tile_count[tile] = total;
        return;
    }
    const int64_t g0 = D.tile_base[tile];  // byte offset of the tile's text in the output
    // bytes before each token of the tile, then out_offsets of the documents whose first token is in the tile
    // (consecutive documents from tile_first_doc on, empty ones included; coalesced reads of id_offsets)
    {
        uint32_t pos = before;
#pragma unroll
        for (int k = 0; k < DEC_PER_THREAD; k++) {
            s_pref[tid * DEC_PER_THREAD + k] = (uint16_t)pos;
            pos += len[k];
        }
    }
    __syncthreads();
    {
        const int64_t t0 = tile * DEC_TILE, t1 = t0 + DEC_TILE;
        for (int64_t d = D.tile_first_doc[tile] + tid; d < D.n_docs; d += DEC_THREADS) {
            const int64_t i = D.id_offsets[d];
            if (i >= t1 || i >= D.n_ids) break;
            D.out_offsets[d] = g0 + s_pref[i - t0];
        }
    }
    if (!D.bytes_out) return;
    if (g0 + (int64_t)total > D.bytes_cap) {
        if (tid == 0) dec_raise(D.err, HUTK_E_CAPACITY);
        return;
    }
    const uint32_t shift = (uint32_t)(g0 & 15);  // LDS index and global address agree modulo 16
    if (total + shift <= (uint32_t)DEC_LDS_BYTES) {  // (always < 65536: s_pref holds 16-bit positions)
        // the first 16 bytes of all eight tokens are loaded before any is used (entries are 4-byte aligned and
        // the blob has 16 bytes of slack): one round trip per thread, not one per byte
        uint4 v[DEC_PER_THREAD];
#pragma unroll
        for (int k = 0; k < DEC_PER_THREAD; k++)
            v[k] = len[k] ? *reinterpret_cast<const uint4*>(T.blob + off[k]) : make_uint4(0, 0, 0, 0);
        uint32_t pos = before + shift;
#pragma unroll
        for (int k = 0; k < DEC_PER_THREAD; k++) {
            const uint32_t w[4] = {v[k].x, v[k].y, v[k].z, v[k].w};
#pragma unroll
            for (int j = 0; j < 8; j++)
                if ((uint32_t)j < len[k]) s_text[pos + j] = (uint8_t)(w[j >> 2] >> (8 * (j & 3)));
            if (len[k] > 8) {
#pragma unroll
                for (int j = 8; j < 16; j++)
                    if ((uint32_t)j < len[k]) s_text[pos + j] = (uint8_t)(w[j >> 2] >> (8 * (j & 3)));
                const uint8_t* src = T.blob + off[k];
                for (uint32_t j = 16; j < len[k]; j++) s_text[pos + j] = src[j];
            }
            pos += len[k];
        }
        __syncthreads();
        uint8_t* gbase = D.bytes_out + (g0 - shift);  // 16-byte aligned
        const uint32_t end = total + shift;
        for (uint32_t c = (uint32_t)tid * 16; c < end; c += DEC_THREADS * 16) {
            if (c >= shift && c + 16 <= end) {
                *reinterpret_cast<uint4*>(gbase + c) = *reinterpret_cast<const uint4*>(s_text + c);
            } else {  // first and last chunk: only the bytes that belong to this tile
                for (uint32_t j = (c < shift ? shift : c); j < c + 16 && j < end; j++) gbase[j] = s_text[j];
            }
        }
    } else {  // a tile of unusually long tokens: straight to memory
        uint8_t* dst = D.bytes_out + g0 + before;
#pragma unroll
        for (int k = 0; k < DEC_PER_THREAD; k++) {
            const uint8_t* src = T.blob + off[k];
            for (uint32_t j = 0; j < len[k]; j++) dst[j] = src[j];
            dst += len[k];
        }
    }
}

__global__ void k_dec_tail(DecArgs D) {
    const int64_t d = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (d > D.n_docs) return;
    // documents without a first token of their own (empty ones) that are not followed by a non-empty one
    // inside the batch, and the end marker: everything decoded so far = the grand total
    if (d == D.n_docs || D.id_offsets[d] >= D.n_ids) D.out_offsets[d] = D.tile_base[D.n_tiles];
}

int64_t dec_tile_ids() { return DEC_TILE; }

void launch_dec_mark(const DecArgs& d, hipStream_t s) {
    hipLaunchKernelGGL(k_dec_mark, dim3((unsigned)((d.n_docs + 255) / 256)), dim3(256), 0, s, d);
}
void launch_dec_sizes(const DecTables& t, const DecArgs& d, hipStream_t s) {
    hipLaunchKernelGGL(k_dec_pre, dim3((unsigned)((d.n_tiles + 255) / 256)), dim3(256), 0, s, d);
    hipLaunchKernelGGL(k_dec_tiles<false>, dim3((unsigned)d.n_tiles), dim3(DEC_THREADS), 0, s, t, d);
}
void launch_dec_write(const DecTables& t, const DecArgs& d, hipStream_t s) {
    hipLaunchKernelGGL(k_dec_tiles<true>, dim3((unsigned)d.n_tiles), dim3(DEC_THREADS), 0, s, t, d);
    hipLaunchKernelGGL(k_dec_tail, dim3((unsigned)((d.n_docs + 1 + 255) / 256)), dim3(256), 0, s, d);
}

}  // namespace hutk
