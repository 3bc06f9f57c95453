// valu_issue_bench.hip -- what does one wave64 instruction cost a gfx950 SIMD?
//
// bench.py prices k_tiles' instruction stream against issue slots; MI355X_MICROARCH.md says a wave64 VALU instruction
// takes 2 cycles of a SIMD-32 and that one wave alone sustains one per 4.  This program measures it for the integer
// opcodes k_tiles is made of (tools/isa_histogram.py lists them), with 1, 2, 4 and 8 wavefronts per SIMD, as
// independent streams (8 accumulators round-robin) and as one dependent chain:
//
//     cycles per wave-instruction per SIMD = kernel time x clock / (instructions per wave x waves per SIMD)
//
// Grid = 256 CUs x k workgroups of 256 threads (one wavefront per SIMD each); k workgroups per CU are forced by the
// dynamic LDS size (160 KiB / k).  Every wavefront runs REPS x 64 copies of the instruction.  The clock is taken from
// the same launch: s_memtime ticks of a wavefront's loop over the kernel's event time at k = 1 (one tick = one shader
// cycle).  Output: one JSON line per (opcode, form, k); tools/summarize_issue.py turns them into the cost table.
//
//   hipcc --offload-arch=gfx950 -O2 tools/valu_issue_bench.hip -o tools/bin/valu_issue_bench
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CHECK(x)                                                                      \
    do {                                                                              \
        hipError_t e_ = (x);                                                          \
        if (e_ != hipSuccess) {                                                       \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            exit(1);                                                                  \
        }                                                                             \
    } while (0)

constexpr int REPS = 2048;  // loop trips; 64 instructions per trip

struct Stamp { unsigned long long t0, t1; };

template <typename AccT, class Body>
__device__ __forceinline__ void run(Stamp* out, uint32_t seed, Body body) {
    extern __shared__ uint32_t lds[];
    AccT a[8];
#pragma unroll
    for (int i = 0; i < 8; i++) a[i] = (AccT)(seed * (2 * i + 1) + threadIdx.x);
    if (seed == 0xFFFFFFFFu) lds[threadIdx.x] = (uint32_t)a[0];  // keeps the LDS allocation alive
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < REPS; r++) body(a);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    AccT x = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) x ^= a[i];
    if (x == (AccT)0x12345679u) out[0].t0 = (unsigned long long)x;  // the results are "used"
    if ((threadIdx.x & 63) == 0) {
        Stamp s{t0, t1};
        out[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = s;
    }
}

// OP(A) = the instruction's text with accumulator operand A; %8 and %9 are two more VGPR inputs.
// s[20:21] / s22 are scratch scalars of the forms that write or read one.
#define INDEP8(OP) ".rept 8\n" OP("%0") OP("%1") OP("%2") OP("%3") OP("%4") OP("%5") OP("%6") OP("%7") ".endr\n"
#define DEP64(OP) ".rept 64\n" OP("%0") ".endr\n"
#define DEF(NAME, ACC, OP)                                                                                             \
    __global__ void k_##NAME##_indep(Stamp* out, uint32_t seed) {                                                      \
        const ACC b = (seed & 7u) | 1u, c = seed >> 3;                                                                 \
        run<ACC>(out, seed, [&](ACC* a) {                                                                              \
            asm volatile("s_mov_b32 s22, 3\n" INDEP8(OP)                                                               \
                         : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) \
                         : "v"(b), "v"(c)                                                                              \
                         : "vcc", "s20", "s21", "s22", "scc");                                                         \
        });                                                                                                            \
    }                                                                                                                  \
    __global__ void k_##NAME##_dep(Stamp* out, uint32_t seed) {                                                        \
        const ACC b = (seed & 7u) | 1u, c = seed >> 3;                                                                 \
        run<ACC>(out, seed, [&](ACC* a) {                                                                              \
            asm volatile("s_mov_b32 s22, 3\n" DEP64(OP)                                                                \
                         : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) \
                         : "v"(b), "v"(c)                                                                              \
                         : "vcc", "s20", "s21", "s22", "scc");                                                         \
        });                                                                                                            \
    }

// ---- 32-bit VOP1 / VOP2 ----
#define OP_mov(A) "v_mov_b32 " A ", " A "\n"
#define OP_not(A) "v_not_b32 " A ", " A "\n"
#define OP_ffbl(A) "v_ffbl_b32 " A ", " A "\n"
#define OP_ffbh(A) "v_ffbh_u32 " A ", " A "\n"
#define OP_bfrev(A) "v_bfrev_b32 " A ", " A "\n"
#define OP_xor(A) "v_xor_b32 " A ", " A ", %8\n"
#define OP_and(A) "v_and_b32 " A ", " A ", %8\n"
#define OP_or(A) "v_or_b32 " A ", " A ", %8\n"
#define OP_and_lit(A) "v_and_b32 " A ", 0x7ffe1, " A "\n"
#define OP_add(A) "v_add_u32 " A ", " A ", %8\n"
#define OP_sub(A) "v_sub_u32 " A ", " A ", %8\n"
#define OP_min(A) "v_min_u32 " A ", " A ", %8\n"
#define OP_max(A) "v_max_u32 " A ", " A ", %8\n"
#define OP_lshl(A) "v_lshlrev_b32 " A ", %8, " A "\n"
#define OP_lshr(A) "v_lshrrev_b32 " A ", %8, " A "\n"
#define OP_ashr(A) "v_ashrrev_i32 " A ", %8, " A "\n"
#define OP_mul24(A) "v_mul_u32_u24 " A ", " A ", %8\n"
#define OP_cndmask(A) "v_cndmask_b32 " A ", " A ", %8, vcc\n"
#define OP_addf(A) "v_add_f32 " A ", " A ", %8\n"
#define OP_fmaf(A) "v_fma_f32 " A ", " A ", %8, %9\n"
// ---- VOP3 ----
#define OP_mullo(A) "v_mul_lo_u32 " A ", " A ", %8\n"
#define OP_mulhi(A) "v_mul_hi_u32 " A ", " A ", %8\n"
#define OP_lshl_or(A) "v_lshl_or_b32 " A ", " A ", %8, %9\n"
#define OP_lshl_add(A) "v_lshl_add_u32 " A ", " A ", %8, %9\n"
#define OP_add_lshl(A) "v_add_lshl_u32 " A ", " A ", %8, %9\n"
#define OP_add3(A) "v_add3_u32 " A ", " A ", %8, %9\n"
#define OP_and_or(A) "v_and_or_b32 " A ", " A ", %8, %9\n"
#define OP_or3(A) "v_or3_b32 " A ", " A ", %8, %9\n"
#define OP_xad(A) "v_xad_u32 " A ", " A ", %8, %9\n"
#define OP_bitop3(A) "v_bitop3_b32 " A ", " A ", %8, %9 bitop3:0x96\n"
#define OP_bfe(A) "v_bfe_u32 " A ", " A ", %8, 5\n"
#define OP_bfi(A) "v_bfi_b32 " A ", " A ", %8, %9\n"
#define OP_alignbit(A) "v_alignbit_b32 " A ", " A ", %8, %9\n"
#define OP_perm(A) "v_perm_b32 " A ", " A ", %8, %9\n"
#define OP_min3(A) "v_min3_u32 " A ", " A ", %8, %9\n"
#define OP_mad24(A) "v_mad_u32_u24 " A ", " A ", %8, %9\n"
#define OP_bcnt(A) "v_bcnt_u32_b32 " A ", " A ", %8\n"
#define OP_mbcnt(A) "v_mbcnt_lo_u32_b32 " A ", " A ", %8\n"
#define OP_cndmask64(A) "v_cndmask_b32_e64 " A ", " A ", %8, s[20:21]\n"
#define OP_lshl_e64(A) "v_lshlrev_b32_e64 " A ", 3, " A "\n"
// ---- compares: VOPC writes VCC, the e64 form an SGPR pair; the accumulator is only read ----
#define OP_cmp(A) "v_cmp_gt_u32 vcc, " A ", %8\n"
#define OP_cmp64(A) "v_cmp_gt_u32_e64 s[20:21], " A ", %8\n"
#define OP_cmp_sel(A) "v_cmp_lt_u32 vcc, " A ", %8\nv_cndmask_b32 " A ", " A ", %8, vcc\n"
// ---- lane crossing ----
#define OP_readlane(A) "v_readlane_b32 s20, " A ", 5\n"
#define OP_readfirstlane(A) "v_readfirstlane_b32 s20, " A "\n"
#define OP_writelane(A) "v_writelane_b32 " A ", s22, 5\n"
#define OP_dpp_add(A) "v_add_u32_dpp " A ", " A ", " A " row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define OP_dpp_mov(A) "v_mov_b32_dpp " A ", " A " row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define OP_dpp_bcast(A) "v_add_u32_dpp " A ", " A ", " A " row_bcast:15 row_mask:0xa bank_mask:0xf\n"
#define OP_sdwa_lshl(A) "v_lshlrev_b32_sdwa " A ", %8, " A " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n"
#define OP_sdwa_and(A) "v_and_b32_sdwa " A ", " A ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n"
// ---- 64-bit ----
#define OP_lshr64(A) "v_lshrrev_b64 " A ", 0, " A "\n"
#define OP_lshl64(A) "v_lshlrev_b64 " A ", 0, " A "\n"
#define OP_mov64(A) "v_mov_b64 " A ", " A "\n"
#define OP_lshl_add64(A) "v_lshl_add_u64 " A ", " A ", 0, %8\n"
#define OP_mad64(A) "v_mad_u64_u32 " A ", s[20:21], 3, 5, " A "\n"
#define OP_add_co(A) "v_add_co_u32 " A ", vcc, " A ", %8\n"
#define OP_addc(A) "v_addc_co_u32 " A ", vcc, " A ", %8, vcc\n"
// ---- scalar ----
#define OP_s_add(A) "s_add_u32 s20, s20, s22\n"
#define OP_s_and64(A) "s_and_b64 s[20:21], s[20:21], exec\n"
#define OP_s_mix(A) "v_xor_b32 " A ", " A ", %8\ns_add_u32 s20, s20, 1\n"
#define OP_s_mix2(A) "v_lshlrev_b32 " A ", %8, " A "\ns_add_u32 s20, s20, 1\n"
#define OP_s_nop(A) "s_nop 0\n"
// ---- mixes of a "2-cycle" and a "4-cycle" opcode ----
#define OP_mix_fast_slow(A) "v_xor_b32 " A ", " A ", %8\nv_lshlrev_b32 " A ", %8, " A "\n"
// ---- LDS ----
#define OP_lds_u16(A) "ds_read_u16 " A ", " A "\ns_waitcnt lgkmcnt(0)\nv_and_b32 " A ", 0x7fe, " A "\n"
#define OP_lds_b32(A) "ds_read_b32 %9, %8 offset:256\n"

#define ALL32(X)                                                                                                          \
    X(mov, 1) X(not, 1) X(ffbl, 1) X(ffbh, 1) X(bfrev, 1) X(xor, 1) X(and, 1) X(or, 1) X(and_lit, 1) X(add, 1) X(sub, 1) X(min, 1) \
    X(max, 1) X(lshl, 1) X(lshr, 1) X(ashr, 1) X(mul24, 1) X(cndmask, 1) X(addf, 1) X(fmaf, 1) X(mullo, 1) X(mulhi, 1)   \
    X(lshl_or, 1) X(lshl_add, 1) X(add_lshl, 1) X(add3, 1) X(and_or, 1) X(or3, 1) X(xad, 1) X(bitop3, 1) X(bfe, 1) X(bfi, 1) \
    X(alignbit, 1) X(perm, 1) X(min3, 1) X(mad24, 1) X(bcnt, 1) X(mbcnt, 1) X(cndmask64, 1) X(lshl_e64, 1) X(cmp, 1)     \
    X(cmp64, 1) X(cmp_sel, 2) X(readlane, 1) X(readfirstlane, 1) X(writelane, 1) X(dpp_add, 1) X(dpp_mov, 1)             \
    X(dpp_bcast, 1) X(sdwa_lshl, 1) X(sdwa_and, 1) X(add_co, 1) X(addc, 1) X(s_add, 1) X(s_and64, 1) X(s_mix, 2)         \
    X(s_mix2, 2) X(s_nop, 1) X(mix_fast_slow, 2) X(lds_u16, 3)
#define ALL64(X) X(lshr64, 1) X(lshl64, 1) X(mov64, 1) X(lshl_add64, 1) X(mad64, 1)

#define GEN32(N, K) DEF(N, uint32_t, OP_##N)
#define GEN64(N, K) DEF(N, uint64_t, OP_##N)
ALL32(GEN32)
ALL64(GEN64)

struct Case { const char* name; const char* form; void (*fn)(Stamp*, uint32_t); int insts_per_trip; };
#define CASES(N, K) {#N, "indep", k_##N##_indep, 64 * K}, {#N, "dep", k_##N##_dep, 64 * K},

int main(int argc, char** argv) {
    CHECK(hipSetDevice(0));
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, 0));
    const int n_cu = p.multiProcessorCount;
    std::vector<Case> cases = {ALL32(CASES) ALL64(CASES)};
    Stamp* d_out;
    const int max_waves = n_cu * 8 * 4;
    CHECK(hipMalloc(&d_out, sizeof(Stamp) * max_waves));
    std::vector<Stamp> h(max_waves);
    printf("{\"device\": \"%s\", \"cus\": %d, \"max_clock_khz\": %d, \"reps\": %d}\n", p.gcnArchName, n_cu, p.clockRate, REPS);
    for (const Case& c : cases) {
        if (argc > 1 && !strstr(argv[1], c.name)) continue;
        double clock_ghz = 0;
        for (int k : {1, 2, 4, 8}) {
            const size_t lds = (size_t)(160 * 1024) / k - (k == 1 ? 0 : 512);
            CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(c.fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            const int grid = n_cu * k;
            hipLaunchKernelGGL(c.fn, dim3(grid), dim3(256), lds, 0, d_out, 12345u);
            CHECK(hipDeviceSynchronize());
            hipEvent_t e0, e1;
            CHECK(hipEventCreate(&e0));
            CHECK(hipEventCreate(&e1));
            float best = 1e30f;
            for (int rep = 0; rep < 3; rep++) {
                CHECK(hipEventRecord(e0, 0));
                hipLaunchKernelGGL(c.fn, dim3(grid), dim3(256), lds, 0, d_out, 777u);
                CHECK(hipEventRecord(e1, 0));
                CHECK(hipDeviceSynchronize());
                float ms = 0;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                best = std::min(best, ms);
            }
            const int n_waves = grid * 4;
            CHECK(hipMemcpy(h.data(), d_out, sizeof(Stamp) * n_waves, hipMemcpyDeviceToHost));
            std::vector<double> life(n_waves);
            for (int i = 0; i < n_waves; i++) life[i] = (double)(h[i].t1 - h[i].t0);
            std::sort(life.begin(), life.end());
            const double med = life[n_waves / 2];
            const double insts = (double)REPS * c.insts_per_trip;
            const double ns = best * 1e6;
            if (k == 1) clock_ghz = med / ns;  // ticks of the loop over the launch's time (launch overhead ~1 %)
            printf("{\"op\": \"%s\", \"form\": \"%s\", \"waves_per_simd\": %d, \"insts_per_wave\": %.0f, \"median_wave_ticks\": %.0f, "
                   "\"kernel_us\": %.2f, \"ns_per_inst_per_simd\": %.4f, \"clock_ghz_k1\": %.3f, \"cycles_per_inst_per_simd\": %.3f}\n",
                   c.name, c.form, k, insts, med, ns * 1e-3, ns / (insts * k), clock_ghz, ns / (insts * k) * clock_ghz);
            CHECK(hipEventDestroy(e0));
            CHECK(hipEventDestroy(e1));
        }
    }
    CHECK(hipFree(d_out));
    return 0;
}
