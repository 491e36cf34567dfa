// Cycles per wave64 instruction on gfx950, by instruction kind and by the number of wavefronts per SIMD.
// Every wavefront runs K x 64 instructions of ONE kind on 8 independent accumulators (no dependency stalls within a wave) and
// reads the shader clock (s_memtime) before and after.  With W wavefronts resident per SIMD, the SIMD's issue cost of one
// instruction is (cycles a wavefront saw) / (K * 64) / W once W is large enough to hide the pipeline depth.
// build: hipcc --offload-arch=gfx950 -O2 tools/micro/valu_issue.hip -o /tmp/valu_issue    (run on the GPU box)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include <algorithm>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

#define K_ITERS 512

// 8 independent chains, 8 rounds = 64 instructions per iteration
#define BODY32(OP)                                                                                                     \
    _Pragma("unroll") for (int r = 0; r < 8; r++) {                                                                    \
        asm volatile(OP " %0, %0, %8\n" OP " %1, %1, %8\n" OP " %2, %2, %8\n" OP " %3, %3, %8\n" OP " %4, %4, %8\n" OP \
                        " %5, %5, %8\n" OP " %6, %6, %8\n" OP " %7, %7, %8\n"                                          \
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)                  \
                     : "v"(c));                                                                                        \
    }
#define BODY32_3(OP)                                                                                                    \
    _Pragma("unroll") for (int r = 0; r < 8; r++) {                                                                     \
        asm volatile(OP " %0, %0, %8, %8\n" OP " %1, %1, %8, %8\n" OP " %2, %2, %8, %8\n" OP " %3, %3, %8, %8\n" OP     \
                        " %4, %4, %8, %8\n" OP " %5, %5, %8, %8\n" OP " %6, %6, %8, %8\n" OP " %7, %7, %8, %8\n"       \
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)                   \
                     : "v"(c));                                                                                         \
    }
#define BODY_UN(OP)                                                                                                   \
    _Pragma("unroll") for (int r = 0; r < 8; r++) {                                                                   \
        asm volatile(OP " %0, %0\n" OP " %1, %1\n" OP " %2, %2\n" OP " %3, %3\n" OP " %4, %4\n" OP " %5, %5\n" OP     \
                        " %6, %6\n" OP " %7, %7\n"                                                                   \
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));               \
    }

enum { OP_ADD_U32, OP_AND, OP_LSHL, OP_MIN_U32, OP_PK_MIN_U16, OP_BFE, OP_LSHL_OR, OP_ADD3, OP_MAD_U24, OP_PERM, OP_MUL_LO, OP_FMA_F32, OP_ADD_F64, OP_MUL_F64, OP_FMA_F64, OP_RNDNE_F64, OP_CVT_F64_I32, OP_CVT_I32_F64, OP_LDS_U8, OP_CNDMASK, OP_COUNT };
static const char* const OP_NAMES[OP_COUNT] = {"v_add_u32", "v_and_b32", "v_lshlrev_b32", "v_min_u32", "v_pk_min_u16", "v_bfe_u32", "v_lshl_or_b32", "v_add3_u32", "v_mad_u32_u24", "v_perm_b32", "v_mul_lo_u32", "v_fma_f32", "v_add_f64", "v_mul_f64", "v_fma_f64",
                                               "v_rndne_f64", "v_cvt_f64_i32", "v_cvt_i32_f64", "ds_read_u8", "v_cndmask_b32"};

template <int OP>
__global__ void issue(unsigned long long* out, int seed) {
    __shared__ unsigned char lds[4096];
    const int tid = threadIdx.x;
    for (int q = tid; q < 4096; q += blockDim.x) lds[q] = (unsigned char)(q * 7 + seed);
    __syncthreads();
    unsigned long long t0 = 0, t1 = 0;
    const unsigned long long w0 = wall_clock64();  // 100 MHz
    if (OP == OP_ADD_U32 || OP == OP_PERM || OP == OP_MUL_LO || OP == OP_FMA_F32 || OP == OP_CNDMASK || OP == OP_AND || OP == OP_LSHL || OP == OP_MIN_U32 ||
        OP == OP_PK_MIN_U16 || OP == OP_BFE || OP == OP_LSHL_OR || OP == OP_ADD3 || OP == OP_MAD_U24) {
        uint32_t a0 = tid, a1 = tid + 1, a2 = tid + 2, a3 = tid + 3, a4 = tid + 4, a5 = tid + 5, a6 = tid + 6, a7 = tid + 7, c = seed | 3;
        const unsigned long long cond = 0x5555AAAA3333CCCCull ^ (unsigned long long)seed;  // lane mask of v_cndmask
        t0 = __builtin_amdgcn_s_memtime();
        for (int k = 0; k < K_ITERS; k++) {
            if (OP == OP_ADD_U32) { BODY32("v_add_u32") }
            if (OP == OP_AND) { BODY32("v_and_b32") }
            if (OP == OP_LSHL) { BODY32("v_lshlrev_b32") }
            if (OP == OP_MIN_U32) { BODY32("v_min_u32") }
            if (OP == OP_PK_MIN_U16) { BODY32("v_pk_min_u16") }
            if (OP == OP_BFE) { BODY32_3("v_bfe_u32") }
            if (OP == OP_LSHL_OR) { BODY32_3("v_lshl_or_b32") }
            if (OP == OP_ADD3) { BODY32_3("v_add3_u32") }
            if (OP == OP_MAD_U24) { BODY32_3("v_mad_u32_u24") }
            if (OP == OP_PERM) { BODY32_3("v_perm_b32") }
            if (OP == OP_MUL_LO) { BODY32("v_mul_lo_u32") }
            if (OP == OP_FMA_F32) { BODY32_3("v_fma_f32") }
            if (OP == OP_CNDMASK) {
                _Pragma("unroll") for (int r = 0; r < 8; r++) {
                    asm volatile("v_cndmask_b32 %0, %0, %8, %9\nv_cndmask_b32 %1, %1, %8, %9\nv_cndmask_b32 %2, %2, %8, %9\nv_cndmask_b32 %3, %3, %8, %9\n"
                                 "v_cndmask_b32 %4, %4, %8, %9\nv_cndmask_b32 %5, %5, %8, %9\nv_cndmask_b32 %6, %6, %8, %9\nv_cndmask_b32 %7, %7, %8, %9\n"
                                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                                 : "v"(c), "s"(cond));
                }
            }
        }
        t1 = __builtin_amdgcn_s_memtime();
        if ((a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7) == 0x12345u) out[0] = 1;
    } else if (OP == OP_LDS_U8) {
        uint32_t a0 = tid & 4095, a1 = (tid * 3) & 4095, a2 = (tid * 5) & 4095, a3 = (tid * 7) & 4095, a4 = (tid * 9) & 4095, a5 = (tid * 11) & 4095,
                 a6 = (tid * 13) & 4095, a7 = (tid * 15) & 4095;
        const uint32_t base = (uint32_t)(uintptr_t)lds;
        a0 += base; a1 += base; a2 += base; a3 += base; a4 += base; a5 += base; a6 += base; a7 += base;
        uint32_t s = 0;
        t0 = __builtin_amdgcn_s_memtime();
        for (int k = 0; k < K_ITERS; k++) {
            _Pragma("unroll") for (int r = 0; r < 8; r++) {
                uint32_t v0, v1, v2, v3, v4, v5, v6, v7;
                asm volatile("ds_read_u8 %0, %8\nds_read_u8 %1, %9\nds_read_u8 %2, %10\nds_read_u8 %3, %11\nds_read_u8 %4, %12\nds_read_u8 %5, %13\n"
                             "ds_read_u8 %6, %14\nds_read_u8 %7, %15\ns_waitcnt lgkmcnt(0)\n"
                             : "=v"(v0), "=v"(v1), "=v"(v2), "=v"(v3), "=v"(v4), "=v"(v5), "=v"(v6), "=v"(v7)
                             : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7));
                s += v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;  // (adds 8 VALU per 8 loads: the LDS row is an upper bound)
            }
        }
        t1 = __builtin_amdgcn_s_memtime();
        if (s == 0x12345u) out[0] = 1;
    } else {
        double a0 = tid, a1 = tid + 1, a2 = tid + 2, a3 = tid + 3, a4 = tid + 4, a5 = tid + 5, a6 = tid + 6, a7 = tid + 7, c = 1.0000001 + seed * 1e-9;
        uint32_t i0 = tid, i1 = tid + 1, i2 = tid + 2, i3 = tid + 3, i4 = tid + 4, i5 = tid + 5, i6 = tid + 6, i7 = tid + 7;
        t0 = __builtin_amdgcn_s_memtime();
        for (int k = 0; k < K_ITERS; k++) {
            if (OP == OP_ADD_F64) { BODY32("v_add_f64") }
            if (OP == OP_MUL_F64) { BODY32("v_mul_f64") }
            if (OP == OP_FMA_F64) { BODY32_3("v_fma_f64") }
            if (OP == OP_RNDNE_F64) { BODY_UN("v_rndne_f64") }
            if (OP == OP_CVT_F64_I32) {
                _Pragma("unroll") for (int r = 0; r < 8; r++) {
                    asm volatile("v_cvt_f64_i32 %0, %8\nv_cvt_f64_i32 %1, %9\nv_cvt_f64_i32 %2, %10\nv_cvt_f64_i32 %3, %11\nv_cvt_f64_i32 %4, %12\n"
                                 "v_cvt_f64_i32 %5, %13\nv_cvt_f64_i32 %6, %14\nv_cvt_f64_i32 %7, %15\n"
                                 : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3), "=v"(a4), "=v"(a5), "=v"(a6), "=v"(a7)
                                 : "v"(i0), "v"(i1), "v"(i2), "v"(i3), "v"(i4), "v"(i5), "v"(i6), "v"(i7));
                }
            }
            if (OP == OP_CVT_I32_F64) {
                _Pragma("unroll") for (int r = 0; r < 8; r++) {
                    asm volatile("v_cvt_i32_f64 %0, %8\nv_cvt_i32_f64 %1, %9\nv_cvt_i32_f64 %2, %10\nv_cvt_i32_f64 %3, %11\nv_cvt_i32_f64 %4, %12\n"
                                 "v_cvt_i32_f64 %5, %13\nv_cvt_i32_f64 %6, %14\nv_cvt_i32_f64 %7, %15\n"
                                 : "=v"(i0), "=v"(i1), "=v"(i2), "=v"(i3), "=v"(i4), "=v"(i5), "=v"(i6), "=v"(i7)
                                 : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7));
                }
            }
        }
        t1 = __builtin_amdgcn_s_memtime();
        if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 0.12345 || (i0 ^ i1 ^ i2 ^ i3 ^ i4 ^ i5 ^ i6 ^ i7) == 0x12345u) out[0] = 1;
    }
    if ((tid & 63) == 0) {
        const size_t wv = (size_t)blockIdx.x * (blockDim.x >> 6) + (tid >> 6);
        out[1 + 3 * wv] = t1 - t0;
        out[2 + 3 * wv] = w0;
        out[3 + 3 * wv] = wall_clock64();
    }
}

template <int OP>
static int run(unsigned long long* d, int n_cu) {
    printf("%-14s", OP_NAMES[OP]);
    for (int w = 1; w <= 8; w *= 2) {  // wavefronts per SIMD: w workgroups of 4 wavefronts per CU
        const int blocks = n_cu * w, waves = blocks * 4;
        std::vector<unsigned long long> h(3 * (size_t)waves), cyc(waves);
        double best = 1e30, span = 0, mhz = 0;
        for (int pass = 0; pass < 3; pass++) {
            issue<OP><<<dim3(blocks), dim3(256), 0, 0>>>(d, pass);
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(h.data(), d + 1, sizeof(unsigned long long) * 3 * waves, hipMemcpyDeviceToHost));
            unsigned long long first = ~0ull, last = 0;
            double ratio = 0;
            for (int q = 0; q < waves; q++) {
                cyc[q] = h[3 * q];
                first = std::min(first, h[3 * q + 1]);
                last = std::max(last, h[3 * q + 2]);
                ratio += (double)h[3 * q] / (double)(h[3 * q + 2] - h[3 * q + 1]);
            }
            std::sort(cyc.begin(), cyc.end());
            if ((double)cyc[waves / 2] < best) {
                best = (double)cyc[waves / 2];
                span = (double)(last - first);  // 100 MHz ticks from the first wavefront's start to the last one's end
                mhz = 100.0 * ratio / waves;
            }
        }
        // SIMD issue cost from the whole launch: all wavefronts' instructions over 4 SIMDs per CU and the launch's span
        const double per_simd = span * (mhz / 100.0) / ((double)K_ITERS * 64.0 * w);
        printf("  W=%d: %5.2f/wave %5.2f/SIMD", w, best / (K_ITERS * 64.0), per_simd);
        if (w == 8) printf("  (shader clock %.0f MHz)", mhz);
    }
    printf("\n");
    return 0;
}

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int n_cu = prop.multiProcessorCount;
    printf("%s, %d CUs.  Per wave64 instruction, in cycles of the shader clock (s_memtime): /wave = as one wavefront sees it (median);\n"
           "/SIMD = the launch's span x clock / (instructions per wavefront x wavefronts per SIMD), i.e. the SIMD's issue cost if the\n"
           "W workgroups of 4 wavefronts per CU ran side by side, one wavefront of each per SIMD\n", prop.name, n_cu);
    unsigned long long* d = nullptr;
    CK(hipMalloc(&d, sizeof(unsigned long long) * (1 + (size_t)n_cu * 32 * 3)));
    CK(hipMemset(d, 0, sizeof(unsigned long long) * (1 + (size_t)n_cu * 32 * 3)));
    if (run<OP_ADD_U32>(d, n_cu) || run<OP_AND>(d, n_cu) || run<OP_LSHL>(d, n_cu) || run<OP_MIN_U32>(d, n_cu) || run<OP_PK_MIN_U16>(d, n_cu) || run<OP_BFE>(d, n_cu) ||
        run<OP_LSHL_OR>(d, n_cu) || run<OP_ADD3>(d, n_cu) || run<OP_MAD_U24>(d, n_cu) || run<OP_PERM>(d, n_cu) || run<OP_CNDMASK>(d, n_cu) || run<OP_MUL_LO>(d, n_cu) || run<OP_FMA_F32>(d, n_cu) ||
        run<OP_ADD_F64>(d, n_cu) || run<OP_MUL_F64>(d, n_cu) || run<OP_FMA_F64>(d, n_cu) || run<OP_RNDNE_F64>(d, n_cu) ||
        run<OP_CVT_F64_I32>(d, n_cu) || run<OP_CVT_I32_F64>(d, n_cu) || run<OP_LDS_U8>(d, n_cu))
        return 1;
    return 0;
}
