// What a compute unit of gfx950 pays for one wave64 GATHER instruction that hits in its L1, by address pattern, and for one
// scalar-ALU instruction -- the two limits of k_crop_big besides vector issue (tools/micro/valu_issue.hip).
// Every wavefront issues K x 8 loads (8 in flight) from a 16 KB table; W workgroups of 4 wavefronts per CU.  Reported:
// launch span x shader clock / (loads per wavefront x wavefronts per CU) = cycles of the CU's texture-address path per
// instruction once enough wavefronts are resident.
// build: hipcc --offload-arch=gfx950 -O2 tools/micro/gather_rate.hip -o /tmp/gather_rate    (run on the GPU box)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

#define K_ITERS 256
#define TABLE 16384

enum { P_DWORD_COALESCED, P_BYTE_ONE_LINE, P_BYTE_2_LINES, P_BYTE_4_LINES, P_BYTE_8_LINES, P_BYTE_12_ROWS, P_BYTE_64_LINES, P_DWORD_12_ROWS, P_DWORD_64_LINES,
       P_BYTE_12_ROWS_16_LANES, P_DWORDX2_9_BLOCKS, P_SALU, P_COUNT };
static const char* const NAMES[P_COUNT] = {"dword, coalesced (256 B)", "byte, one 64 B line", "byte, 2 lines", "byte, 4 lines", "byte, 8 lines",
                                           "byte, 12 map rows x ~5 (a rotated 8x8 tile)", "byte, 64 lines", "dword, 12 rows", "dword, 64 lines",
                                           "byte, 12 rows, 16 lanes active", "dwordx2, 9 blocks of 8 B in 3 rows", "s_mul_i32 (scalar ALU)"};

__device__ __forceinline__ uint32_t lane_offset(int p, int lane) {
    switch (p) {
        case P_DWORD_COALESCED: return lane * 4;
        case P_BYTE_ONE_LINE: return lane;
        case P_BYTE_2_LINES: return (lane & 1) * 128 + (lane >> 1);
        case P_BYTE_4_LINES: return (lane & 3) * 128 + (lane >> 2);
        case P_BYTE_8_LINES: return (lane & 7) * 128 + (lane >> 3);
        case P_BYTE_12_ROWS:
        case P_BYTE_12_ROWS_16_LANES: return ((lane * 5) % 12) * 733 + (lane & 7) + (lane >> 4);
        case P_BYTE_64_LINES: return lane * 128 + (lane & 31);
        case P_DWORD_12_ROWS: return (((lane * 5) % 12) * 733 + (lane & 7) * 4) & ~3u;
        case P_DWORD_64_LINES: return lane * 128;
        case P_DWORDX2_9_BLOCKS: return ((lane % 9) / 3) * 736 + ((lane % 9) % 3) * 8;
    }
    return 0;
}

template <int P>
__global__ void gather(const unsigned char* table, unsigned long long* out, int seed) {
    const int lane = threadIdx.x & 63;
    uint32_t acc = 0;
    const unsigned long long w0 = wall_clock64();
    if (P == P_SALU) {
        int s0 = seed, s1 = seed + 1, s2 = seed + 2, s3 = seed + 3;
        for (int k = 0; k < K_ITERS; k++) {
#pragma unroll
            for (int r = 0; r < 2; r++)
                asm volatile("s_mul_i32 %0, %0, %4\ns_mul_i32 %1, %1, %4\ns_mul_i32 %2, %2, %4\ns_mul_i32 %3, %3, %4\n" : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : "s"(seed | 3));
        }
        acc = s0 + s1 + s2 + s3;
    } else {
        const uint32_t off = lane_offset(P, lane);
        const bool active = P != P_BYTE_12_ROWS_16_LANES || lane < 16;
        uint32_t base = (uint32_t)seed & 63u;
        for (int k = 0; k < K_ITERS; k++) {
            uint32_t v[8];
            if (active) {
#pragma unroll
                for (int q = 0; q < 8; q++) {
                    const uint32_t a = (base + q * 1472u + off) & (TABLE - 1);  // another set of rows every load
                    if (P == P_DWORD_COALESCED || P == P_DWORD_12_ROWS || P == P_DWORD_64_LINES) v[q] = *(const uint32_t*)(table + (a & ~3u));
                    else if (P == P_DWORDX2_9_BLOCKS) v[q] = (uint32_t)*(const unsigned long long*)(table + (a & ~7u));
                    else v[q] = *(table + a);
                }
#pragma unroll
                for (int q = 0; q < 8; q++) {
                    asm volatile("" : "+v"(v[q]));  // (the load stays, eight of them in flight)
                    acc += v[q];
                }
            }
            base = (base + 192u) & (TABLE - 1);
        }
    }
    const unsigned long long w1 = wall_clock64();
    if (lane == 0) {
        atomicMin(out + 0, w0);
        atomicMax(out + 1, w1);
    }
    if (acc == 0x12345u) out[2] = acc;
}

template <int P>
static int run(const unsigned char* table, unsigned long long* d, int n_cu, double clock_ghz) {
    printf("%-46s", NAMES[P]);
    for (int W : {1, 2, 4, 8}) {
        const unsigned long long init[2] = {~0ull, 0ull};
        double best = 1e30;
        for (int rep = 0; rep < 3; rep++) {
            CK(hipMemcpy(d, init, sizeof(init), hipMemcpyHostToDevice));
            gather<P><<<n_cu * W, 256>>>(table, d, 1 + rep);
            CK(hipDeviceSynchronize());
            unsigned long long o[2];
            CK(hipMemcpy(o, d, sizeof(o), hipMemcpyDeviceToHost));
            const double span_ns = (double)(o[1] - o[0]) * 10.0;  // wall clock: 100 MHz
            best = span_ns < best ? span_ns : best;
        }
        const double per_cu = best * clock_ghz / ((double)K_ITERS * 8 * 4 * W);
        printf("  W=%d: %6.2f", W, per_cu);
    }
    printf("   cycles / instruction / CU\n");
    return 0;
}

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int n_cu = prop.multiProcessorCount;
    const double clock_ghz = prop.clockRate * 1e-6;
    printf("%s, %d CUs, %.2f GHz (nominal: the cycle figures assume it).  W workgroups of 4 wavefronts per CU.\n", prop.name, n_cu, clock_ghz);
    unsigned char* table = nullptr;
    unsigned long long* d = nullptr;
    CK(hipMalloc(&table, TABLE + 4096));
    CK(hipMemset(table, 1, TABLE + 4096));
    CK(hipMalloc(&d, 64));
    if (run<P_DWORD_COALESCED>(table, d, n_cu, clock_ghz) || run<P_BYTE_ONE_LINE>(table, d, n_cu, clock_ghz) || run<P_BYTE_2_LINES>(table, d, n_cu, clock_ghz) ||
        run<P_BYTE_4_LINES>(table, d, n_cu, clock_ghz) || run<P_BYTE_8_LINES>(table, d, n_cu, clock_ghz) || run<P_BYTE_12_ROWS>(table, d, n_cu, clock_ghz) ||
        run<P_BYTE_64_LINES>(table, d, n_cu, clock_ghz) || run<P_DWORD_12_ROWS>(table, d, n_cu, clock_ghz) || run<P_DWORD_64_LINES>(table, d, n_cu, clock_ghz) ||
        run<P_BYTE_12_ROWS_16_LANES>(table, d, n_cu, clock_ghz) || run<P_DWORDX2_9_BLOCKS>(table, d, n_cu, clock_ghz) || run<P_SALU>(table, d, n_cu, clock_ghz))
        return 1;
    return 0;
}
