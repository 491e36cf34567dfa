// dep_gap.hip -- what does a stream pay between two DEPENDENT kernels?  A (small or large grid, short or long) -> B (8192 one-wavefront
// workgroups) -> C (the same), back to back on one stream, `reps` times; run under `rocprofv3 --kernel-trace` and read the gaps
// between A's end and B's start, B's end and C's start (tools/micro/dep_gap.sh prints them).  Spin kernels: duration = cycles asked for.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/dep_gap.hip -o /tmp/dep_gap && /tmp/dep_gap <a_blocks> <a_us> [reps]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ void kA(unsigned long long ticks, int* sink) {
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(2);
    if (sink && threadIdx.x == 0 && blockIdx.x == 0) sink[0] = 1;
}
__global__ void kB(unsigned long long ticks, int* sink) {
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(2);
    if (sink && threadIdx.x == 0 && blockIdx.x == 0) sink[1] = 1;
}
__global__ void kC(unsigned long long ticks, int* sink) {
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(2);
    if (sink && threadIdx.x == 0 && blockIdx.x == 0) sink[2] = 1;
}
int main(int argc, char** argv) {
    const int a_blocks = argc > 1 ? atoi(argv[1]) : 36;
    const double a_us = argc > 2 ? atof(argv[2]) : 7.0;
    const int reps = argc > 3 ? atoi(argv[3]) : 200;
    int* sink = nullptr;
    hipMalloc(&sink, 64);
    hipStream_t st;
    hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int r = 0; r < 20; r++) {
        kA<<<a_blocks, 256, 0, st>>>((unsigned long long)(a_us * 100), sink);
        kB<<<8192, 64, 0, st>>>(1500ull, sink);
        kC<<<8192, 64, 0, st>>>(4000ull, sink);
    }
    hipStreamSynchronize(st);
    hipEventRecord(e0, st);
    for (int r = 0; r < reps; r++) {
        kA<<<a_blocks, 256, 0, st>>>((unsigned long long)(a_us * 100), sink);
        kB<<<8192, 64, 0, st>>>(1500ull, sink);
        kC<<<8192, 64, 0, st>>>(4000ull, sink);
    }
    hipEventRecord(e1, st);
    hipStreamSynchronize(st);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    printf("A %d blocks x %.0f us -> B 15 us -> C 40 us: %.1f us per round (%.1f of them are the kernels' own spins)\n", a_blocks, a_us, 1e3 * ms / reps, a_us + 15 + 40);
    return 0;
}
