// How fast does the dispatcher fill the chip with small workgroups?  8192 wavefronts of work, each holding 5 KB of LDS and
// spinning for a fixed time, launched as 8192 x 64, 4096 x 128, 2048 x 256 or 1024 x 512 threads: kernel time and when the
// wavefronts started.  (run on the GPU box)
// build: hipcc --offload-arch=gfx950 -O2 tools/micro/launch_ramp.hip -o /tmp/launch_ramp
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <algorithm>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void spin(int us, unsigned long long* start, int waves_per_block) {
    extern __shared__ unsigned char lds[];
    const int wave = blockIdx.x * waves_per_block + (threadIdx.x >> 6);
    const long long t0 = wall_clock64();
    if ((threadIdx.x & 63) == 0) start[wave] = (unsigned long long)t0;
    lds[threadIdx.x] = (unsigned char)t0;  // the allocation is real
    while (wall_clock64() - t0 < (long long)us * 100) {}
    if (lds[threadIdx.x] == 7 && us < 0) start[0] = 0;
}
int main() {
    const int W = 8192, US = 30;
    unsigned long long* d = nullptr;
    CK(hipMalloc(&d, sizeof(unsigned long long) * W));
    std::vector<unsigned long long> h(W);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int wpb = 1; wpb <= 8; wpb *= 2) {
        for (int pass = 0; pass < 3; pass++) {
            CK(hipEventRecord(e0, 0));
            spin<<<dim3(W / wpb), dim3(64 * wpb), 5064 * wpb, 0>>>(US, d, wpb);
            CK(hipEventRecord(e1, 0));
            CK(hipDeviceSynchronize());
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            CK(hipMemcpy(h.data(), d, sizeof(unsigned long long) * W, hipMemcpyDeviceToHost));
            std::sort(h.begin(), h.end());
            if (pass == 2)
                printf("%4d x %3d threads: kernel %.1f us (spin %d); wave starts after the first: p50 %.1f p90 %.1f p99 %.1f max %.1f us\n", W / wpb,
                       64 * wpb, ms * 1e3, US, (h[W / 2] - h[0]) / 100.0, (h[W * 9 / 10] - h[0]) / 100.0, (h[W * 99 / 100] - h[0]) / 100.0,
                       (h[W - 1] - h[0]) / 100.0);
        }
    }
    return 0;
}
