// How much does it cost a stream to fork work onto a second stream and join it again?  (run on the GPU box)
// Result on MI355X / ROCm 7.2: 38.7 / 53.2 / 45.2 / 40.9 us per iteration for variants 0-3 (35 us of kernels).  The wait-value
// variants did NOT carry over to the library: a wait-value packet blocks the hardware queue it sits in.
//   variant 0: A -> B -> T on one stream (no second stream at all): the floor
//   variant 1: A -> [record] -> B -> [wait] -> T, side: [wait] -> C -> [record]          (events, as the library does)
//   variant 2: A -> B (B's first thread raises a flag) -> [wait ev] -> T, side: [wait value] -> C -> [record]
//   variant 3: like 2, and the join is a flag too: side: C -> F (one thread raises flag 2), main: [wait value] -> T
// build: hipcc --offload-arch=gfx950 -O2 tools/micro/fork_latency.hip -o /tmp/fork_latency
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <chrono>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void spin(int us, uint32_t* flag, uint32_t seq) {
    if (flag && blockIdx.x == 0 && threadIdx.x == 0) { __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < (long long)us * 100) {}
}
__global__ void raise(uint32_t* flag, uint32_t seq) { __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
int main() {
    hipStream_t m, s;
    CK(hipStreamCreateWithFlags(&m, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t ef, ej;
    CK(hipEventCreateWithFlags(&ef, hipEventDisableTiming | hipEventDisableSystemFence));
    CK(hipEventCreateWithFlags(&ej, hipEventDisableTiming | hipEventDisableSystemFence));
    uint32_t *f1 = nullptr, *f2 = nullptr;
    CK(hipExtMallocWithFlags((void**)&f1, 8, hipMallocSignalMemory));
    CK(hipExtMallocWithFlags((void**)&f2, 8, hipMallocSignalMemory));
    *f1 = 0; *f2 = 0;
    int can = 0;
    CK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
    printf("can use stream wait value: %d\n", can);
    const int N = 2000, A = 10, B = 20, C = 15, T = 5;  // us of spinning per kernel; 256 blocks x 64 threads each
    uint32_t seq = 0;
    for (int variant = 0; variant < 4; variant++) {
        for (int pass = 0; pass < 2; pass++) {
            CK(hipDeviceSynchronize());
            auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < N; i++) {
                seq++;
                spin<<<256, 64, 0, m>>>(A, nullptr, 0);
                if (variant == 0) {
                    spin<<<256, 64, 0, m>>>(B, nullptr, 0);
                    spin<<<256, 64, 0, m>>>(T, nullptr, 0);
                } else if (variant == 1) {
                    CK(hipEventRecord(ef, m));
                    CK(hipStreamWaitEvent(s, ef, 0));
                    spin<<<256, 64, 0, s>>>(C, nullptr, 0);
                    CK(hipEventRecord(ej, s));
                    spin<<<256, 64, 0, m>>>(B, nullptr, 0);
                    CK(hipStreamWaitEvent(m, ej, 0));
                    spin<<<256, 64, 0, m>>>(T, nullptr, 0);
                } else {
                    CK(hipStreamWaitValue32(s, f1, seq, hipStreamWaitValueGte, 0xFFFFFFFFu));
                    spin<<<256, 64, 0, s>>>(C, nullptr, 0);
                    if (variant == 2) CK(hipEventRecord(ej, s));
                    else raise<<<1, 1, 0, s>>>(f2, seq);
                    spin<<<256, 64, 0, m>>>(B, f1, seq);
                    if (variant == 2) CK(hipStreamWaitEvent(m, ej, 0));
                    else CK(hipStreamWaitValue32(m, f2, seq, hipStreamWaitValueGte, 0xFFFFFFFFu));
                    spin<<<256, 64, 0, m>>>(T, nullptr, 0);
                }
            }
            CK(hipDeviceSynchronize());
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / N;
            if (pass == 1) printf("variant %d: %.1f us per iteration (kernels alone: %d)\n", variant, us, A + B + T);
        }
    }
    return 0;
}
