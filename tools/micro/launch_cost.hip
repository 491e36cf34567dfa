// Host-side cost of a kernel launch by the size of its by-value argument (DevWorld is ~3.5 KB) -- is the step's host issue
// time (~7 us per launch) the argument copy?   build: hipcc --offload-arch=gfx950 -O2 tools/micro/launch_cost.hip -o /tmp/launch_cost
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
template <int N> struct Arg { int v[N / 4]; };
template <int N> __global__ void k(Arg<N> a, int* out) { if (a.v[0] == 12345 && a.v[N / 4 - 1] == 1) out[0] = 1; }
template <int N> static int run(int* d, hipStream_t st) {
    Arg<N> a; for (int i = 0; i < N / 4; i++) a.v[i] = i;
    for (int i = 0; i < 200; i++) k<N><<<1, 64, 0, st>>>(a, d);
    CK(hipStreamSynchronize(st));
    const int n = 4000;
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < n; i++) k<N><<<1, 64, 0, st>>>(a, d);
    auto t1 = std::chrono::steady_clock::now();
    CK(hipStreamSynchronize(st));
    auto t2 = std::chrono::steady_clock::now();
    printf("argument %5d bytes: host issue %.2f us / launch, end to end %.2f us / launch\n", N, std::chrono::duration<double, std::micro>(t1 - t0).count() / n,
           std::chrono::duration<double, std::micro>(t2 - t0).count() / n);
    return 0;
}
int main() {
    int* d; CK(hipMalloc(&d, 64));
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    if (run<64>(d, st) || run<512>(d, st) || run<1024>(d, st) || run<2048>(d, st) || run<3584>(d, st) || run<4032>(d, st)) return 1;
    return 0;
}
