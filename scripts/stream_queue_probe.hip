// Which of a process's streams share a hardware queue?  A spinning kernel (one workgroup, ~1 ms) on stream i and another on
// stream j: 1 ms when they run side by side, 2 ms when the two streams sit on one queue.
// hipcc --offload-arch=gfx950 -O2 scripts/stream_queue_probe.hip -o /tmp/sqp && /tmp/sqp
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void spin(long long cycles, int *sink) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) {}
    if (sink && cycles < 0) *sink = 1;
}
int main(int argc, char **argv) {
    const int N = 7;
    int *d;
    hipMalloc(&d, 4);
    if (argc > 1) hipMemset(d, 0, 4);  // any argument: the null stream works first
    hipStream_t s[N];
    for (int i = 0; i < N; ++i) hipStreamCreate(&s[i]);
    for (int i = 0; i < N; ++i) { hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s[i], 1000, nullptr); }
    hipDeviceSynchronize();
    const long long ms1 = 100000;  // wall_clock64 ticks at 100 MHz: 1 ms
    printf("pair times (ms), streams in creation order%s:\n", argc > 1 ? ", null stream used first" : "");
    for (int i = 0; i < N; ++i) {
        for (int j = 0; j < N; ++j) {
            if (j <= i) { printf("   . "); continue; }
            auto t0 = std::chrono::steady_clock::now();
            hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s[i], ms1, nullptr);
            hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s[j], ms1, nullptr);
            hipDeviceSynchronize();
            printf("%4.1f ", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
        }
        printf("\n");
    }
    return 0;
}
