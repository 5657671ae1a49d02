// Micro-benchmark behind k_select_stream (round 3): what does the chip deliver for the C5 table's access shapes?
// A table of R rows x C fp64 (4 096 x 200 000 = 6.55 GB) is read once; every lane keeps `ring` 16-byte loads in flight.
//   rows      one 256-thread workgroup per row, each wavefront a contiguous quarter of it (k_select_stream's shape)
//   rows-il   one workgroup per row, the four wavefronts interleaved in 1-KB pieces (one stream per workgroup)
//   flat      the table as one array, grid-stride over 1-KB pieces (one stream per chip)
// (hipcc --offload-arch=gfx950 -O3 scripts/stream_probe.hip -o scripts/bin/stream_probe)
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int RING, int MODE>
__global__ __launch_bounds__(256) void k_stream(const double2 *__restrict__ tab, long rows, long cols2, double *__restrict__ out) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    double acc = 0;
    if (MODE == 2) {  // flat
        const long n = rows * cols2;
        const long stride = (long)gridDim.x * 256;
        long i = (long)blockIdx.x * 256 + threadIdx.x;
        double2 ring[RING];
#pragma unroll
        for (int u = 0; u < RING; ++u) ring[u] = tab[min(i + u * stride, n - 1)];
        for (; i < n; i += RING * stride) {
#pragma unroll
            for (int u = 0; u < RING; ++u) {
                const double2 c = ring[u];
                ring[u] = tab[min(i + (u + RING) * stride, n - 1)];
                acc += (c.x >= 0 && c.x <= 0.2) + (c.y >= 0 && c.y <= 0.2);
            }
        }
    } else {
        for (long r = blockIdx.x; r < rows; r += gridDim.x) {
            const double2 *row = tab + r * cols2;
            long lo, hi, step, first;
            if (MODE == 0) { const long per = (cols2 + 3) / 4; lo = per * wv; hi = min(cols2, per * (wv + 1)); step = 64; first = lo + lane; }
            else { lo = 0; hi = cols2; step = 256; first = wv * 64 + lane; }
            double2 ring[RING];
#pragma unroll
            for (int u = 0; u < RING; ++u) ring[u] = row[min(first + u * step, hi - 1)];
            for (long i = first; i < hi; i += RING * step) {
#pragma unroll
                for (int u = 0; u < RING; ++u) {
                    const double2 c = ring[u];
                    ring[u] = row[min(i + (u + RING) * step, hi - 1)];
                    acc += (c.x >= 0 && c.x <= 0.2) + (c.y >= 0 && c.y <= 0.2);
                }
            }
        }
    }
    if (acc == 12345.678) out[0] = acc;
}

template <int RING, int MODE>
void run(const char *name, const double2 *tab, long rows, long cols2, double *out, int grid) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k_stream<RING, MODE>), dim3(grid), dim3(256), 0, 0, tab, rows, cols2, out);
    CHECK(hipEventRecord(e0));
    for (int w = 0; w < 5; ++w) hipLaunchKernelGGL((k_stream<RING, MODE>), dim3(grid), dim3(256), 0, 0, tab, rows, cols2, out);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    ms /= 5;
    printf("%-8s ring %2d grid %5d: %.3f ms  %.0f GB/s\n", name, RING, grid, ms, rows * cols2 * 16.0 / (ms * 1e-3) / 1e9);
}

int main() {
    const long rows = 4096, cols = 200000, cols2 = cols / 2;
    double2 *tab; double *out;
    CHECK(hipMalloc(&tab, rows * cols2 * 16));
    CHECK(hipMalloc(&out, 8));
    CHECK(hipMemset(tab, 0x3f, rows * cols2 * 16));
    for (int grid : {1024, 2048, 4096}) {
        run<4, 0>("rows", tab, rows, cols2, out, grid);
        run<8, 0>("rows", tab, rows, cols2, out, grid);
        run<4, 1>("rows-il", tab, rows, cols2, out, grid);
        run<8, 1>("rows-il", tab, rows, cols2, out, grid);
        run<4, 2>("flat", tab, rows, cols2, out, grid);
        run<8, 2>("flat", tab, rows, cols2, out, grid);
    }
    return 0;
}
