// Micro-benchmark behind the sweep's record layout: what does a wavefront pay for gathering 64 scattered
// 64-byte records, by access shape?  (hipcc --offload-arch=gfx950 -O3 scripts/gather_probe.hip -o /tmp/gather_probe)
//   V1  one lane per record, four 16-byte loads per lane (the sweep's shape in round 1)
//   V2  four lanes per record, one 16-byte load per lane, four instructions for 64 records
//   V3  one lane per record, three loads (48 of the 64 bytes)
//   V4  one lane per record, two loads (32-byte records)
//   V5  one lane per record, one load (16-byte records)
// Records sit in a table of N x 64 bytes read in random order; `dep` makes every gather's indices depend
// on the previous gather's data (a chain, as a level step depends on the level below).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int V, bool DEP>
__global__ __launch_bounds__(256) void k_gather(const uint4 *__restrict__ tab, const int *__restrict__ idx, int n_rec, int iters,
                                                unsigned *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int *my = idx + (size_t)wave * iters * 64;
    unsigned acc = 0;
    unsigned carry = 0;
    for (int it = 0; it < iters; ++it) {
        int r = my[it * 64 + lane];
        if (DEP) r = (int)((unsigned)(r + (carry & 1023u)) % (unsigned)n_rec);
        uint4 a = make_uint4(0, 0, 0, 0), b = a, c = a, d = a;
        if (V == 2) {
            const int sub = lane & 3, grp = lane >> 2;
            const int r0 = __shfl(r, grp, 64), r1 = __shfl(r, grp + 16, 64), r2 = __shfl(r, grp + 32, 64), r3 = __shfl(r, grp + 48, 64);
            a = tab[(size_t)r0 * 4 + sub];
            b = tab[(size_t)r1 * 4 + sub];
            c = tab[(size_t)r2 * 4 + sub];
            d = tab[(size_t)r3 * 4 + sub];
        } else {
            const uint4 *p = tab + (size_t)r * 4;
            a = p[0];
            if (V == 1 || V == 3 || V == 4) b = p[1];
            if (V == 1 || V == 3) c = p[2];
            if (V == 1) d = p[3];
        }
        const unsigned s = a.x ^ a.y ^ a.z ^ a.w ^ b.x ^ b.y ^ b.z ^ b.w ^ c.x ^ c.y ^ c.z ^ c.w ^ d.x ^ d.y ^ d.z ^ d.w;
        acc += s;
        if (DEP) carry = s;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int V, bool DEP>
float run(const uint4 *tab, const int *idx, int n_rec, int iters, unsigned *out, int wgs) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k_gather<V, DEP>), dim3(wgs), dim3(256), 0, 0, tab, idx, n_rec, iters, out);
    CHECK(hipEventRecord(e0, 0));
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((k_gather<V, DEP>), dim3(wgs), dim3(256), 0, 0, tab, idx, n_rec, iters, out);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    return ms / 5;
}

int main() {
    const int iters = 400;
    for (int n_rec : {20000, 400000, 8000000}) {
        for (int wgs : {512, 1024}) {  // 2 or 4 wavefronts per SIMD on 256 CUs
            const int waves = wgs * 4;
            std::vector<int> h((size_t)waves * iters * 64);
            unsigned s = 12345u;
            for (auto &v : h) { s = s * 1664525u + 1013904223u; v = (int)((s >> 8) % (unsigned)n_rec); }
            uint4 *tab; int *idx; unsigned *out;
            CHECK(hipMalloc(&tab, (size_t)n_rec * 64));
            CHECK(hipMemset(tab, 1, (size_t)n_rec * 64));
            CHECK(hipMalloc(&idx, h.size() * 4));
            CHECK(hipMemcpy(idx, h.data(), h.size() * 4, hipMemcpyHostToDevice));
            CHECK(hipMalloc(&out, (size_t)waves * 64 * 4));
            const double gathers = (double)waves * iters;  // wave-level gathers of 64 records per launch
#define ROW(V, DEP) { float ms = run<V, DEP>(tab, idx, n_rec, iters, out, wgs); \
            printf("table %8.1f MB  waves/SIMD %d  V%d %s : %.3f ms  %.0f ns per 64-record gather per wave, %.2f G records/s\n", n_rec * 64.0 / 1e6, \
                   wgs / 256, V, DEP ? "chained" : "free   ", ms, ms * 1e6 / iters, gathers * 64 / (ms * 1e-3) / 1e9); }
            ROW(1, false) ROW(2, false) ROW(3, false) ROW(4, false) ROW(5, false)
            ROW(1, true) ROW(2, true) ROW(4, true) ROW(5, true)
            CHECK(hipFree(tab)); CHECK(hipFree(idx)); CHECK(hipFree(out));
        }
    }
    return 0;
}
