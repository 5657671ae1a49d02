// Bare fp4 MFMA loops on random operands, one wavefront per SIMD and two (512-thread workgroups, one per CU): the clock the chip
// holds under v_mfma_scale_f32_32x32x64_f8f6f4 against v_mfma_scale_f32_16x16x128_f8f6f4 at equal FLOPs per wavefront
// (MI355X_MICROARCH.md, DVFS item 7, measured that for bf16).  Prints TFLOP/s for both shapes.
//   hipcc --offload-arch=gfx950 -O3 scripts/mfma_shape_probe.hip -o scripts/bin/mfma_shape_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));

template <int SHAPE>
__global__ __launch_bounds__(512, 2) void k(const int *__restrict__ in, float *__restrict__ out, int iters) {
    const int tid = threadIdx.x + blockIdx.x * blockDim.x;
    v8i a[2], b[4];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 4; ++j) a[i][j] = in[(tid * 8 + i * 4 + j) & 0xffff];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) b[i][j] = in[(tid * 16 + i * 4 + j + 77) & 0xffff];
    for (int i = 0; i < 2; ++i) for (int j = 4; j < 8; ++j) a[i][j] = 0;
    for (int i = 0; i < 4; ++i) for (int j = 4; j < 8; ++j) b[i][j] = 0;
    if (SHAPE == 32 || SHAPE == 326) {
        // (SHAPE 326: the first operand as fp6 -- 6 of its 8 dwords in use -- against fp4: the mixed form of dist_sd.hip's F6 filter)
        if (SHAPE == 326) for (int i = 0; i < 2; ++i) { a[i][4] = a[i][0] ^ 0x1111; a[i][5] = a[i][1] ^ 0x2222; }
        v16f acc[2][4];
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 4; ++j) for (int x = 0; x < 16; ++x) acc[i][j][x] = 0.f;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = SHAPE == 326 ? __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[i], b[j], acc[i][j], 2, 4, 0, 0, 0, 0)
                                             : __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[i], b[j], acc[i][j], 4, 4, 0, 0, 0, 0);
        float s = 0;
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 4; ++j) for (int x = 0; x < 16; ++x) s += acc[i][j][x];
        out[tid] = s;
    } else {
        // the same 64 x 128 wavefront tile as 4 x 8 tiles of 16 x 16: 32 MFMAs of K = 128 per 128 K values (16 of K = 64 above per 64)
        v4f acc[4][8];
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 8; ++j) for (int x = 0; x < 4; ++x) acc[i][j][x] = 0.f;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[i & 1], b[j & 3], acc[i][j], 4, 4, 0, 0, 0, 0);
        float s = 0;
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 8; ++j) for (int x = 0; x < 4; ++x) s += acc[i][j][x];
        out[tid] = s;
    }
}

int main() {
    int *in; float *out;
    std::vector<int> h(65536);
    srand(1);
    for (auto &x : h) { unsigned v = 0; for (int n = 0; n < 8; ++n) { const unsigned c[3] = {0x0, 0x2, 0xA}; v |= c[rand() % 3] << (4 * n); } x = (int)v; }
    hipMalloc(&in, 65536 * 4); hipMalloc(&out, 256 * 512 * 4);
    hipMemcpy(in, h.data(), 65536 * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int waves = 0; waves < 2; ++waves) {
        const int threads = waves ? 512 : 256;
        for (int shape : {32, 16, 326, 32, 16, 326}) {
            // iters: 32x32x64: 8 MFMAs per iteration x 131072 FLOP; 16x16x128: 32 MFMAs x 65536 FLOP
            const int iters = shape != 16 ? 40000 : 10000;
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (shape == 32) hipLaunchKernelGGL(k<32>, dim3(256), dim3(threads), 0, 0, in, out, iters);
                else if (shape == 326) hipLaunchKernelGGL(k<326>, dim3(256), dim3(threads), 0, 0, in, out, iters);
                else hipLaunchKernelGGL(k<16>, dim3(256), dim3(threads), 0, 0, in, out, iters);
                hipEventRecord(e1); hipEventSynchronize(e1);
            }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double flop = (double)256 * (threads / 64) * iters * (shape != 16 ? 8 * 131072.0 : 32 * 65536.0);
            printf("%d wave(s) per SIMD, shape %s: %.2f ms, %.0f TFLOP/s\n", threads / 256, shape == 32 ? "32x32x64 fp4 x fp4" : (shape == 326 ? "32x32x64 fp6 x fp4" : "16x16x128 fp4 x fp4"), ms, flop / ms / 1e9);
        }
    }
    return 0;
}
