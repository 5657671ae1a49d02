// Does hipExtStreamCreateWithCUMask restrict a stream's kernels to the masked CUs on this box?
// A compute-bound kernel of 4096 workgroups is timed on streams with 256, 128 and 64 CUs enabled.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void spin(float *out, int iters) {
    float a = threadIdx.x * 1e-3f, b = 1.0001f;
    for (int i = 0; i < iters; ++i) a = a * b + 0.5f;
    if (a == 12345.f) out[0] = a;
}
int main() {
    float *d;
    hipMalloc(&d, 4);
    for (int keep = 4; keep >= 1; --keep) {
        uint32_t mask[8];
        for (int w = 0; w < 8; ++w) {
            mask[w] = 0;
            for (int b = 0; b < 32; ++b)
                if ((b & 3) < keep) mask[w] |= 1u << b;
        }
        hipStream_t st;
        hipError_t e = hipExtStreamCreateWithCUMask(&st, 8, mask);
        hipEvent_t t0, t1;
        hipEventCreate(&t0); hipEventCreate(&t1);
        spin<<<4096, 256, 0, st>>>(d, 20000);
        hipStreamSynchronize(st);
        hipEventRecord(t0, st);
        spin<<<4096, 256, 0, st>>>(d, 200000);
        hipEventRecord(t1, st);
        hipStreamSynchronize(st);
        float ms = 0;
        hipEventElapsedTime(&ms, t0, t1);
        uint32_t got[8] = {0};
        hipError_t g = hipExtStreamGetCUMask(st, 8, got);
        printf("keep %d of 4: create=%d  %.2f ms  getmask=%d %08x %08x\n", keep, (int)e, ms, (int)g, got[0], got[7]);
        hipStreamDestroy(st);
    }
    return 0;
}
