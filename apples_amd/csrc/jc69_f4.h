// fp4 operand images of the JC69 pair counts on the matrix cores (dist.hip: k_jc69_mfma; select.hip: k_cluster_dist_mfma):
// the helpers both kernels share.  Encoding and orders: the comment above k_jc69_mfma.
#pragma once
#include "common.h"

typedef int v4i_t __attribute__((ext_vector_type(4)));
typedef int v8i_t __attribute__((ext_vector_type(8)));
typedef float v16f_t __attribute__((ext_vector_type(16)));

#define MF_RS 144  // LDS row stride in bytes: 4 components x 32 bytes + 16 (conflict-free 16-byte reads)

// One thread expands 16 sites of one reference row (dwords 2*(quarter&1), +1 of one 32-site word: the
// word's bits j, j+4, ... become the nibbles of dword j) into the four component chunks.  A valid site
// is the nibble 0x2 (+1.0); a set sign bit makes it 0xA (-1.0).  The sign planes are zero at gaps.
// The third component's sign is the XOR of the other two's.
__device__ __forceinline__ void expand_quarter(uint32_t m, uint32_t c0, uint32_t c1, uint8_t *row, int quarter) {
    const int sh = (quarter & 1) * 2;
    m >>= sh; c0 >>= sh; c1 >>= sh;
    const uint32_t K2 = 0x22222222u, K8 = 0x88888888u;
    const uint32_t v0 = (m << 1) & K2, v1 = m & K2;
    const uint32_t c2 = c0 ^ c1;
    uint2 *d = reinterpret_cast<uint2 *>(row + quarter * 8);  // chunk c of the row starts at c * 32
    d[0] = make_uint2(((c1 << 3) & K8) | v0, ((c1 << 2) & K8) | v1);
    d[4] = make_uint2(((c0 << 3) & K8) | v0, ((c0 << 2) & K8) | v1);
    d[8] = make_uint2(((c2 << 3) & K8) | v0, ((c2 << 2) & K8) | v1);
    d[12] = make_uint2(v0, v1);
}

__device__ __forceinline__ v16f_t mfma_f4(const v4i_t &a, const v4i_t &b, const v16f_t &c) {
    const v8i_t a8 = {a[0], a[1], a[2], a[3], 0, 0, 0, 0}, b8 = {b[0], b[1], b[2], b[3], 0, 0, 0, 0};
    // cbsz = blgp = 4: both operands fp4 (e2m1); scales 0 select the unscaled instruction
    return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, c, 4, 4, 0, 0, 0, 0);
}

