// Packing kernels: the reference's 1-byte-per-site rows (apples/fasta2dic.py:71) -> device layouts.
//
// Bit-plane layout (JC69 path): sites are cut into 32-site words, 4 words form a group g.
// Per row: a gap plane M (bit = site is not '-') and P code planes.  P = 2 is the ACGT fast
// path (code = (byte >> 1) & 3: A=0 C=1 T=2 G=3); P = 8 keeps the raw byte, so "any other byte
// is an ordinary symbol" (distance.py:733 compares bytes) holds for every input.  In the 2-plane form a
// byte beyond ACGT- is packed AS A GAP and reported (the context's flag, the row's flag): what is then missing
// from a pair's counts is added back exactly for the pairs that matter (dist.hip: k_exotic_fix) or the row takes
// the 8-plane form as well (api.hip: exotic rows and queries).
//   reference rows : packed[(g*(P+1) + plane) * slots_pad + slot]   (uint4; a wave reads 1 KiB)
//   query rows     : packed[(((q/16)*G + g)*16 + q%16) * (P+1) + plane]   (uint4; wave-uniform reads:
//                    for one word group the 16 queries of a tile are contiguous, so several
//                    queries arrive per scalar load)
#include "common.h"

template <int P>
__global__ __launch_bounds__(APPLES_TPB) void k_pack_rows(const uint8_t *__restrict__ raw, int64_t n_rows, int L, int G,
                                                          uint4 *__restrict__ out, int64_t slots_pad, int query_layout,
                                                          int *__restrict__ exotic, const int32_t *__restrict__ src_row,
                                                          int32_t *__restrict__ row_bad) {
    int64_t row = (int64_t)blockIdx.x * APPLES_TPB + threadIdx.x;
    int g = blockIdx.y;
    if (row >= n_rows) return;
    // reference layout: slot `row` holds the caller's row src_row[row] (the alignment stays in the caller's
    // order on the device; no re-ordered copy is made on the host)
    const uint8_t *src = raw + (src_row ? (int64_t)src_row[row] : row) * (int64_t)L;
    uint32_t m[4] = {0, 0, 0, 0};
    uint32_t c[P][4];
#pragma unroll
    for (int p = 0; p < P; ++p)
#pragma unroll
        for (int k = 0; k < 4; ++k) c[p][k] = 0;
    int bad = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        int base = (g * 4 + k) * 32;
        for (int i = 0; i < 32; ++i) {
            int site = base + i;
            if (site >= L) break;
            uint32_t b = src[site];
            uint32_t nd = (b != (uint32_t)'-');
            if (P == 2) {
                const uint32_t ok = (b == 'A' || b == 'C' || b == 'G' || b == 'T');
                bad |= nd & !ok;
                nd = ok;  // (a byte beyond ACGT-: a gap here, and reported)
            }
            m[k] |= nd << i;
            if (P == 2) {
                uint32_t code = (b >> 1) & 3u;
                c[0][k] |= ((code & 1u) & nd) << i;
                c[1][k] |= ((code >> 1) & nd) << i;
            } else {
#pragma unroll
                for (int p = 0; p < P; ++p) c[p][k] |= (((b >> p) & 1u) & nd) << i;
            }
        }
    }
    if (bad) {
        if (exotic) atomicOr(exotic, 1);
        if (row_bad) row_bad[row] = 1;
    }
    if (query_layout) {
        uint4 *dst = out + (((row >> 4) * G + g) * 16 + (row & 15)) * (P + 1);
        dst[0] = make_uint4(m[0], m[1], m[2], m[3]);
#pragma unroll
        for (int p = 0; p < P; ++p) dst[1 + p] = make_uint4(c[p][0], c[p][1], c[p][2], c[p][3]);
    } else {
        out[((int64_t)g * (P + 1) + 0) * slots_pad + row] = make_uint4(m[0], m[1], m[2], m[3]);
#pragma unroll
        for (int p = 0; p < P; ++p)
            out[((int64_t)g * (P + 1) + 1 + p) * slots_pad + row] = make_uint4(c[p][0], c[p][1], c[p][2], c[p][3]);
    }
}

// The same packing with the lanes along the SITES (round 6): a thread = one 32-site word of one row, consecutive threads =
// consecutive words, rows back to back -- a wavefront reads 2 KB of consecutive row bytes (k_pack_rows above reads a row per
// thread: 64 rows, 64 cache lines per load instruction; 0.86 ms per 33 000 queries of 1 000 sites, 0.8 of them overhead) and the
// four threads of a word group hand their words to the first, which stores the planes' uint4s.  Needs 4 | threads per row.
template <int P>
__global__ __launch_bounds__(APPLES_TPB) void k_pack_rows_w(const uint8_t *__restrict__ raw, int64_t n_rows, int L, int G,
                                                            uint4 *__restrict__ out, int64_t slots_pad, int query_layout,
                                                            int *__restrict__ exotic, const int32_t *__restrict__ src_row,
                                                            int32_t *__restrict__ row_bad) {
    const int64_t idx = (int64_t)blockIdx.x * APPLES_TPB + threadIdx.x;
    const int W4 = G * 4;  // words per row, padded to whole groups
    const int64_t row = idx / W4;
    const int w = (int)(idx - row * W4);
    const bool on = row < n_rows;
    uint32_t m = 0, c[P];
#pragma unroll
    for (int p = 0; p < P; ++p) c[p] = 0;
    int bad = 0;
    if (on && w * 32 < L) {
        const uint8_t *src = raw + (src_row ? (int64_t)src_row[row] : row) * (int64_t)L + w * 32;
        const int nb = min(32, L - w * 32);
        uint32_t bytes[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (nb == 32 && ((reinterpret_cast<uintptr_t>(src) & 3) == 0)) {
#pragma unroll
            for (int k = 0; k < 8; ++k) bytes[k] = reinterpret_cast<const uint32_t *>(src)[k];
        } else {
            for (int i = 0; i < nb; ++i) bytes[i >> 2] |= (uint32_t)src[i] << (8 * (i & 3));
            for (int i = nb; i < 32; ++i) bytes[i >> 2] |= (uint32_t)'-' << (8 * (i & 3));  // (beyond the row: gaps)
        }
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            const uint32_t b = (bytes[i >> 2] >> (8 * (i & 3))) & 0xffu;
            uint32_t nd = (b != (uint32_t)'-');
            if (P == 2) {
                const uint32_t ok = (b == 'A' || b == 'C' || b == 'G' || b == 'T');
                bad |= nd & !ok;
                nd = ok;  // (a byte beyond ACGT-: a gap here, and reported)
            }
            m |= nd << i;
            if (P == 2) {
                const uint32_t code = (b >> 1) & 3u;
                c[0] |= ((code & 1u) & nd) << i;
                c[1] |= ((code >> 1) & nd) << i;
            } else {
#pragma unroll
                for (int p = 0; p < P; ++p) c[p] |= (((b >> p) & 1u) & nd) << i;
            }
        }
    }
    if (bad) {
        if (exotic) atomicOr(exotic, 1);
        if (row_bad) row_bad[row] = 1;
    }
    // words k = 0..3 of a group sit in four consecutive lanes (W4 is a multiple of 4, so a group never straddles rows or wavefronts)
    uint4 pm, pc[P];
    pm = make_uint4(m, __shfl_down(m, 1, 64), __shfl_down(m, 2, 64), __shfl_down(m, 3, 64));
#pragma unroll
    for (int p = 0; p < P; ++p) pc[p] = make_uint4(c[p], __shfl_down(c[p], 1, 64), __shfl_down(c[p], 2, 64), __shfl_down(c[p], 3, 64));
    if (!on || (w & 3) != 0) return;
    const int g = w >> 2;
    if (query_layout) {
        uint4 *dst = out + (((row >> 4) * G + g) * 16 + (row & 15)) * (P + 1);
        dst[0] = pm;
#pragma unroll
        for (int p = 0; p < P; ++p) dst[1 + p] = pc[p];
    } else {
        out[((int64_t)g * (P + 1) + 0) * slots_pad + row] = pm;
#pragma unroll
        for (int p = 0; p < P; ++p) out[((int64_t)g * (P + 1) + 1 + p) * slots_pad + row] = pc[p];
    }
}

int launch_pack_rows(apples_ctx *ctx, const uint8_t *d_raw, int64_t n_rows, int L, int planes, uint4 *d_out,
                     int64_t slots_pad, bool query_layout, int *d_exotic, hipStream_t st, const int32_t *d_src_row,
                     int32_t *d_row_bad) {
    if (n_rows == 0) return 0;
    if (!st) st = ctx->stream;
    int G = ctx->aln.G;
    if (!knob_on(ctx, "APPLES_PACK_BY_ROW")) {  // (the knob: the thread-per-row form of rounds 1 - 5, for comparison)
        const int64_t total = n_rows * (int64_t)G * 4;
        const dim3 gw((unsigned)((total + APPLES_TPB - 1) / APPLES_TPB));
        if (planes == 2)
            hipLaunchKernelGGL(k_pack_rows_w<2>, gw, dim3(APPLES_TPB), 0, st, d_raw, n_rows, L, G, d_out, slots_pad,
                               query_layout ? 1 : 0, d_exotic, d_src_row, d_row_bad);
        else
            hipLaunchKernelGGL(k_pack_rows_w<8>, gw, dim3(APPLES_TPB), 0, st, d_raw, n_rows, L, G, d_out, slots_pad,
                               query_layout ? 1 : 0, d_exotic, d_src_row, (int32_t *)nullptr);
        HIP_TRY(ctx, hipGetLastError());
        return 0;
    }
    dim3 grid((unsigned)((n_rows + APPLES_TPB - 1) / APPLES_TPB), (unsigned)G);
    if (planes == 2)
        hipLaunchKernelGGL(k_pack_rows<2>, grid, dim3(APPLES_TPB), 0, st, d_raw, n_rows, L, G, d_out, slots_pad,
                           query_layout ? 1 : 0, d_exotic, d_src_row, d_row_bad);
    else
        hipLaunchKernelGGL(k_pack_rows<8>, grid, dim3(APPLES_TPB), 0, st, d_raw, n_rows, L, G, d_out, slots_pad,
                           query_layout ? 1 : 0, d_exotic, d_src_row, (int32_t *)nullptr);
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}

// scoredist layout: residue index per site, a2i of apples/distance.py:418-678 (ARNDCQEGHILKMFPSTWYV
// both cases -> 0..19, every other byte -> 0) with '-' -> 20 (a zero row/column of the table, so
// gapped sites add +0.0 exactly as nondash*BLOSUM45[...] does at distance.py:706).
//   reference rows : aa[(s16 * slots_pad + slot) * 16 + k] = index * 8 (the byte offset of the
//                    table column, so the kernel adds it to a row base without scaling)
//   query rows     : aa[q * Lpad + site] = index
//   gap masks      : mask[s16 * slots_pad + slot] / mask[q * (Lpad/16) + s16]: bit k = site is not '-'
// (aa_index: common.h)

__global__ __launch_bounds__(APPLES_TPB) void k_pack_aa(const uint8_t *__restrict__ raw, int64_t n_rows, int L, int Lpad,
                                                        uint8_t *__restrict__ out, uint16_t *__restrict__ mask,
                                                        int64_t slots_pad, int query_layout, const int32_t *__restrict__ src_row) {
    int64_t row = (int64_t)blockIdx.x * APPLES_TPB + threadIdx.x;
    int s16 = blockIdx.y;
    if (row >= n_rows) return;
    const uint8_t *src = raw + (src_row ? (int64_t)src_row[row] : row) * (int64_t)L;
    uint32_t w[4] = {0, 0, 0, 0};
    uint32_t m = 0;
    const uint32_t scale = query_layout ? 1u : 8u;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        int site = s16 * 16 + k;
        uint32_t v = site < L ? aa_index(src[site]) : 20u;
        m |= (uint32_t)(v != 20u) << k;
        w[k >> 2] |= (v * scale) << (8 * (k & 3));
    }
    uint4 val = make_uint4(w[0], w[1], w[2], w[3]);
    if (query_layout) {
        *reinterpret_cast<uint4 *>(out + row * (int64_t)Lpad + s16 * 16) = val;
        mask[row * (int64_t)(Lpad / 16) + s16] = (uint16_t)m;
    } else {
        *reinterpret_cast<uint4 *>(out + ((int64_t)s16 * slots_pad + row) * 16) = val;
        mask[(int64_t)s16 * slots_pad + row] = (uint16_t)m;
    }
}

// The query layout (row-major output): consecutive threads take consecutive 16-site groups of ONE row, so a wavefront reads a
// kilobyte of one row and writes a kilobyte of its image (k_pack_aa's thread-per-row mapping suits the slot-major reference
// layout: there the writes coalesce; on query rows its byte loads touch 64 rows per instruction -- 0.12 ms for the 12 500 rows
// in front of config 4's first distance launch)
__global__ __launch_bounds__(APPLES_TPB) void k_pack_aa_rows(const uint8_t *__restrict__ raw, int64_t n_rows, int L, int Lpad,
                                                             uint8_t *__restrict__ out, uint16_t *__restrict__ mask,
                                                             const int32_t *__restrict__ src_row) {
    const int n16 = Lpad / 16;
    const int64_t idx = (int64_t)blockIdx.x * APPLES_TPB + threadIdx.x;
    const int64_t row = idx / n16;
    const int s16 = (int)(idx - row * n16);
    if (row >= n_rows) return;
    const uint8_t *src = raw + (src_row ? (int64_t)src_row[row] : row) * (int64_t)L;
    uint32_t w[4] = {0, 0, 0, 0};
    uint32_t m = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int site = s16 * 16 + k;
        const uint32_t v = site < L ? aa_index(src[site]) : 20u;
        m |= (uint32_t)(v != 20u) << k;
        w[k >> 2] |= v << (8 * (k & 3));
    }
    *reinterpret_cast<uint4 *>(out + row * (int64_t)Lpad + s16 * 16) = make_uint4(w[0], w[1], w[2], w[3]);
    mask[row * (int64_t)n16 + s16] = (uint16_t)m;
}

int launch_pack_aa(apples_ctx *ctx, const uint8_t *d_raw, int64_t n_rows, int L, uint8_t *d_out, uint16_t *d_mask,
                   int64_t slots_pad, bool query_layout, hipStream_t st, const int32_t *d_src_row) {
    if (n_rows == 0) return 0;
    if (!st) st = ctx->stream;
    int Lpad = (L + 15) / 16 * 16;
    if (query_layout) {
        const int64_t n = n_rows * (Lpad / 16);
        hipLaunchKernelGGL(k_pack_aa_rows, dim3((unsigned)((n + APPLES_TPB - 1) / APPLES_TPB)), dim3(APPLES_TPB), 0, st, d_raw, n_rows, L, Lpad,
                           d_out, d_mask, d_src_row);
        HIP_TRY(ctx, hipGetLastError());
        return 0;
    }
    dim3 grid((unsigned)((n_rows + APPLES_TPB - 1) / APPLES_TPB), (unsigned)(Lpad / 16));
    hipLaunchKernelGGL(k_pack_aa, grid, dim3(APPLES_TPB), 0, st, d_raw, n_rows, L, Lpad, d_out, d_mask,
                       slots_pad, query_layout ? 1 : 0, d_src_row);
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}
