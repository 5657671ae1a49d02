// scoredist (apples/distance.py:681-715) with singleton clusters, fused route: "most pairs fail the threshold".
//
// d <= f  <=>  tot / valid <= c = 1 - exp(-f / 1.3), tot = sum over sites of T[q_s][r_s] (the BLOSUM45 dissimilarities,
// all >= 0, zero against a gap), and only 0.3 % of the pairs of the benchmark shape pass.  Two kernels:
//
//  1. k_sd_gemm: an exact LOWER BOUND of tot for every pair on the matrix cores.  tot is the bilinear form
//     sum_s sum_a Tq[q_s][a] * [r_s == a] over the 20 residues: the reference row as a one-hot image (20 fp4 values per
//     site: 1.0 at its residue, all zero at a gap), the query row as 20 table values per site, each ROUNDED DOWN to the
//     fp4 grid v / 4 with v in {0, .5, 1, 1.5, 2, 3, 4, 6} (e2m1).  Products and sums of these are exact in the f32
//     accumulators (multiples of 0.5 below 2^24), so acc / 4 <= tot holds as a statement about real numbers, and a
//     pair whose bound exceeds c * min(valid sites of the query, of the reference) >= c * valid cannot pass.  What is
//     left (0.8 - 1.1 % of the pairs at the benchmark shape: the quantised table gives 0.83 of tot on average) is
//     written per 64-slot segment in slot order -- the format k_select_fast reads -- as CANDIDATES.
//     The kernel is dist_gemm.hip's pipeline (pre-expanded operand images in HBM stored as the kernel's LDS tile
//     images, LDS-DMA two steps ahead, three generations, 256 x 256 workgroup tiles, persistent workgroups walking
//     strips per XCD) with a plain K loop: a step is 128 K values = 64 bytes per row = two MFMAs per tile pair.
//  2. k_sd_exact: a workgroup per query walks its candidates, evaluates each pair exactly as k_scoredist does (fp64,
//     sites left to right, the same table in LDS: the same bits), applies the real test 0 <= d <= f and closes the
//     segments up in place.  k_select_fast then sees what k_scoredist<MODE 1> would have written.
//
// Queries left with fewer than `-b` survivors take the top-up rule on rows of the same lower bounds (k_sd_gemm<ROWS>) with exact
// distances only where the `-b` nearest can be (k_sd_topup below); full rows (k_scoredist listed mode) behind APPLES_DBG_NO_SD_TOPUP.
#include <cmath>
#include <cstdlib>

#include "common.h"
#include "libm_log.h"

typedef int v4i_t __attribute__((ext_vector_type(4)));
typedef int v8i_t __attribute__((ext_vector_type(8)));
typedef float v16f_t __attribute__((ext_vector_type(16)));

#define SD_T 256             // tile edge: queries and reference slots per workgroup
#ifndef SD_STRIP
#define SD_STRIP 4           // reference tiles per strip
#endif
#define SD_IMG (SD_T * 64)   // bytes of one tile-step image: 256 rows x 128 fp4 values
#define SD_IMG6 (SD_T * 96)  // ... x 128 fp6 values: the fp4-shaped part (16 bytes of each lane's 24) + 8 KB of [chunk][row] 8-byte tails

namespace {

__device__ __forceinline__ v16f_t sd_mfma(const v4i_t &a, const v4i_t &b, const v16f_t &c) {
    const v8i_t a8 = {a[0], a[1], a[2], a[3], 0, 0, 0, 0}, b8 = {b[0], b[1], b[2], b[3], 0, 0, 0, 0};
    return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, c, 4, 4, 0, 0, 0, 0);  // fp4 x fp4, unscaled
}
// the query side as fp6 (e2m3: 32 values = 192 bits per lane, the matrix cores' fp4 rate), the one-hot side fp4
__device__ __forceinline__ v16f_t sd_mfma6(const v4i_t &a, const int2 &a2, const v4i_t &b, const v16f_t &c) {
    const v8i_t a8 = {a[0], a[1], a[2], a[3], a2.x, a2.y, 0, 0}, b8 = {b[0], b[1], b[2], b[3], 0, 0, 0, 0};
    return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, c, 2, 4, 0, 0, 0, 0);
}

// rows -> operand image.  K index of (site s, residue a) = 20 s + a; a step is 128 K values, chunk c of a row its
// values 32 c .. 32 c + 31 (16 bytes: value e in nibble e & 7 of dword e >> 3 -- any order serves as long as both
// operands share it).  The image is stored as the kernel's LDS image of a 256-row tile and one step: row rr's chunk c
// at position 4 rr + (c ^ ((rr >> 2) & 3)) of the tile-step's 1024 chunks (dist_gemm.hip).  tq4 == nullptr: one-hot
// (reference side); else the fp4 codes of the rounded-down table row of the query's residue.  nv[image row] = the
// row's sites that are not gaps, -1 for padding rows (no pair with them may pass).
__global__ __launch_bounds__(APPLES_TPB) void k_sd_expand(const uint8_t *__restrict__ raw, int64_t n, int L, int NB,
                                                          uint4 *__restrict__ out, int64_t n_img,
                                                          const int32_t *__restrict__ src_row, int64_t row0,
                                                          const uint8_t *__restrict__ tq4, float *__restrict__ nv,
                                                          const int32_t *__restrict__ n_dev, int fp6) {
    // two values per byte and 20 per site: a byte never straddles sites, so the image is a byte table look-up -- rowb[v][h] =
    // values 2 h, 2 h + 1 of the row of residue v (row 20 = a gap: zeros)
    __shared__ uint8_t rowb[21 * 10];
    for (int i = threadIdx.x; i < 210; i += APPLES_TPB) {
        const int v = i / 10, a = (i - v * 10) * 2;
        uint32_t lo = 0, hi = 0;
        if (v < 20) {
            lo = tq4 ? tq4[v * 20 + a] : (v == a ? 2u : 0u);
            hi = tq4 ? tq4[v * 20 + a + 1] : (v == a + 1 ? 2u : 0u);
        }
        rowb[i] = (uint8_t)(lo | (hi << 4));
    }
    __syncthreads();
    const int64_t idx = (int64_t)blockIdx.x * APPLES_TPB + threadIdx.x;  // one thread per (row, step, chunk)
    if (n_dev) n = n_img = *n_dev;  // listed rows: the list's length lives on the device (nothing beyond it is written)
    if (idx >= n_img * NB * 4) return;
    const int c = (int)(idx & 3);
    const int64_t rb = idx >> 2;
    const int b = (int)(rb % NB);
    const int64_t q = rb / NB;
    if (fp6) {
        // 32 six-bit codes per chunk (value e at bits 6 e .. 6 e + 5 of the lane's 192): the first 128 bits where the fp4 chunk would
        // lie, the last 64 in the tile-step's tail area [chunk][row]
        uint32_t w6[6] = {0, 0, 0, 0, 0, 0};
        if (q < n) {
            const uint8_t *row = raw + (src_row ? (int64_t)src_row[q] : q) * (int64_t)L;
            const int k0 = b * 128 + c * 32;
            int s = k0 / 20, a = k0 - s * 20;
            uint32_t v = s < L ? aa_index(row[s]) : 20u;
#pragma unroll
            for (int e = 0; e < 32; ++e) {
                const uint32_t code = v < 20u ? tq4[v * 20u + a] : 0u;
                const int bit = 6 * e;
                w6[bit >> 5] |= code << (bit & 31);
                if ((bit & 31) > 26) w6[(bit >> 5) + 1] |= code >> (32 - (bit & 31));
                if (++a == 20) { a = 0; ++s; v = s < L ? aa_index(row[s]) : 20u; }
            }
        }
        const int64_t ar = row0 + q, tile = ar >> 8;
        const int rr = (int)(ar & 255), sw = (rr >> 2) & 3;
        uint8_t *ts = reinterpret_cast<uint8_t *>(out) + (tile * NB + b) * (int64_t)SD_IMG6;
        reinterpret_cast<uint4 *>(ts)[rr * 4 + (c ^ sw)] = make_uint4(w6[0], w6[1], w6[2], w6[3]);
        reinterpret_cast<uint2 *>(ts + SD_IMG)[c * 256 + rr] = make_uint2(w6[4], w6[5]);
        return;
    }
    uint32_t w[4] = {0, 0, 0, 0};
    if (q < n) {
        const uint8_t *row = raw + (src_row ? (int64_t)src_row[q] : q) * (int64_t)L;
        const int k0 = b * 128 + c * 32;
        int s = k0 / 20, h = (k0 - s * 20) >> 1;
        // the chunk's 16 bytes lie in at most three sites
        uint32_t v3[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) v3[i] = s + i < L ? aa_index(row[s + i]) : 20u;
        int si = 0;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const uint32_t v = si == 0 ? v3[0] : (si == 1 ? v3[1] : v3[2]);
            w[j >> 2] |= (uint32_t)rowb[v * 10 + h] << (8 * (j & 3));
            if (++h == 10) { h = 0; ++si; }
        }
    }
    const int64_t ar = row0 + q, tile = ar >> 8;
    const int rr = (int)(ar & 255), sw = (rr >> 2) & 3;
    out[(tile * NB + b) * 1024 + rr * 4 + (c ^ sw)] = make_uint4(w[0], w[1], w[2], w[3]);
}

// sites of a row that are not gaps, from the 16-bit gap masks the packing kernel left (query layout [row][n16], reference
// layout [n16][slots_pad]); -1 for the image's padding rows (no pair with them may pass the filter)
__global__ __launch_bounds__(APPLES_TPB) void k_sd_nv(const uint16_t *__restrict__ mask, int64_t n, int64_t n_img, int n16,
                                                      int64_t ref_stride, float *__restrict__ nv) {
    const int64_t r = (int64_t)blockIdx.x * APPLES_TPB + threadIdx.x;
    if (r >= n_img) return;
    int cnt = -1;
    if (r < n) {
        cnt = 0;
        for (int s = 0; s < n16; ++s) cnt += __popc((uint32_t)(ref_stride ? mask[(int64_t)s * ref_stride + r] : mask[r * n16 + s]));
    }
    nv[r] = (float)cnt;
}

// R = (NB - 2) % 3: the shape of the main loop's tail, fixed per launch (as k_jc69_gemm)
// ROWS: the bounds themselves, for a list of queries (the top-up path): the query image holds the listed queries in list
// order, *qcount of them; acc of (list entry r, slot) goes to row qlist[r] of `rows` (floats, row stride in doubles)
// F6 (APPLES_DBG_SD_FP6, off by default): the query image holds fp6 values (SD_IMG6 bytes per tile-step), 24 bytes per lane and
// MFMA: the table rounded down to the e2m3 grid keeps 0.955 of tot where the fp4 grid keeps 0.83, which halves the candidates
// (k_sd_exact 4.2 -> 1.9 ms per C4 pass) -- and costs the filter more than that: 10.5 -> 13.6 ms (a quarter more bytes through
// an LDS-DMA path that already takes a third of the kernel, the mixed fp6 x fp4 instruction 7 % slower in a bare loop,
// 256 registers).  Same bytes out; kept as the measured alternative.
template <int R, bool ROWS, bool F6>
__global__ __launch_bounds__(512, 2) void k_sd_gemm(const uint8_t *__restrict__ rf4, const uint8_t *__restrict__ qf4,
                                                    int64_t qrow0, int64_t slots_pad, int NB, int64_t nq, int TQ, int TR,
                                                    const float *__restrict__ nvr, const float *__restrict__ nvq, float k4c,
                                                    int32_t *__restrict__ seg_slot, int32_t *__restrict__ seg_cnt,
                                                    const int32_t *__restrict__ qlist, const int32_t *__restrict__ qcount,
                                                    double *__restrict__ rows, int64_t row_stride) {
    if (ROWS) {
        nq = *qcount;
        TQ = (int)((nq + SD_T - 1) / SD_T);
        if (TQ == 0) return;
    }
    constexpr int QT = SD_T, AI = QT * 64, A2 = F6 ? QT * 32 : 0, GEN = AI + A2 + SD_IMG, QIMG = F6 ? SD_IMG6 : SD_IMG;
    __shared__ __attribute__((aligned(1024))) uint8_t lds[3 * GEN];
#define Aq(g) (lds + (g) * GEN)
#define A2q(g) (lds + (g) * GEN + AI)
#define Br(g) (lds + (g) * GEN + AI + A2)
    // persistent workgroups, one per CU: XCD x takes the strips x, x + 8, ... of SD_STRIP reference tiles; within the XCD the
    // tiles (query tile major, the strip's reference tiles inside) are dealt round-robin to its workgroups
    const int xcd = blockIdx.x & 7;
    const int64_t per_strip = (int64_t)TQ * SD_STRIP, stride = gridDim.x >> 3;
    const int64_t l_end = (((TR + SD_STRIP - 1) / SD_STRIP + 7) / 8) * per_strip;
    auto tile_at = [&](int64_t l, int64_t &qt_, int64_t &rt_) __attribute__((always_inline)) {
        const int64_t strip = (l / per_strip) * 8 + xcd, within = l % per_strip;
        rt_ = strip * SD_STRIP + within % SD_STRIP;
        qt_ = within / SD_STRIP;
    };
    int64_t l = blockIdx.x >> 3, qt = 0, rt = 0;
    for (;; l += stride) {  // first tile
        if (l >= l_end) return;
        tile_at(l, qt, rt);
        if (rt < TR) break;
    }
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wq = wv >> 1, wr = wv & 1;
    // DMA roles (dist_gemm.hip): a tile-step image is 16 pieces of 1 KB; this wavefront moves pieces 2 wv, 2 wv + 1 of the
    // query image and of the reference image.  The query tile starts at image row qrow0 + q0, a multiple of 32: its rows
    // may lie in two image tiles.
    uint32_t doff[3];
    const uint8_t *qtile, *rtile;
    auto set_tile = [&](int64_t qt_, int64_t rt_) __attribute__((always_inline)) {
        const int64_t qabs = qrow0 + qt_ * QT;
        const int qin = (int)(qabs & 255);
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int row = (wv * 2 + k) * 16 + (lane >> 2), slot = lane & 3, ar = qin + row;
            doff[k] = (uint32_t)((ar >> 8) * NB * QIMG + (ar & 255) * 64 + slot * 16);
        }
        if (F6) {  // this wavefront's piece of the 8-byte tails: chunk wv >> 1, rows (wv & 1) * 128 + 2 lane, + 1 (one image tile: qin is even)
            const int ar = qin + (wv & 1) * 128 + 2 * lane;
            doff[2] = (uint32_t)((ar >> 8) * NB * QIMG + SD_IMG + ((wv >> 1) * 256 + (ar & 255)) * 8);
        }
        qtile = qf4 + (qabs >> 8) * (int64_t)NB * QIMG;
        rtile = rf4 + rt_ * (int64_t)NB * SD_IMG;
    };
    set_tile(qt, rt);
    const uint32_t roff = (uint32_t)(wv * 2 * 1024 + lane * 16);
    const int fr = lane & 31, fh = lane >> 5;
    int coff[2];  // byte offset of this lane's chunk of K half h of the step (its row, its 32 of the MFMA's 64 values)
#pragma unroll
    for (int h = 0; h < 2; ++h) coff[h] = ((h * 2 + fh) ^ ((fr >> 2) & 3)) * 16;
    const int arow = (wq * 64 + fr) * 64, brow = (wr * 128 + fr) * 64;
    v16f_t acc[2][4];
    auto dma = [&](int b, int g, int part) __attribute__((always_inline)) {  // part 0: query pieces, 1: reference pieces
#pragma unroll
        for (int k = 0; k < (F6 ? 3 : 2); ++k) {
            if (k == 2 && part != 0) continue;
            const uint8_t *src = part == 0 ? (qtile + b * QIMG) + doff[k] : (rtile + b * SD_IMG + k * 1024) + roff;
            uint8_t *dst = part == 0 ? (k == 2 ? A2q(g) + wv * 1024 : Aq(g) + (wv * 2 + k) * 1024) : Br(g) + (wv * 2 + k) * 1024;
#ifndef SD_NO_DMA  // (timing experiments: scripts/r04_sd_parts_exp.sh)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
#endif
        }
    };
    v4i_t fa[2][2], fb[2][4];  // fragment sets: [0] the step's first K half, [1] its second
    int2 fa2[2][2];            // fp6: the last 64 bits of the query side's 192
    auto load_frags = [&](int g, int h) __attribute__((always_inline)) {
#ifdef SD_NO_FRAGS
        if (g >= 0) return;
#endif
        const uint8_t *A = Aq(g) + arow + coff[h], *B = Br(g) + brow + coff[h];
#pragma unroll
        for (int i = 0; i < 2; ++i) fa[h][i] = *reinterpret_cast<const v4i_t *>(A + i * 32 * 64);
        if (F6) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
                fa2[h][i] = *reinterpret_cast<const int2 *>(A2q(g) + ((h * 2 + fh) * 256 + wq * 64 + i * 32 + fr) * 8);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) fb[h][j] = *reinterpret_cast<const v4i_t *>(B + j * 32 * 64);
    };
    auto mfmas = [&](int h) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = F6 ? sd_mfma6(fa[h][i], fa2[h][i], fb[h][j], acc[i][j]) : sd_mfma(fa[h][i], fb[h][j], acc[i][j]);
    };
    // One step = 128 K values = 2 sections of 8 MFMAs per wavefront, generation g = step % 3.  Entry: set 0 holds the
    // first half's fragments (read after the previous barrier).  Section 0: the second half's fragments are requested, the
    // query pieces of step + 2 go out into the generation every wavefront left at the previous barrier.  Then the
    // barrier: own reads of generation g done, step + 1 landed (everything but the two pieces just issued).  Section 1:
    // the reference pieces of step + 2, the first fragments of step + 1.
    auto step = [&](int b, int g, bool feed, bool more) __attribute__((always_inline)) {
        const int gn = g == 2 ? 0 : g + 1, gf = g == 0 ? 2 : g - 1;
        load_frags(g, 1);
        if (feed) dma(b + 2, gf, 0);
        mfmas(0);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (k < (F6 ? 8 : 6)) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            if (k == 1 || k == 4 || (F6 && k == 6)) {
                __builtin_amdgcn_sched_group_barrier(0x006, 4, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (feed && F6) asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)" ::: "memory");
        else if (feed) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (more) load_frags(gn, 0);
        if (feed) dma(b + 2, gf, 1);
        mfmas(1);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (k < (F6 ? 8 : 6)) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            if (k == 1 || k == 4) {
                __builtin_amdgcn_sched_group_barrier(0x006, 4, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    const int64_t n_seg = slots_pad >> 6;
    const uint32_t below = (1u << fr) - 1u;
    dma(0, 0, 0); dma(0, 0, 1);
    dma(1, 1, 0); dma(1, 1, 1);
    for (;;) {  // tiles of this workgroup; entry: the first two steps of the tile are on their way
        const int64_t r0 = rt * SD_T, q0 = qt * QT;  // (q0: relative to this launch's first query)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int x = 0; x < 16; ++x) acc[i][j][x] = 0.f;
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        load_frags(0, 0);
        __builtin_amdgcn_sched_barrier(0);
        // NB (>= 2) steps: NB - 2 feeding ones, then one that only prefetches fragments, then the last
        const int F = NB - 2;
        int b = 0;
        for (; b + 3 <= F; b += 3) {
            step(b, 0, true, true);
            step(b + 1, 1, true, true);
            step(b + 2, 2, true, true);
        }
        if (R == 0) {
            step(b, 0, false, true);
            step(b + 1, 1, false, false);
        } else if (R == 1) {
            step(b, 0, true, true);
            step(b + 1, 1, false, true);
            step(b + 2, 2, false, false);
        } else {
            step(b, 0, true, true);
            step(b + 1, 1, true, true);
            step(b + 2, 2, false, true);
            step(b + 3, 0, false, false);
        }
        int64_t nl = l + stride, nqt = 0, nrt = 0;
        bool have = false;
        for (; nl < l_end; nl += stride) {
            tile_at(nl, nqt, nrt);
            if (nrt < TR) { have = true; break; }
        }
#ifdef SD_SKIP_EPILOGUE
        {   // timing experiment: main loop only (every accumulator stays live)
            float s = 0;
            for (int i = 0; i < 2; ++i) for (int j = 0; j < 4; ++j) for (int x = 0; x < 16; ++x) s += acc[i][j][x];
            if (s == 123456.f) seg_cnt[0] = 1;
            if (have) { set_tile(nqt, nrt); dma(0, 0, 0); dma(0, 0, 1); dma(1, 1, 0); dma(1, 1, 1); }
            if (!have) break;
            l = nl; qt = nqt; rt = nrt;
            continue;
        }
#endif
        // C layout of the 32x32 tiles: column (reference slot) = lane & 31, row (query) = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5).
        if (ROWS) {
            // the next tile's first two steps: after the last barrier nobody reads LDS any more
            if (have) {
                set_tile(nqt, nrt);
                dma(0, 0, 0); dma(0, 0, 1);
                dma(1, 1, 0); dma(1, 1, 1);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int64_t qbase = q0 + wq * 64 + i * 32 + 4 * fh;
                const int rem = (int)(nq - qbase < 32 ? nq - qbase : 32);
#pragma unroll
                for (int x = 0; x < 16; ++x) {
                    const int cx = (x & 3) + 8 * (x >> 2);
                    if (cx >= rem) continue;
                    float *row = reinterpret_cast<float *>(rows + (int64_t)qlist[qbase + cx] * row_stride) + r0 + wr * 128 + fr;
#pragma unroll
                    for (int j = 0; j < 4; ++j) row[j * 32] = acc[i][j][x];
                }
            }
        } else {
        // the thresholds of this tile's rows and columns, then the next tile's first two steps: after the last barrier
        // nobody reads LDS any more.  Loads complete in order, so the counted wait below leaves the 8 pieces in flight.
        float cq[2][16], cr[4];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const float *p = nvq + qrow0 + q0 + wq * 64 + i * 32 + 4 * fh;
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const float4 v = *reinterpret_cast<const float4 *>(p + 8 * g4);
                cq[i][4 * g4] = v.x * k4c; cq[i][4 * g4 + 1] = v.y * k4c; cq[i][4 * g4 + 2] = v.z * k4c; cq[i][4 * g4 + 3] = v.w * k4c;
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) cr[j] = nvr[r0 + wr * 128 + j * 32 + fr] * k4c;
        if (have) {
            __builtin_amdgcn_sched_barrier(0);
            set_tile(nqt, nrt);
            dma(0, 0, 0); dma(0, 0, 1);
            dma(1, 1, 0); dma(1, 1, 1);
            if (F6) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_sched_barrier(0);
        // A pair is a candidate when acc <= 4 c (1 + 1e-6) min(nv of its query, nv of its reference row)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int64_t qbase = q0 + wq * 64 + i * 32 + 4 * fh;  // this lane half's first query of the 32
            // (rows past this launch's queries may be real queries of the next sub-batch: not ours to write)
            const int rem = (int)(nq - qbase < 32 ? nq - qbase : 32);
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int64_t seg = (r0 + wr * 128 + s * 64) >> 6;
                int32_t *row0 = seg_slot + qbase * slots_pad + seg * 64;
                int32_t *cnt0 = seg_cnt + qbase * n_seg + seg;
#pragma unroll
                for (int x = 0; x < 16; ++x) {
                    const int cx = (x & 3) + 8 * (x >> 2);  // this register's query, relative to qbase
                    const bool c0 = acc[i][2 * s][x] <= fminf(cq[i][x], cr[2 * s]);
                    const bool c1 = acc[i][2 * s + 1][x] <= fminf(cq[i][x], cr[2 * s + 1]);
                    if (__ballot(c0 || c1) == 0ull) continue;
                    const bool in = cx < rem;
                    const bool k0 = c0 && in, k1 = c1 && in;
                    const unsigned long long b0 = __ballot(k0), b1 = __ballot(k1);
                    if ((b0 | b1) == 0) continue;  // the counts stay at their preset zero
                    // this lane half's query: slots 0..31 of the segment from tile 2 s, 32..63 from tile 2 s + 1
                    const uint32_t lo = (uint32_t)(b0 >> (32 * fh)), hi = (uint32_t)(b1 >> (32 * fh));
                    int32_t *row = row0 + (int64_t)cx * slots_pad;
                    if (k0) row[__popc(lo & below)] = (int32_t)(seg * 64 + fr);
                    if (k1) row[__popc(lo) + __popc(hi & below)] = (int32_t)(seg * 64 + 32 + fr);
                    if (fr == 0 && in) cnt0[(int64_t)cx * n_seg] = __popc(lo) + __popc(hi);
                }
            }
        }
        }
        if (!have) break;
        l = nl; qt = nqt; rt = nrt;
    }
#undef Aq
#undef A2q
#undef Br
}

// Exact scoredist of (the workgroup's query, this lane's reference slot) for the 64 lanes of a wavefront together: the
// arithmetic of k_scoredist (dist.hip) -- sites left to right in fp64, the 21 x 21 table (zero gap row / column) in LDS at Tb
// -- so the same bits.  The rows come from the slot-major copy (DevAlign::aa_rows: Lrow bytes per slot, Lrow a multiple of
// 64, the bytes past the alignment are gaps): a pair's 16-byte pieces would otherwise be Lpad / 16 separate cache-line
// requests (that form was bound by requests, not by lookups: 5.9 ms per C4 pass).  Here four lanes fetch one slot's 64-byte
// piece (one request per sector), the pieces are staged in the wavefront's own LDS area (80-byte stride: conflict-free
// 16-byte reads) and every lane reads its own slot's bytes back; the next piece is on its way meanwhile.  All 64 lanes call
// (a lane without a pair passes any valid slot and drops the result).  The query's row and masks are staged in LDS by the
// caller (lq: per site the byte offset of its table row, residue index x 168, as 16-bit values; lqm: the masks): two broadcast
// reads per 16 sites, where a scalar memory load per 16 sites had every wavefront wait for the scalar cache between its short
// bursts of look-ups, and scalars by readfirstlane two scalar operations per site.
#define SDE_PIECE 64
#define SDE_STRIDE 80
#define SDE_WBUF (64 * SDE_STRIDE)  // bytes of LDS per wavefront
__device__ __forceinline__ double sd_eval64(const uint8_t *__restrict__ rrows, const uint16_t *__restrict__ mrows, int Lrow,
                                            int slot, uint8_t *wbuf, int *wslot, const uint8_t *lq, const uint16_t *lqm,
                                            int n16, const char *Tb, int L, double overlap, double *ratio = nullptr, int dbg = 0) {
    const int lane = threadIdx.x & 63, sub = lane & 3, grp = lane >> 2;
    __builtin_amdgcn_wave_barrier();
    wslot[lane] = slot;
    __builtin_amdgcn_wave_barrier();
    // (scalars, not arrays, and an unconditional prefetch: an array carried through the loop went to scratch memory, with
    // every load waited for at once)
    const uint8_t *src0 = rrows + (int64_t)wslot[grp] * Lrow + sub * 16, *src1 = rrows + (int64_t)wslot[16 + grp] * Lrow + sub * 16;
    const uint8_t *src2 = rrows + (int64_t)wslot[32 + grp] * Lrow + sub * 16, *src3 = rrows + (int64_t)wslot[48 + grp] * Lrow + sub * 16;
    const uint16_t *mrow = mrows + (int64_t)slot * (Lrow / 16);
    const int npiece = Lrow / SDE_PIECE;
    uint4 n0 = *reinterpret_cast<const uint4 *>(src0), n1 = *reinterpret_cast<const uint4 *>(src1);
    uint4 n2 = *reinterpret_cast<const uint4 *>(src2), n3 = *reinterpret_cast<const uint4 *>(src3);
    uint2 nm = *reinterpret_cast<const uint2 *>(mrow);  // the piece's four 16-bit gap masks
    uint8_t *wdst = wbuf + grp * SDE_STRIDE + sub * 16;
    double tot = 0.0;
    uint32_t valid = 0;
    for (int c = 0; c < npiece; ++c) {
        __builtin_amdgcn_wave_barrier();  // (the previous piece has been read by every lane: LDS operations of a wavefront complete in order)
        *reinterpret_cast<uint4 *>(wdst) = n0;
        *reinterpret_cast<uint4 *>(wdst + 16 * SDE_STRIDE) = n1;
        *reinterpret_cast<uint4 *>(wdst + 32 * SDE_STRIDE) = n2;
        *reinterpret_cast<uint4 *>(wdst + 48 * SDE_STRIDE) = n3;
        const uint2 cm = nm;
        const int cn = (dbg & 1) ? 0 : (c + 1 < npiece ? c + 1 : c);  // (the last round fetches its own piece again: nobody reads it)
        n0 = *reinterpret_cast<const uint4 *>(src0 + cn * SDE_PIECE);
        n1 = *reinterpret_cast<const uint4 *>(src1 + cn * SDE_PIECE);
        n2 = *reinterpret_cast<const uint4 *>(src2 + cn * SDE_PIECE);
        n3 = *reinterpret_cast<const uint4 *>(src3 + cn * SDE_PIECE);
        nm = *reinterpret_cast<const uint2 *>(mrow + cn * 4);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int s16 = c * 4 + s;
            if (s16 < n16 && !(dbg & 2)) {  // (wave-uniform: the query row ends with the alignment)
                const uint4 cw = *reinterpret_cast<const uint4 *>(wbuf + lane * SDE_STRIDE + s * 16);
                // the query's table-row offsets (residue index x 168 bytes) come ready-made as 16-bit values, 16 per site block
                // (two broadcast reads): the look-up's address is then ONE vector add with a word and a byte selected
                const uint4 qw0 = *reinterpret_cast<const uint4 *>(lq + s16 * 32), qw1 = *reinterpret_cast<const uint4 *>(lq + s16 * 32 + 16);
                const uint32_t rmask = ((s & 2 ? cm.y : cm.x) >> (16 * (s & 1))) & 0xffffu;
                valid += __popc(rmask & (uint32_t)lqm[s16]);
                const uint32_t rr[4] = {cw.x, cw.y, cw.z, cw.w};
                const uint32_t qq[8] = {qw0.x, qw0.y, qw0.z, qw0.w, qw1.x, qw1.y, qw1.z, qw1.w};
                double v[16];
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    const uint32_t r8 = (rr[k >> 2] >> (8 * (k & 3))) & 0xffu;
                    const uint32_t qr = (qq[k >> 1] >> (16 * (k & 1))) & 0xffffu;
                    v[k] = *reinterpret_cast<const double *>(Tb + qr + r8);
                }
#pragma unroll
                for (int k = 0; k < 16; ++k) tot += v[k];
            }
        }
    }
    if (valid == 0 || (double)valid / (double)L < overlap) return -1.0;
    if (ratio) *ratio = tot / (double)valid;
    const double r1 = 1 - tot / (double)valid;
    if (0 >= r1) return -1.0;
    return -log_libm(r1) * 1.3;
}

// slot-major copy of the packed reference rows for sd_eval64: rows[slot][Lrow] = residue index * 8 (160 = gap, also past the
// alignment), mrows[slot][Lrow / 16] = 16-bit gap masks
__global__ __launch_bounds__(APPLES_TPB) void k_sd_rows(const uint8_t *__restrict__ refa, const uint16_t *__restrict__ refm,
                                                        int64_t slots_pad, int n16, int Lrow, uint8_t *__restrict__ rows,
                                                        uint16_t *__restrict__ mrows) {
    const int64_t slot = (int64_t)blockIdx.x * APPLES_TPB + threadIdx.x;
    const int s16 = blockIdx.y;
    if (slot >= slots_pad) return;
    uint4 v = make_uint4(0xa0a0a0a0u, 0xa0a0a0a0u, 0xa0a0a0a0u, 0xa0a0a0a0u);
    uint16_t m = 0;
    if (s16 < n16) {
        v = *reinterpret_cast<const uint4 *>(refa + ((int64_t)s16 * slots_pad + slot) * 16);
        m = refm[(int64_t)s16 * slots_pad + slot];
    }
    *reinterpret_cast<uint4 *>(rows + slot * (int64_t)Lrow + s16 * 16) = v;
    mrows[slot * (int64_t)(Lrow / 16) + s16] = m;
}

// candidates -> survivors (see the head of the file).  One workgroup per query: the exclusive prefix of the segments' candidate
// counts goes to LDS once (the candidates are then one flat list in slot order), and from there the four wavefronts work
// on their own: a wavefront takes the next 64 candidates from a counter in LDS, evaluates them (sd_eval64) and writes each
// distance over its candidate's entry of seg_d -- -1 where the pair fails 0 <= d <= thr.  Nothing is moved: k_select_fast
// skips the entries with d < 0, and takes the number of survivors (what decides "enough inside the threshold or top-up
// rule") from n_surv[query].  (Closing the segments up in place took a barrier per round of 256 and a scan per 256
// segments: 5.4 ms per C4 pass against 1.5 for this form.)
__global__ __launch_bounds__(APPLES_TPB) void k_sd_exact(const uint8_t *__restrict__ rrows, const uint16_t *__restrict__ mrows,
                                                         int Lrow, const uint8_t *__restrict__ qa, const uint16_t *__restrict__ qm,
                                                         const double *__restrict__ table, int64_t n_slots, int64_t slots_pad,
                                                         int Lpad, int L, double overlap, double thr,
                                                         double *__restrict__ seg_d, const int32_t *__restrict__ seg_slot,
                                                         const int32_t *__restrict__ seg_cnt, int32_t *__restrict__ n_surv, int dbg) {
    constexpr int TPB = APPLES_TPB, NW = TPB / 64;
    __shared__ double T[21 * 21];
    __shared__ int sh_w[NW];
    __shared__ int sh_next;
    __shared__ __attribute__((aligned(16))) uint8_t sh_wbuf[NW][SDE_WBUF];
    __shared__ int sh_wslot[NW][64];
    extern __shared__ __attribute__((aligned(16))) uint8_t sh_dyn[];  // the query's table-row offsets (2 Lpad bytes), its masks, the prefix [n_seg + 1]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    for (int i = tid; i < 21 * 21; i += TPB) T[i] = table[i];
    const char *Tb = reinterpret_cast<const char *>(T);
    const int64_t q = blockIdx.x;
    const int n16 = Lpad / 16;
    const int n_seg = (int)(slots_pad >> 6);
    uint8_t *sh_q = sh_dyn;
    uint16_t *sh_qm = reinterpret_cast<uint16_t *>(sh_dyn + 2 * Lpad);
    int *pref = reinterpret_cast<int *>(sh_dyn + (2 * Lpad + Lpad / 8 + 15) / 16 * 16);
    for (int i = tid; i < Lpad; i += TPB) reinterpret_cast<uint16_t *>(sh_q)[i] = (uint16_t)(qa[q * (int64_t)Lpad + i] * 168u);
    for (int i = tid; i < n16; i += TPB) sh_qm[i] = qm[q * (int64_t)n16 + i];
    const int32_t *cnt = seg_cnt + q * (int64_t)n_seg;
    const int32_t *sslot = seg_slot + q * slots_pad;
    double *sd = seg_d + q * slots_pad;
    // exclusive prefix of the counts (coalesced in, summed out of LDS: k_select_fast's form)
    const int K = (n_seg + TPB - 1) / TPB;
    const int s_lo = tid * K, s_hi = s_lo + K < n_seg ? s_lo + K : n_seg;
    for (int s = tid; s < n_seg; s += TPB) pref[s] = cnt[s];
    __syncthreads();
    int local = 0;
    for (int s = s_lo; s < s_hi; ++s) { const int v = pref[s]; pref[s] = local; local += v; }
    int incl = local;
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(incl, o, 64);
        if (lane >= o) incl += t;
    }
    if (lane == 63) sh_w[w] = incl;
    if (tid == 0) sh_next = 0;
    __syncthreads();
    int base = 0, total = 0;
    for (int k = 0; k < NW; ++k) {
        if (k < w) base += sh_w[k];
        total += sh_w[k];
    }
    const int at = base + incl - local;
    for (int s = s_lo; s < s_hi; ++s) pref[s] += at;
    if (tid == 0) pref[n_seg] = total;
    __syncthreads();
    int kept = 0;
    for (;;) {
        if (dbg & 4) break;  // (timing experiments: the set-up alone)
        int e0 = 0;
        if (lane == 0) e0 = atomicAdd(&sh_next, 64);
        e0 = __builtin_amdgcn_readfirstlane(e0);
        if (e0 >= total) break;
        const int e = e0 + lane;
        const bool in = e < total;
        int lo = 0, hi = n_seg + 1;  // last segment whose prefix <= e (empty segments share a prefix: the last one wins, and holds e)
        const int ee = in ? e : 0;
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (pref[mid] <= ee) lo = mid; else hi = mid;
        }
        const int64_t src = (int64_t)lo * 64 + (ee - pref[lo]);
        const int slot = sslot[src];
        const bool ok = in && slot < n_slots;
        if (dbg & 8) { if (in) sd[src] = (double)slot; continue; }  // (timing experiments: no evaluation)
        const double d = sd_eval64(rrows, mrows, Lrow, ok ? slot : 0, sh_wbuf[w], sh_wslot[w], sh_q, sh_qm, n16, Tb, L, overlap, nullptr, dbg);
        if (in) {
            const bool keep = ok && d >= 0 && d <= thr;
            sd[src] = keep ? d : -1.0;
            kept += keep ? 1 : 0;
        }
    }
    for (int o = 32; o > 0; o >>= 1) kept += __shfl_down(kept, o, 64);
    __syncthreads();  // (sh_w is free)
    if (lane == 0) sh_w[w] = kept;
    __syncthreads();
    if (tid == 0) {
        int s = 0;
        for (int k = 0; k < NW; ++k) s += sh_w[k];
        n_surv[q] = s;
    }
}

// The top-up rule (apples/Reference.py:144-152) for a listed query without its full row: the `-b` nearest references by
// exact distance are among those whose LOWER BOUND does not exceed the `-b`-th smallest exact value.  Input: the query's row
// of bounds (k_sd_gemm<ROWS>: acc = 4 x the bound of tot).  key = acc / 4 / min(valid sites of the query, of the row) <=
// tot / valid = x (kept in single precision, within 2e-7: the comparisons carry a margin of 1e-6), and d grows with x.  (1) a histogram of the keys gives t0 with at least `-b` + 8 keys below it; those
// references are evaluated exactly (as k_scoredist would: same bits); (2) X = a bin edge at or above the `-b`-th smallest
// exact x among them, an upper bound of the true `-b`-th smallest x; (3) every other reference with key <= X is evaluated
// too.  Whatever was not evaluated has x > X: it cannot be among the `-b` nearest, nor inside the threshold (a listed query
// has fewer than `-b` references there).  Output: row r of `out` (r = list position) with the exact distances of the evaluated
// references and -1 (= missing) everywhere else -- k_select's top-up rule on it selects what it would select on the full row.
// Fewer than `-b` valid distances among the first set: everything is evaluated.
// COMPACT: the evaluated references leave as a short list instead of a row of n_slots values -- row r of `out` then holds
// len[r] distances (slot order) followed, from double `cap` on, by their slots as 32-bit integers (k_select_stream's compact
// rows); a query that evaluates more than `cap` references gets len[r] = -1 and goes on list2 for the row form of this
// kernel (keys_ready: its row of bounds holds the keys already).  The row form writes 8 n_slots bytes per query and the
// selection reads them back: 2.0 of config 4's 22.9 ms for 11 % of its queries.
#define SDT_BINS 4096
#define SDT_SCALE 2048.0
#define SDT_CAP 512  // entries of a compact row at most (6 KB of LDS).  k_sd_topup<true> per config 4 pass with 4 096 / 2 048 / 1 024 / 512 / 256
                     // entries: 2.40 / 1.50 / 1.30 / 0.99 / 0.96 ms (the lists' LDS decides how many workgroups a CU holds); the row form: 1.52
template <bool COMPACT>
__global__ __launch_bounds__(APPLES_TPB) void k_sd_topup(const uint8_t *__restrict__ rrows, const uint16_t *__restrict__ mrows,
                                                         int Lrow, const uint8_t *__restrict__ qa, const uint16_t *__restrict__ qm,
                                                         const double *__restrict__ table, int64_t n_slots, int64_t slots_pad,
                                                         int Lpad, int L, double overlap, const int32_t *__restrict__ qlist,
                                                         const int32_t *__restrict__ qcount, double *lbrows,
                                                         int64_t row_stride, const float *__restrict__ nvr,
                                                         const float *__restrict__ nvq, int64_t qrow0, int baseobs,
                                                         double *__restrict__ out_rows, int cap, int32_t *__restrict__ len,
                                                         int32_t *__restrict__ list2, int32_t *__restrict__ count2, int keys_ready) {
    constexpr int TPB = APPLES_TPB, NW = TPB / 64;
    __shared__ double T[21 * 21];
    __shared__ int hist[SDT_BINS];
    __shared__ int c_s[COMPACT ? SDT_CAP : 1];     // compact form: the evaluated references' slots ...
    __shared__ double c_d[COMPACT ? SDT_CAP : 1];  // ... and distances, in the order they were evaluated
    __shared__ int sh_cn;
    __shared__ int sh_w[NW];
    __shared__ int sh_bin;
    __shared__ int wqueue[NW][128];
    __shared__ __attribute__((aligned(16))) uint8_t sh_wbuf[NW][SDE_WBUF];
    __shared__ int sh_wslot[NW][64];
    extern __shared__ __attribute__((aligned(16))) uint8_t sh_q[];  // the query's table-row offsets (2 Lpad bytes), then its masks
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    for (int i = tid; i < 21 * 21; i += TPB) T[i] = table[i];
    const char *Tb = reinterpret_cast<const char *>(T);
    const int n16 = Lpad / 16;
    uint16_t *sh_qm = reinterpret_cast<uint16_t *>(sh_q + 2 * Lpad);
    const int n_list = *qcount;
    const double INF = __longlong_as_double(0x7ff0000000000000LL);
    for (int r = blockIdx.x; r < n_list; r += gridDim.x) {
        const int64_t q = __builtin_amdgcn_readfirstlane(qlist[r]);
        float *lb = reinterpret_cast<float *>(lbrows + q * row_stride);
        double *out = out_rows + (int64_t)r * row_stride;
        // (the previous list entry's last use lies behind its closing barrier)
        for (int i = tid; i < Lpad; i += TPB) reinterpret_cast<uint16_t *>(sh_q)[i] = (uint16_t)(qa[q * (int64_t)Lpad + i] * 168u);
        for (int i = tid; i < n16; i += TPB) sh_qm[i] = qm[q * (int64_t)n16 + i];
        const float nq_ = nvq[qrow0 + q];
        float *keys = lb;  // the row of bounds becomes the row of keys in place (pass A below)
        auto bin_of = [&](double x) -> int { return x >= (double)(SDT_BINS - 1) / SDT_SCALE ? SDT_BINS - 1 : (int)(x * SDT_SCALE); };
        // smallest bin whose cumulative count reaches `need` (SDT_BINS if the total does not)
        auto find_bin = [&](int need) -> int {
            __syncthreads();  // hist is complete
            constexpr int PER = SDT_BINS / TPB;
            int local = 0;
            for (int k = 0; k < PER; ++k) local += hist[tid * PER + k];
            int incl = local;
            for (int o = 1; o < 64; o <<= 1) {
                const int t = __shfl_up(incl, o, 64);
                if (lane >= o) incl += t;
            }
            if (lane == 63) sh_w[w] = incl;
            if (tid == 0) sh_bin = SDT_BINS;
            __syncthreads();
            int base = 0;
            for (int k = 0; k < w; ++k) base += sh_w[k];
            int cum = base + incl - local;  // count below this thread's bins
            if (cum < need && cum + local >= need) {
                for (int k = 0; k < PER; ++k) {
                    cum += hist[tid * PER + k];
                    if (cum >= need) { sh_bin = tid * PER + k; break; }
                }
            }
            __syncthreads();
            return sh_bin;
        };
        // exact distances of the references with lo < key <= hi, streamed through per-wavefront queues; with_hist: the valid ones'
        // ratios are counted into hist
        auto eval_range = [&](double lo, double hi, bool with_hist) {
            int head = 0, count = 0;  // wave-uniform ring of 128 slots
            auto drain = [&](int n) {
                const int slot = lane < n ? wqueue[w][(head + lane) & 127] : 0;
                double x = 0.0;
                const double d = sd_eval64(rrows, mrows, Lrow, slot, sh_wbuf[w], sh_wslot[w], sh_q, sh_qm, n16, Tb, L, overlap, &x);
                if (COMPACT) {  // (missing distances are simply absent from the list)
                    const bool keep = lane < n && d >= 0;
                    const unsigned long long km = __ballot(keep);
                    int at = 0;
                    if (lane == 0 && km) at = atomicAdd(&sh_cn, __popcll(km));
                    at = __shfl(at, 0, 64) + __popcll(km & ((1ull << lane) - 1ull));
                    if (keep && at < cap) { c_s[at] = slot; c_d[at] = d; }
                } else if (lane < n) {
                    out[slot] = d;
                }
                if (lane < n && with_hist && d >= 0) atomicAdd(&hist[bin_of(x)], 1);
                head = (head + n) & 127;
                count -= n;
            };
            // a wavefront streams blocks of 256 keys (four per lane), the next block's load leaving before this one is looked at
            const float flo = (float)lo, fhi = (float)hi;
            const int64_t step = (int64_t)NW * 256;
            int64_t s0 = (int64_t)w * 256 + lane * 4;
            float4 nxt = s0 < slots_pad ? *reinterpret_cast<const float4 *>(keys + s0) : make_float4(0.f, 0.f, 0.f, 0.f);
            for (; s0 - lane * 4 < n_slots; s0 += step) {
                const float4 cur = nxt;
                if (s0 + step < slots_pad) nxt = *reinterpret_cast<const float4 *>(keys + s0 + step);
                const float kk[4] = {cur.x, cur.y, cur.z, cur.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const bool take = s0 + j < n_slots && kk[j] > flo && kk[j] <= fhi;
                    const unsigned long long m = __ballot(take);
                    if (m) {
                        if (take) wqueue[w][(head + count + __popcll(m & ((1ull << lane) - 1ull))) & 127] = (int)(s0 + j);
                        count += __popcll(m);
                        if (count >= 64) drain(64);
                    }
                }
            }
            if (count > 0) drain(count);
        };
        // the row starts as "everything missing" (row form); the histogram empty
        if (!COMPACT)
            for (int64_t s = (int64_t)tid * 2; s < slots_pad; s += TPB * 2) *reinterpret_cast<double2 *>(out + s) = make_double2(-1.0, -1.0);
        for (int i = tid; i < SDT_BINS; i += TPB) hist[i] = 0;
        if (tid == 0) sh_cn = 0;
        __syncthreads();
        // pass A: key = bound / 4 / min(valid sites of the query, of the row) in single precision (within 2e-7 of the quotient:
        // the tests below carry a margin of 1e-6), written over the bound; +inf where no pair can be valid
        for (int64_t s = (int64_t)tid * 4; s < slots_pad; s += TPB * 4) {
            const float4 b4 = *reinterpret_cast<const float4 *>(lb + s), n4 = *reinterpret_cast<const float4 *>(nvr + s);
            const float bb[4] = {b4.x, b4.y, b4.z, b4.w}, nn[4] = {n4.x, n4.y, n4.z, n4.w};
            float kk[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float vub = nn[j] < nq_ ? nn[j] : nq_;
                kk[j] = keys_ready ? bb[j] : ((s + j < n_slots && vub > 0.f) ? bb[j] * 0.25f / vub : __int_as_float(0x7f800000));
                if (kk[j] < __int_as_float(0x7f800000)) atomicAdd(&hist[bin_of((double)kk[j])], 1);
            }
            if (!keys_ready) *reinterpret_cast<float4 *>(keys + s) = make_float4(kk[0], kk[1], kk[2], kk[3]);
        }
        __threadfence_block();
        const int b0 = find_bin(baseobs + 8);
        const double t0 = b0 >= SDT_BINS - 1 ? INF : (double)(b0 + 1) / SDT_SCALE;
        for (int i = tid; i < SDT_BINS; i += TPB) hist[i] = 0;
        __syncthreads();
        eval_range(-1.0, t0, true);
        if (t0 < INF) {
            const int b1 = find_bin(baseobs);
            const double X = b1 >= SDT_BINS - 1 ? INF : (double)(b1 + 1) / SDT_SCALE * (1.0 + 1e-6);
            if (X > t0) eval_range(t0, X, false);
        }
        __syncthreads();  // (hist, the queues and sh_bin are reused by the next list entry; sh_cn and the list are complete)
        if (COMPACT) {
            const int n = sh_cn;
            if (n > cap) {  // too many for a compact row: the row form takes this query
                if (tid == 0) { len[r] = -1; list2[atomicAdd(count2, 1)] = (int32_t)q; }
            } else {
                // slot order: an entry's place = the number of entries with a smaller slot (slots are distinct)
                int32_t *out_s = reinterpret_cast<int32_t *>(out + cap);
                for (int e = tid; e < n; e += TPB) {
                    const int se = c_s[e];
                    int rank = 0;
                    for (int f = 0; f < n; ++f) rank += c_s[f] < se;
                    out[rank] = c_d[e];
                    out_s[rank] = se;
                }
                if (tid == 0) len[r] = n;
            }
            __syncthreads();  // (the list is free for the next entry)
        }
    }
}

}  // namespace

// fp4 codes of the rounded-down table: code of the largest v in {0, .5, 1, 1.5, 2, 3, 4, 6} with v / 4 <= T[a][b]
// fp6: the same on the e2m3 grid (code = exponent << 3 | mantissa: m / 8 for exponent 0, 2^(e - 1) (1 + m / 8) above: 0 .. 7.5)
void sd_table_codes(const double *blosum20x20, uint8_t *codes /* [20][20] */, bool fp6) {
    static const double grid[8] = {0, 0.5, 1, 1.5, 2, 3, 4, 6};
    for (int i = 0; i < 400; ++i) {
        int c = 0;
        if (fp6) {
            for (int k = 0; k < 32; ++k) {
                const int e = k >> 3, m = k & 7;
                const double v = e == 0 ? m / 8.0 : std::ldexp(1.0 + m / 8.0, e - 1);
                if (v / 4.0 <= blosum20x20[i]) c = k;  // (the grid grows with the code)
            }
        } else {
            for (int k = 0; k < 8; ++k)
                if (grid[k] / 4.0 <= blosum20x20[i]) c = k;
        }
        codes[i] = (uint8_t)c;
    }
}

int64_t sd_query_image_bytes(const apples_ctx *ctx, int64_t rows256) {  // rows256: image rows, a multiple of 256
    return rows256 / 256 * sd_steps(ctx->aln.L) * (int64_t)(ctx->aln.sd_fp6 ? SD_IMG6 : SD_IMG);
}

int sd_steps(int L) { return (20 * L + 127) / 128; }

bool sd_gemm_usable(const apples_ctx *ctx) {
    // (a wide threshold keeps most pairs: nothing to filter; the full rows of k_scoredist are then the cheaper form)
    return ctx->aln.sd_ref4 != nullptr && ctx->params.filt_threshold <= SD_GEMM_MAX_THRESHOLD;
}

// d_mask / ref_stride: the rows' gap masks (k_pack_aa; ref_stride = slots_pad for the reference layout, 0 for the query
// layout) when d_nv is wanted
int launch_sd_expand(apples_ctx *ctx, const uint8_t *d_raw, int64_t n, uint8_t *d_out, int64_t n_img, hipStream_t st,
                     const int32_t *d_src_row, int64_t row0, bool query, float *d_nv, const int32_t *d_n, const uint16_t *d_mask,
                     int64_t ref_stride) {
    if (n_img <= 0) return 0;
    const int NB = sd_steps(ctx->aln.L);
    const int64_t total = n_img * NB * 4;
    hipLaunchKernelGGL(k_sd_expand, dim3((unsigned)((total + APPLES_TPB - 1) / APPLES_TPB)), dim3(APPLES_TPB), 0, st ? st : ctx->stream,
                       d_raw, n, ctx->aln.L, NB, reinterpret_cast<uint4 *>(d_out), n_img, d_src_row, row0,
                       query ? ctx->sd_tq4 : (const uint8_t *)nullptr, d_nv, d_n, query && ctx->aln.sd_fp6 ? 1 : 0);
    if (d_nv)
        hipLaunchKernelGGL(k_sd_nv, dim3((unsigned)((n_img + APPLES_TPB - 1) / APPLES_TPB)), dim3(APPLES_TPB), 0, st ? st : ctx->stream,
                           d_mask, n, n_img, (ctx->aln.L + 15) / 16, ref_stride, d_nv + row0);
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}

// candidates of queries [q0, q0 + nq) of the block: seg_slot / seg_cnt rows relative to q0 (seg_cnt preset to zero)
int launch_sd_filter(apples_ctx *ctx, const QueryBlock &qb, int64_t q0, int64_t nq, int32_t *seg_slot, int32_t *seg_cnt) {
    if (nq == 0) return 0;
    const DevAlign &a = ctx->aln;
    const int NB = sd_steps(a.L);
    const int TQ = (int)((nq + SD_T - 1) / SD_T), TR = (int)(a.slots_pad / SD_T);
    if (ctx->n_cu == 0) {
        hipDeviceProp_t prop;
        HIP_TRY(ctx, hipGetDeviceProperties(&prop, ctx->device));
        ctx->n_cu = prop.multiProcessorCount;
    }
    const int64_t grid = std::max(8, ctx->n_cu / 8 * 8);
    // d <= f <=> tot <= c valid with c = 1 - exp(-f / 1.3); the bound is 4 sum of the rounded-down values: margin 1e-6
    const double c = 1.0 - std::exp(-ctx->params.filt_threshold / 1.3);
    float k4c = (float)(4.0 * c * (1.0 + 1e-6));
    k4c = std::nextafterf(k4c, INFINITY);
    const int R = (NB - 2) % 3;
#define SD_LAUNCH(R_, F6_)                                                                                                     \
    hipLaunchKernelGGL((k_sd_gemm<R_, false, F6_>), dim3((unsigned)grid), dim3(512), 0, ctx->stream, a.sd_ref4, qb.sd_q4, q0,  \
                       a.slots_pad, NB, nq, TQ, TR, a.sd_nvr, qb.sd_nvq, k4c, seg_slot, seg_cnt, (const int32_t *)nullptr,      \
                       (const int32_t *)nullptr, (double *)nullptr, (int64_t)0)
    if (a.sd_fp6) { if (R == 0) SD_LAUNCH(0, true); else if (R == 1) SD_LAUNCH(1, true); else SD_LAUNCH(2, true); }
    else { if (R == 0) SD_LAUNCH(0, false); else if (R == 1) SD_LAUNCH(1, false); else SD_LAUNCH(2, false); }
#undef SD_LAUNCH
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}

int launch_sd_exact(apples_ctx *ctx, const QueryBlock &qb, int64_t q0, int64_t nq, double *seg_d, const int32_t *seg_slot,
                    const int32_t *seg_cnt, int32_t *n_surv) {
    if (nq == 0) return 0;
    const DevAlign &a = ctx->aln;
    const int Lpad = (a.L + 15) / 16 * 16;
    const size_t dyn = (size_t)((2 * Lpad + Lpad / 8 + 15) / 16 * 16) + ((size_t)(a.slots_pad >> 6) + 1) * sizeof(int);
    const int sd_dbg = (int)knob(ctx, "APPLES_SD_DBG", 0);  // timing experiments only (wrong results)
    hipLaunchKernelGGL(k_sd_exact, dim3((unsigned)nq), dim3(APPLES_TPB), dyn, ctx->stream, a.aa_rows, a.aa_mrows, a.aa_Lrow,
                       qb.aa_idx + q0 * Lpad, qb.aa_mask + q0 * (Lpad / 16), ctx->blosum, a.n_rows, a.slots_pad, Lpad, a.L,
                       ctx->params.overlap_frac, ctx->params.filt_threshold, seg_d, seg_slot, seg_cnt, n_surv, sd_dbg);
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}

// The top-up path of the fused route for the queries on a device list (entries relative to q0): their operand image in list
// order (scratch `img`, room for nq_max rows), their rows of bounds on the matrix cores (into the listed queries' own rows of
// `lbrows`, which the selection has consumed), then k_sd_topup: row r of out_rows = what k_select needs of list entry r's
// full row.
// compact rows hold `cap` entries: SDT_CAP, or what two thirds of a row's 8 n_slots bytes hold (12 bytes per entry)
int sd_compact_cap(const apples_ctx *ctx) {
    if (ctx->dbg & APPLES_DBG_SD_COMPACT_TINY) return 16;  // diagnostic switch: nearly every listed query overflows into the row form
    return (int)std::min<int64_t>(SDT_CAP, ctx->aln.slots_pad * 2 / 3 / 64 * 64);
}

// k_sd_topup alone for a device list whose rows of bounds hold keys already (the queries a compact pass could not hold): row form
int launch_sd_topup_rows_again(apples_ctx *ctx, const QueryBlock &qb, int64_t q0, int64_t nq_max, const int32_t *qlist,
                               const int32_t *qcount, double *lbrows, double *out_rows) {
    if (nq_max == 0) return 0;
    const DevAlign &a = ctx->aln;
    const int Lpad = (a.L + 15) / 16 * 16;
    const unsigned wgs = (unsigned)std::min<int64_t>(nq_max, (int64_t)ctx->n_cu * 8);
    hipLaunchKernelGGL(k_sd_topup<false>, dim3(wgs), dim3(APPLES_TPB), (size_t)(2 * Lpad + Lpad / 8), ctx->stream, a.aa_rows, a.aa_mrows, a.aa_Lrow,
                       qb.aa_idx + q0 * Lpad, qb.aa_mask + q0 * (Lpad / 16), ctx->blosum, a.n_rows, a.slots_pad, Lpad, a.L,
                       ctx->params.overlap_frac, qlist, qcount, lbrows, a.slots_pad, a.sd_nvr, qb.sd_nvq, q0, ctx->params.base_observation,
                       out_rows, 0, (int32_t *)nullptr, (int32_t *)nullptr, (int32_t *)nullptr, 1);
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}

// `len` != nullptr: compact rows (len[r] entries in row r of out_rows, -1 = on list2 / count2 for launch_sd_topup_rows_again)
int launch_sd_topup(apples_ctx *ctx, const QueryBlock &qb, int64_t q0, int64_t nq_max, const int32_t *qlist, const int32_t *qcount,
                    uint8_t *img, double *lbrows, double *out_rows, int32_t *len, int32_t *list2, int32_t *count2) {
    if (nq_max == 0) return 0;
    const DevAlign &a = ctx->aln;
    const int NB = sd_steps(a.L), Lpad = (a.L + 15) / 16 * 16;
    if (launch_sd_expand(ctx, qb.raw + q0 * a.L, nq_max, img, nq_max, ctx->stream, qlist, 0, true, nullptr, qcount)) return 1;
    const int TR = (int)(a.slots_pad / SD_T);
    if (ctx->n_cu == 0) {
        hipDeviceProp_t prop;
        HIP_TRY(ctx, hipGetDeviceProperties(&prop, ctx->device));
        ctx->n_cu = prop.multiProcessorCount;
    }
    const int64_t grid = std::max(8, ctx->n_cu / 8 * 8);
    const int R = (NB - 2) % 3;
#define SD_LAUNCH(R_, F6_)                                                                                                     \
    hipLaunchKernelGGL((k_sd_gemm<R_, true, F6_>), dim3((unsigned)grid), dim3(512), 0, ctx->stream, a.sd_ref4, img, (int64_t)0, \
                       a.slots_pad, NB, nq_max, 0, TR, a.sd_nvr, (const float *)nullptr, 0.f, (int32_t *)nullptr,             \
                       (int32_t *)nullptr, qlist, qcount, lbrows, a.slots_pad)
    if (a.sd_fp6) { if (R == 0) SD_LAUNCH(0, true); else if (R == 1) SD_LAUNCH(1, true); else SD_LAUNCH(2, true); }
    else { if (R == 0) SD_LAUNCH(0, false); else if (R == 1) SD_LAUNCH(1, false); else SD_LAUNCH(2, false); }
#undef SD_LAUNCH
    const unsigned wgs = (unsigned)std::min<int64_t>(nq_max, (int64_t)ctx->n_cu * 8);
    if (len) {
        HIP_TRY(ctx, hipMemsetAsync(count2, 0, sizeof(int32_t), ctx->stream));
        hipLaunchKernelGGL(k_sd_topup<true>, dim3(wgs), dim3(APPLES_TPB), (size_t)(2 * Lpad + Lpad / 8), ctx->stream, a.aa_rows, a.aa_mrows, a.aa_Lrow,
                           qb.aa_idx + q0 * Lpad, qb.aa_mask + q0 * (Lpad / 16), ctx->blosum, a.n_rows, a.slots_pad, Lpad, a.L,
                           ctx->params.overlap_frac, qlist, qcount, lbrows, a.slots_pad, a.sd_nvr, qb.sd_nvq, q0,
                           ctx->params.base_observation, out_rows, sd_compact_cap(ctx), len, list2, count2, 0);
    } else {
        hipLaunchKernelGGL(k_sd_topup<false>, dim3(wgs), dim3(APPLES_TPB), (size_t)(2 * Lpad + Lpad / 8), ctx->stream, a.aa_rows, a.aa_mrows, a.aa_Lrow,
                           qb.aa_idx + q0 * Lpad, qb.aa_mask + q0 * (Lpad / 16), ctx->blosum, a.n_rows, a.slots_pad, Lpad, a.L,
                           ctx->params.overlap_frac, qlist, qcount, lbrows, a.slots_pad, a.sd_nvr, qb.sd_nvq, q0,
                           ctx->params.base_observation, out_rows, 0, (int32_t *)nullptr, (int32_t *)nullptr, (int32_t *)nullptr, 0);
    }
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}

// slot-major copy of the packed reference rows (what sd_eval64 reads); a.aa_rows / a.aa_mrows allocated by the caller
int launch_sd_rows(apples_ctx *ctx) {
    const DevAlign &a = ctx->aln;
    const int Lpad = (a.L + 15) / 16 * 16;
    hipLaunchKernelGGL(k_sd_rows, dim3((unsigned)(a.slots_pad / APPLES_TPB), (unsigned)(a.aa_Lrow / 16)), dim3(APPLES_TPB), 0, ctx->stream,
                       a.aa_idx, a.aa_mask, a.slots_pad, Lpad / 16, a.aa_Lrow, a.aa_rows, a.aa_mrows);
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}
