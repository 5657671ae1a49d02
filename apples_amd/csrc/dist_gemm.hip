// The fused JC69 pass as a plain fp4 GEMM (default for ACGT- inputs, L <= 2046, singleton clusters).
//
// Same arithmetic as k_jc69_mfma (dist.hip): every site is a corner of a tetrahedron (components t1, t2, t3 =
// +-1, 0 for a gap) plus a validity flag v; sum t.t = 3 match - mism and sum v.v = valid.  Differences:
//   * BOTH operands are kept pre-expanded in HBM (1 byte per site: t1 and t2 only -- t3 is t1 with the sign flipped
//     where t2 is negative and the validity operand is t1 with the sign bits cleared, one bit operation per
//     fragment register each; the reference image is built once per context, 205 MB at 200 k x 1000), so the
//     kernel has no expansion arithmetic and no register -> LDS stores at all: the 256 x 64-byte tile images of a
//     64-site step arrive by LDS-DMA (global_load_lds_dwordx4) as contiguous kilobytes, stored in HBM as they lie
//     in LDS (chunk permutation for conflict-free ds_read_b128), three generations deep: the DMA runs two steps
//     ahead of the MFMAs;
//   * ONE accumulator set: the v component is multiplied through the block-scaled MFMA with a scale of 2^13 on
//     the query side, acc = sum t.t + 8192 valid.  With valid <= 2046 both integers decode exactly
//     (-valid <= sum t.t <= 3 valid: the ranges of neighbouring `valid` do not overlap, and acc < 2^24);
//   * which makes room for a 64 x 128 wavefront tile (2 x 4 MFMA tiles, 128 accumulator registers) and a
//     256 x 256 workgroup tile: 12 fragment reads and 4 DMA pieces per 32 MFMAs and wavefront (k_jc69_mfma: 16
//     reads, 4 wide stores and the expansion arithmetic per 16 MFMAs);
//   * persistent workgroups walk the tile grid in strips of 4 reference tiles per XCD.
// Output format is that of k_jc69_mfma<1> (threshold on the integers, ballot compaction per 64-slot segment, one
// packed word per survivor); the epilogue decodes on packed fp32 pairs and tests a host-verified linear threshold.
//
// Alignments of 2 047 to 4 092 sites (VS = 12, round 6): the validity sum rides at 2^12, acc = sum t.t + 4096 valid = 4099 valid -
// 4 mism, still below 2^24.  The two fields overlap now (sum t.t reaches 3 valid > 4096), but the pair decodes all the same:
// 0 <= mism <= valid puts valid into [acc / 4099, acc / 4095], an interval narrower than 4 while valid < 4099, and 4 mism = 4099
// valid - acc makes 3 valid = acc (mod 4), i.e. valid = 3 acc (mod 4): exactly one count in the interval.  Integer decode, the
// threshold through the table in LDS (a pass of 63 steps per tile pays for it: the epilogue is a few percent of the tile).
#include "common.h"

typedef int v4i_t __attribute__((ext_vector_type(4)));
typedef int v8i_t __attribute__((ext_vector_type(8)));
typedef float v16f_t __attribute__((ext_vector_type(16)));
typedef float v2f_t __attribute__((ext_vector_type(2)));

#define GM_T 256      // tile edge: queries and reference slots per workgroup
#ifndef GM_STRIP
#define GM_STRIP 4    // reference tiles per strip
#endif
#define GM_VSHIFT 13  // the validity sum rides at 2^13
#ifndef GM_AUX
#define GM_AUX 0       // cache policy bits of the DMA loads (experiments: 1 = sc0, 2 = nt)
#endif
#define GM_IMG (GM_T * 64)   // bytes of one tile-step image: 256 rows x (t1, t2) x 32 bytes

namespace {

__device__ __forceinline__ v16f_t mfma_f4(const v4i_t &a, const v4i_t &b, const v16f_t &c) {
    const v8i_t a8 = {a[0], a[1], a[2], a[3], 0, 0, 0, 0}, b8 = {b[0], b[1], b[2], b[3], 0, 0, 0, 0};
    return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, c, 4, 4, 0, 0, 0, 0);  // fp4 x fp4, unscaled
}

template <int VS>
__device__ __forceinline__ v16f_t mfma_f4_v(const v4i_t &a, const v4i_t &b, const v16f_t &c) {
    const v8i_t a8 = {a[0], a[1], a[2], a[3], 0, 0, 0, 0}, b8 = {b[0], b[1], b[2], b[3], 0, 0, 0, 0};
    // E8M0 scales: 127 + VS on the first operand, 127 (= 1.0) on the second
    return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, c, 4, 4, 0, 127 + VS, 0, 127);
}

// acc -> (valid, 4 mism) of the VS = 12 form (the file's header): valid = the count in [acc / 4099, acc / 4095] with valid = 3 acc (mod 4)
__device__ __forceinline__ void decode12(float S, int &valid, int &mism4) {
    const int A = (int)S;
    int v = (A + 4098) / 4099;
    v += (3 * A - v) & 3;
    valid = v;
    mism4 = 4099 * v - A;
}

// R = (NB - 2) % 3: the shape of the main loop's tail, fixed per launch (a run-time choice between the three tails
// merges 128 accumulator registers three ways and the kernel spills)
// QT = query rows per workgroup tile: 256 (eight wavefronts, one workgroup per CU) or 128 (four wavefronts, two
// workgroups per CU: one's epilogue and barrier stalls run beside the other's MFMAs, for half as much reuse of the
// reference image per byte moved)
template <bool LIN, int R, int QT, int VS = GM_VSHIFT>
__global__ __launch_bounds__(QT * 2, 256 / QT) void k_jc69_gemm(const uint8_t *__restrict__ rf4, const uint8_t *__restrict__ qf4,
                                                      int64_t qrow0, int64_t slots_pad, int NB, int64_t nq, int L, int TQ, int TR,
                                                      int32_t *__restrict__ seg_slot, int32_t *__restrict__ seg_cnt,
                                                      const int32_t *__restrict__ mmax, GemmThreshold lin) {
    // ONE LDS object (the compiler's alias analysis then sees constant, disjoint ranges and does not drain the DMA
    // queue before unrelated reads): three generations -- the QT query rows of one 64-site step (64 bytes each:
    // t1, t2), the 256 reference slots behind them -- and, for the table form of the threshold, the table
    constexpr int AI = QT * 64, GEN = AI + GM_IMG, NW = QT / 32, PB = 16 / NW;  // A image, generation, wavefronts, B pieces per wavefront
    constexpr int NTAB = VS == 12 ? 4096 : 2048;  // entries of the threshold table (valid counts the form can carry)
    static_assert(VS == GM_VSHIFT || (VS == 12 && !LIN), "the 2^12 form decodes on integers and tests through the table");
    __shared__ __attribute__((aligned(1024))) uint8_t lds[3 * GEN + (LIN ? 0 : NTAB * 4)];
#define Aq(g) (lds + (g) * GEN)
#define Br(g) (lds + (g) * GEN + AI)
    float *mm_lds = reinterpret_cast<float *>(lds + 3 * GEN);
    // Workgroups are persistent (one per CU) and walk the tile grid: workgroup ids go round the XCDs; XCD x takes the
    // strips x, x + 8, ... of GM_STRIP reference tiles, and within the XCD the tiles (query tile major, the strip's
    // reference tiles inside) are dealt round-robin to its workgroups, so that the 32 tiles in flight on an XCD are 8
    // consecutive query tiles x one strip.
    // (Whole strips dealt round the XCDs leave the first XCDs a strip more than the last -- 100 reference tiles against 96 at
    // config 3, 28 against 24 at config 4.  An eighth of the tiles per XCD, to within one tile, was measured on one box against
    // this: config 3's pass 27.2 -> 28.4 ms, config 4's 14.6 -> 14.5: the eight XCDs then stream regions a fixed 25 MB apart, and
    // an XCD that finishes early hands its share of the power budget to the others anyway.)
    const int xcd = blockIdx.x & 7;
    const int64_t per_strip = (int64_t)TQ * GM_STRIP, stride = gridDim.x >> 3;
    const int64_t l_end = (((TR + GM_STRIP - 1) / GM_STRIP + 7) / 8) * per_strip;
    auto tile_at = [&](int64_t l, int64_t &qt_, int64_t &rt_) __attribute__((always_inline)) {
        const int64_t strip = (l / per_strip) * 8 + xcd, within = l % per_strip;
        rt_ = strip * GM_STRIP + within % GM_STRIP;
        qt_ = within / GM_STRIP;
    };
    int64_t l = blockIdx.x >> 3, qt = 0, rt = 0;
    for (;; l += stride) {  // first tile
        if (l >= l_end) return;
        tile_at(l, qt, rt);
        if (rt < TR) break;
    }
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wq = wv >> 1, wr = wv & 1;
    if (!LIN)
        for (int i = tid; i < NTAB; i += QT * 2) mm_lds[i] = i <= L ? (float)(4 * mmax[i]) : -4.f;
    // DMA roles: a tile-step image is 1024 16-byte chunks = 16 pieces of 1 KB, stored in HBM exactly as it lies in
    // LDS (k_expand_queries_f4, compact form): row r's four chunks (t1 and t2, two 32-site words each) at 4 r, chunk c
    // in slot c ^ ((r >> 2) & 3), so that with the 64-byte stride the 16 lanes of a ds_read_b128 group cover all 16
    // bank quads.  This wavefront moves pieces wv * 2 + k; lane l of piece P fills chunk 64 P + l.  Reference tiles
    // are image tiles: whole contiguous kilobytes.  The query tile starts at image row qrow0 + q0, a multiple of 32
    // (of 256 for the sub-batches the driver cuts): its rows may lie in two image tiles, same slots.
    uint32_t doff[2];
    const uint8_t *qtile, *rtile;
    auto set_tile = [&](int64_t qt_, int64_t rt_) __attribute__((always_inline)) {
        const int64_t qabs = qrow0 + qt_ * QT;
        const int qin = (int)(qabs & 255);  // first row of the query tile inside its image tile
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int row = (wv * 2 + k) * 16 + (lane >> 2), slot = lane & 3, ar = qin + row;
            doff[k] = (uint32_t)(((ar >> 8) * NB * 1024 + (ar & 255) * 4 + slot) * 16);
        }
        qtile = qf4 + (qabs >> 8) * (int64_t)NB * GM_IMG;
        rtile = rf4 + rt_ * (int64_t)NB * GM_IMG;
    };
    set_tile(qt, rt);
    const uint32_t roff = (uint32_t)(wv * PB * 1024 + lane * 16);
    const int fr = lane & 31, fh = lane >> 5;
    int coff[2];  // byte offset of component c's chunk for this lane's row and K half
#pragma unroll
    for (int c = 0; c < 2; ++c) coff[c] = ((c * 2 + fh) ^ ((fr >> 2) & 3)) * 16;
    const int arow = (wq * 64 + fr) * 64, brow = (wr * 128 + fr) * 64;
    v16f_t acc[2][4];
    // this wavefront's pieces of step b's images -> generation g: its two of the query image (part 0), its PB of the
    // reference image two at a time (parts 1, 2)
    auto dma = [&](int b, int g, int part) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            if (part > 0 && (part - 1) * 2 + k >= PB) continue;
            const int kb = (part - 1) * 2 + k;
            // uniform base + 32-bit lane offset
            const uint8_t *src = part == 0 ? (qtile + b * GM_IMG) + doff[k] : (rtile + b * GM_IMG + kb * 1024) + roff;
            uint8_t *dst = part == 0 ? Aq(g) + (wv * 2 + k) * 1024 : Br(g) + (wv * PB + kb) * 1024;
#ifndef GM_NO_DMA
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)dst, 16, 0, GM_AUX);
#endif
        }
    };
    // Three fragment sets that rotate with the generations: in a step of generation g, set g holds t1 (and becomes
    // the validity operand in place: sign bits cleared), set g + 1 holds t2 and becomes t3 in place (t3 = t1 with
    // the sign flipped where t2 is negative: one three-input bit operation per register) and then receives the next
    // step's t1; set g + 2 is the next step's t2 set.  Only t1 and t2 are ever loaded.
    v4i_t fa[3][2], fb[3][4];
    auto load_frags = [&](int g, int c, int set) __attribute__((always_inline)) {
#ifndef GM_NO_FRAGS
        const uint8_t *A = Aq(g) + arow + coff[c], *B = Br(g) + brow + coff[c];
#pragma unroll
        for (int i = 0; i < 2; ++i) fa[set][i] = *reinterpret_cast<const v4i_t *>(A + i * 32 * 64);
#pragma unroll
        for (int j = 0; j < 4; ++j) fb[set][j] = *reinterpret_cast<const v4i_t *>(B + j * 32 * 64);
#endif
    };
    auto mfmas = [&](int set, bool valid) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = valid ? mfma_f4_v<VS>(fa[set][i], fb[set][j], acc[i][j]) : mfma_f4(fa[set][i], fb[set][j], acc[i][j]);
    };
    auto make_t3 = [&](int s1, int s2) __attribute__((always_inline)) {  // set s2 (t2) -> t3 = t1 ^ sign(t2)
#pragma unroll
        for (int i = 0; i < 2; ++i) fa[s2][i] = fa[s1][i] ^ (fa[s2][i] & (int)0x88888888);
#pragma unroll
        for (int j = 0; j < 4; ++j) fb[s2][j] = fb[s1][j] ^ (fb[s2][j] & (int)0x88888888);
    };
    auto strip_signs = [&](int s1) __attribute__((always_inline)) {  // t1 = +-1 or 0 -> v = |t1|: clear the sign bit of every fp4 nibble
#pragma unroll
        for (int i = 0; i < 2; ++i) fa[s1][i] &= 0x77777777;
#pragma unroll
        for (int j = 0; j < 4; ++j) fb[s1][j] &= 0x77777777;
    };
    // One step = 64 sites = 4 sections of 8 MFMAs per wavefront (t1, t2, t3, validity), generation g = step % 3.
    // Entry: set g holds t1 (read after the previous barrier).  The DMA of step + 2 goes out two pieces per section
    // between the MFMAs of the first two sections, into the generation every wavefront left at the previous barrier;
    // this step's barrier -- before the last section, whose MFMAs cover the first fragment reads of the next step --
    // needs step + 1 landed (issued a whole step ago) and lets those four pieces stay in flight.
    auto step = [&](int b, int g, bool feed, bool more) __attribute__((always_inline)) {
        const int gn = g == 2 ? 0 : g + 1, gf = g == 0 ? 2 : g - 1;  // next step's generation; the free one
        load_frags(g, 1, gn);
        if (feed) dma(b + 2, gf, 0);
        mfmas(g, false);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (k < 6) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            if (k == 1 || k == 4) {
                __builtin_amdgcn_sched_group_barrier(0x006, 4, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (feed) dma(b + 2, gf, 1);
        mfmas(gn, false);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (k == 1 || k == 4) {
                __builtin_amdgcn_sched_group_barrier(0x006, 4, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        make_t3(g, gn);
        if (feed && PB > 2) dma(b + 2, gf, 2);
        mfmas(gn, false);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (PB > 2 && (k == 2 || k == 5)) {
                __builtin_amdgcn_sched_group_barrier(0x006, 4, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        // step b + 1 has landed everywhere (own pieces first, then the barrier); step b's images are free
#ifdef GM_NO_VMWAIT
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#else
        if (feed && PB == 2) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
        else if (feed) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#endif
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        strip_signs(g);
        if (more) load_frags(gn, 0, gn);
        mfmas(g, true);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (k < 6) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    const int64_t n_seg = slots_pad >> 6;
    const uint32_t below = (1u << fr) - 1u;
    dma(0, 0, 0); dma(0, 0, 1); dma(0, 0, 2);
    dma(1, 1, 0); dma(1, 1, 1); dma(1, 1, 2);
    for (;;) {  // tiles of this workgroup; entry: the first two steps of the tile are on their way
    const int64_t r0 = rt * GM_T, q0 = qt * QT;  // (q0: relative to this launch's first query)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int x = 0; x < 16; ++x) acc[i][j][x] = LIN ? -2049.f : 0.f;  // (LIN: the decode's first addend rides along, see the epilogue)
    // steps 0 and 1 landed (they had the previous tile's epilogue to do so; the epilogue's stores share the counter,
    // hence no counted wait here), the table is written
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    load_frags(0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    // NB (even, >= 2) steps: NB - 2 feeding ones, then one that only prefetches fragments, then the last
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
    // the next tile's first two steps go out before the epilogue: after the last barrier nobody reads LDS any more
    int64_t nl = l + stride, nqt = 0, nrt = 0;
    bool have = false;
    for (; nl < l_end; nl += stride) {
        tile_at(nl, nqt, nrt);
        if (nrt < TR) { have = true; break; }
    }
    if (have) {
        set_tile(nqt, nrt);
        dma(0, 0, 0); dma(0, 0, 1); dma(0, 0, 2);
        dma(1, 1, 0); dma(1, 1, 1); dma(1, 1, 2);
    }
#ifdef GM_SKIP_EPILOGUE
    {   // timing experiment: main loop only (every accumulator stays live)
        int a = 0;
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 4; ++j) for (int x = 0; x < 16; ++x) a += (int)acc[i][j][x];
        if (a == 0x7fffffff) seg_cnt[0] = 1;
    }
#else
    // C layout of the 32x32 tiles: column (reference slot) = lane & 31, row (query) = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5).
    // acc = sum t.t + 8192 valid.  Decoding, every step exact in f32: 8192 valid = ((acc - 2049) + 2^36) - 2^36 (the
    // sum rounds to the nearest multiple of 8192; acc + 2047 is never a multiple of 8192 while valid <= 2046),
    // 4 mism = 3 valid - sum t.t = 8195 valid - acc.  The test mism <= mmax[valid]: LIN -- mmax[valid] =
    // floor(p_f valid) from valid = vmin on, evaluated as 4 mism <= slope * 8192 valid + off with constants the
    // host verified against the table for every valid (api.hip gemm_threshold), two elements per packed
    // instruction and the compare's lane mask used as it is; otherwise through the table in LDS (-4 where
    // nothing passes).
    if (LIN) {
        // (the accumulators start at -2049, so S below is acc - 2049 already: 8192 valid = (S + 2^36) - 2^36, and
        // m = 8195 valid - S = 4 mism + 2049 is compared with t = slope * 8192 valid + (off + 2049), the form the host verified)
        const v2f_t kC = {68719476736.f, 68719476736.f}, kM = {8195.f / 8192.f, 8195.f / 8192.f};
        const v2f_t kS = {lin.slope, lin.slope}, kO = {lin.off, lin.off};
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
                for (int xp = 0; xp < 8; ++xp) {
                    const v2f_t S0 = {acc[i][2 * s][2 * xp], acc[i][2 * s][2 * xp + 1]};
                    const v2f_t S1 = {acc[i][2 * s + 1][2 * xp], acc[i][2 * s + 1][2 * xp + 1]};
                    const v2f_t v0 = (S0 + kC) - kC, v1 = (S1 + kC) - kC;  // 8192 valid
                    const v2f_t m0 = __builtin_elementwise_fma(v0, kM, -S0), m1 = __builtin_elementwise_fma(v1, kM, -S1);  // 4 mism
                    const v2f_t t0 = __builtin_elementwise_fma(v0, kS, kO), t1 = __builtin_elementwise_fma(v1, kS, kO);
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const int x = 2 * xp + e, cx = (x & 3) + 8 * (x >> 2);  // this register's query, relative to qbase
                        // (rows past this launch's queries are looked at only where something passes)
                        const bool c0 = m0[e] <= t0[e], c1 = m1[e] <= t1[e];
                        if (__ballot(c0 || c1) == 0ull) continue;
                        const bool in = cx < rem;
                        const bool k0 = c0 && in && v0[e] >= lin.vmin8, k1 = c1 && in && v1[e] >= lin.vmin8;
                        const unsigned long long b0 = __ballot(k0), b1 = __ballot(k1);
                        if ((b0 | b1) == 0) continue;  // the counts stay at their preset zero
                        // this lane half's query: slots 0..31 of the segment from tile 0, 32..63 from tile 1
                        const uint32_t lo = (uint32_t)(b0 >> (32 * fh)), hi = (uint32_t)(b1 >> (32 * fh));
                        int32_t *row = row0 + (int64_t)cx * slots_pad;
                        if (k0) {
                            const int valid = (int)(v0[e] * (1.f / 8192.f)), mism = ((int)m0[e] - 2049) >> 2;
                            row[__popc(lo & below)] = (int32_t)(((uint32_t)fr << 26) | ((uint32_t)valid << 13) | (uint32_t)mism);
                        }
                        if (k1) {
                            const int valid = (int)(v1[e] * (1.f / 8192.f)), mism = ((int)m1[e] - 2049) >> 2;
                            row[__popc(lo) + __popc(hi & below)] = (int32_t)(((uint32_t)(32 + fr) << 26) | ((uint32_t)valid << 13) | (uint32_t)mism);
                        }
                        if (fr == 0 && in) cnt0[(int64_t)cx * n_seg] = __popc(lo) + __popc(hi);
                    }
                }
            }
        }
    } else {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int64_t qbase = q0 + wq * 64 + i * 32 + 4 * fh;  // this lane half's first query of the 32
        // (rows past this launch's queries may be real queries of the next sub-batch: not ours to write)
        const int rem = (int)(nq - qbase < 32 ? nq - qbase : 32);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            uint32_t keepbits = 0;
#pragma unroll
            for (int x = 0; x < 16; ++x)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const float S = acc[i][2 * s + j][x];
                    if (VS == 12) {
                        int v, m4;
                        decode12(S, v, m4);
                        keepbits |= ((float)m4 <= mm_lds[v] ? 1u : 0u) << (x * 2 + j);
                    } else {
                        const float valid = __builtin_floorf((S + 2047.f) * (1.f / 8192.f));
                        const float mism4 = __builtin_fmaf(valid, 8195.f, -S);
                        keepbits |= (mism4 <= mm_lds[(int)valid] ? 1u : 0u) << (x * 2 + j);
                    }
                }
            if (__ballot(keepbits != 0) == 0ull) continue;  // no survivor among these 32 queries x 64 slots
            const int64_t seg = (r0 + wr * 128 + s * 64) >> 6;
            int32_t *row0 = seg_slot + qbase * slots_pad + seg * 64;
            int32_t *cnt0 = seg_cnt + qbase * n_seg + seg;
#pragma unroll
            for (int x = 0; x < 16; ++x) {
                const int cx = (x & 3) + 8 * (x >> 2);  // this register's query, relative to qbase
                const bool in = cx < rem;
                const bool k0 = ((keepbits >> (x * 2)) & 1u) && in, k1 = ((keepbits >> (x * 2 + 1)) & 1u) && in;
                const unsigned long long b0 = __ballot(k0), b1 = __ballot(k1);
                if ((b0 | b1) == 0) continue;  // the counts stay at their preset zero
                const uint32_t lo = (uint32_t)(b0 >> (32 * fh)), hi = (uint32_t)(b1 >> (32 * fh));
                int32_t *row = row0 + (int64_t)cx * slots_pad;
                if (k0) {
                    const float S = acc[i][2 * s][x];
                    int valid, mism;
                    if (VS == 12) { decode12(S, valid, mism); mism >>= 2; }
                    else { valid = (int)__builtin_floorf((S + 2047.f) * (1.f / 8192.f)); mism = (8195 * valid - (int)S) >> 2; }
                    row[__popc(lo & below)] = (int32_t)(((uint32_t)fr << 26) | ((uint32_t)valid << 13) | (uint32_t)mism);
                }
                if (k1) {
                    const float S = acc[i][2 * s + 1][x];
                    int valid, mism;
                    if (VS == 12) { decode12(S, valid, mism); mism >>= 2; }
                    else { valid = (int)__builtin_floorf((S + 2047.f) * (1.f / 8192.f)); mism = (8195 * valid - (int)S) >> 2; }
                    row[__popc(lo) + __popc(hi & below)] = (int32_t)(((uint32_t)(32 + fr) << 26) | ((uint32_t)valid << 13) | (uint32_t)mism);
                }
                if (fr == 0 && in) cnt0[(int64_t)cx * n_seg] = __popc(lo) + __popc(hi);
            }
        }
    }
    }
#endif
    if (!have) break;
    l = nl; qt = nqt; rt = nrt;
    }
}

}  // namespace

bool dist_gemm_usable(const apples_ctx *ctx) {
    // (APPLES_DBG_NO_DIST_GEMM: no reference image is built, the bit-plane-fed MFMA kernel runs instead)
    return !(ctx->dbg & APPLES_DBG_NO_DIST_GEMM) && ctx->aln.ref_f4 && ctx->aln.L <= GEMM_MAX_L;
}

int launch_counts_gemm(apples_ctx *ctx, const QueryBlock &qb, int64_t q0, int64_t nq, int32_t *seg_slot, int32_t *seg_cnt) {
    const DevAlign &a = ctx->aln;
    // tuning knob: 128 = two four-wavefront workgroups per CU (measured at C3: 30.8 ms against 29.4 -- what the second
    // workgroup hides of the first one's epilogue and barrier stalls costs more in DMA traffic)
    const int qt_env = (int)knob(ctx, "APPLES_GEMM_QT", 256);
    const int QT = qt_env == 128 ? 128 : 256;
    const int TQ = (int)((nq + QT - 1) / QT), TR = (int)(a.slots_pad / GM_T);
    // persistent workgroups: as many as the CUs hold (the kernel's LDS and registers allow 256 / QT per CU), a
    // multiple of the 8 XCDs
    if (ctx->n_cu == 0) {
        hipDeviceProp_t prop;
        HIP_TRY(ctx, hipGetDeviceProperties(&prop, ctx->device));
        ctx->n_cu = prop.multiProcessorCount;
    }
    const int cus = (int)knob(ctx, "APPLES_GEMM_CUS", 0);  // experiment: leave CUs to a concurrent sweep
    const int per_cu = (int)knob(ctx, "APPLES_GEMM_WGS_PER_CU", 0);  // experiment: one four-wavefront workgroup per CU (QT = 128)
    const int64_t grid = std::max(8, (cus > 0 ? std::min(cus, ctx->n_cu) : ctx->n_cu) / 8 * 8) * (per_cu > 0 ? std::min(per_cu, 256 / QT) : 256 / QT);
    const bool table = knob_on(ctx, "APPLES_GEMM_TABLE");  // diagnostic knob: threshold through the LDS table
    const bool lin = ctx->gemm_thr.ok && !table;
    const int R = (a.G * 2 - 2) % 3;
    if (a.L > GEMM_MAX_L13) {  // 2 047 .. 4 092 sites: the validity sum at 2^12, integer decode, the threshold through the table
#define GM_LAUNCH12(R_, QT_)                                                                                         \
    hipLaunchKernelGGL((k_jc69_gemm<false, R_, QT_, 12>), dim3((unsigned)grid), dim3(QT_ * 2), 0, ctx->stream, a.ref_f4, \
                       qb.qf4, q0, a.slots_pad, a.G * 2, nq, a.L, TQ, TR, seg_slot, seg_cnt, ctx->jc_mmax, ctx->gemm_thr)
        if (QT == 256) { if (R == 0) GM_LAUNCH12(0, 256); else if (R == 1) GM_LAUNCH12(1, 256); else GM_LAUNCH12(2, 256); }
        else { if (R == 0) GM_LAUNCH12(0, 128); else if (R == 1) GM_LAUNCH12(1, 128); else GM_LAUNCH12(2, 128); }
#undef GM_LAUNCH12
        HIP_TRY(ctx, hipGetLastError());
        return 0;
    }
#define GM_LAUNCH(LIN_, R_, QT_)                                                                                     \
    hipLaunchKernelGGL((k_jc69_gemm<LIN_, R_, QT_>), dim3((unsigned)grid), dim3(QT_ * 2), 0, ctx->stream, a.ref_f4,  \
                       qb.qf4, q0, a.slots_pad, a.G * 2, nq, a.L, TQ, TR, seg_slot, seg_cnt,                         \
                       ctx->jc_mmax, ctx->gemm_thr)
#define GM_LAUNCH_R(LIN_, QT_)                                                                                       \
    do { if (R == 0) GM_LAUNCH(LIN_, 0, QT_); else if (R == 1) GM_LAUNCH(LIN_, 1, QT_); else GM_LAUNCH(LIN_, 2, QT_); } while (0)
    if (QT == 256) { if (lin) GM_LAUNCH_R(true, 256); else GM_LAUNCH_R(false, 256); }
    else { if (lin) GM_LAUNCH_R(true, 128); else GM_LAUNCH_R(false, 128); }
#undef GM_LAUNCH_R
#undef GM_LAUNCH
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}
