// Distance kernels: one fp64 distance per (query, alignment row).
//
// JC69 (apples/distance.py:718-745): per pair two integers, valid = #sites where neither is '-',
// mism = #valid sites whose bytes differ; then p = mism/valid and -0.75*ln(1 - 4p/3).  The integer
// part is bit-plane arithmetic: per 32 sites  M = qM & rM ; X = OR_p(qC_p ^ rC_p) ; mism += popc(X & M).
// A thread owns one reference row (coalesced 16 B/lane reads of the plane-major layout), a
// workgroup owns 256 rows x TQ queries; the query words are wave-uniform, so they come through
// the scalar cache and occupy SGPRs, not LDS.  HBM traffic per launch is the packed reference
// once per query tile plus 8 B per pair of output.  No MFMA: this is popcount, not a contraction.
#include <algorithm>
#include <cmath>
#include <cstdlib>

#include "common.h"
#include "libm_log.h"

__device__ __forceinline__ double jc69_from_counts(uint32_t mism, uint32_t valid, int L, double overlap,
                                                   const double *__restrict__ lut) {
    if (lut) return lut[(int64_t)valid * (valid + 1) / 2 + mism];
    // apples/distance.py:735-745, literal order
    if (valid == 0 || (double)valid / (double)L < overlap) return -1.0;
    double p = (double)mism * 1.0 / (double)valid;
    if (p - 2.220446049250313e-16 < 0) return 0.0;
    double loc = 1 - (4 * p / 3);
    if (0 >= loc) return -1.0;
    return -0.75 * log_libm(loc);
}

// popcount with the instruction's own accumulate operand (D = popc(S0) + S1); hipcc otherwise emits
// v_bcnt ..., 0 followed by separate adds
__device__ __forceinline__ uint32_t bcnt_acc(uint32_t x, uint32_t acc) {
    uint32_t r;
    asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(acc));
    return r;
}

// MODE 0: full rows  -> dist[q][slot] (and counts) for queries q0..q0+nq-1 of the block
// MODE 1: fused      -> only entries with 0 <= d <= thr are kept: every wavefront (64 consecutive
//                       slots = one segment) writes them, in slot order, at the start of its segment
//                       of seg_slot/seg_d and their number to seg_cnt[q][segment]
// MODE 2: listed     -> full rows for the queries named by qlist[0..*qcount) (top-up path); row r of
//                       dist belongs to qlist[r].  Also the smallest (d, representative index) key of
//                       every 64-slot segment -> segmin_d/segmin_i[r][segment], so that the top-up
//                       selection can find the `-b` nearest without streaming the row again
template <int P, int TQ, int MODE, bool ASM>
__global__ __launch_bounds__(APPLES_TPB) void k_jc69(const uint4 *__restrict__ refp, const uint4 *__restrict__ qp,
                                                     double *__restrict__ dist, uint32_t *__restrict__ counts,
                                                     int64_t n_slots, int64_t slots_pad, int G, int64_t nq, int L,
                                                     double overlap, const double *__restrict__ lut, double thr,
                                                     int32_t *__restrict__ seg_slot, int32_t *__restrict__ seg_cnt,
                                                     const int32_t *__restrict__ qlist,
                                                     const int32_t *__restrict__ qcount,
                                                     const int32_t *__restrict__ mmax,
                                                     const int32_t *__restrict__ slot_rep, double *__restrict__ segmin_d,
                                                     int32_t *__restrict__ segmin_i) {
    const int64_t slot = (int64_t)blockIdx.x * APPLES_TPB + threadIdx.x;
    if (MODE == 2) nq = *qcount;
    // listed mode: the list length lives on the device, so a bounded grid.y loops over the tiles
    for (int64_t q0 = (int64_t)blockIdx.y * TQ; q0 < nq; q0 += (MODE == 2 ? (int64_t)gridDim.y * TQ : nq)) {
    uint32_t nv[TQ], nm[TQ];
#pragma unroll
    for (int t = 0; t < TQ; ++t) nv[t] = nm[t] = 0;
    int64_t qi[TQ];
#pragma unroll
    for (int t = 0; t < TQ; ++t) qi[t] = (MODE == 2) ? (int64_t)qlist[(q0 + t < nq) ? q0 + t : q0] : q0 + t;
    for (int g = 0; g < G; ++g) {
        const uint4 *rp = refp + ((int64_t)g * (P + 1)) * slots_pad + slot;
        uint4 rm = rp[0];
        uint4 rc[P];
#pragma unroll
        for (int p = 0; p < P; ++p) rc[p] = rp[(int64_t)(1 + p) * slots_pad];
        // query words are wave-uniform (scalar loads).  Software pipeline: the words of the next
        // CH queries are requested before the current CH are consumed, so the scalar-cache latency
        // hides behind ~50 VALU instructions instead of stalling every query.
        constexpr int CH = (TQ >= 2) ? 2 : 1;
        uint4 cur[CH][P + 1], nxt[CH][P + 1];
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const uint4 *qq = qp + (((qi[c] >> 4) * G + g) * 16 + (qi[c] & 15)) * (P + 1);
#pragma unroll
            for (int p = 0; p <= P; ++p) cur[c][p] = qq[p];
        }
#pragma unroll
        for (int t0 = 0; t0 < TQ; t0 += CH) {
            if (t0 + CH < TQ) {
#pragma unroll
                for (int c = 0; c < CH; ++c) {
                    const int64_t qn = qi[(t0 + CH + c) < TQ ? (t0 + CH + c) : 0];
                    const uint4 *qq = qp + (((qn >> 4) * G + g) * 16 + (qn & 15)) * (P + 1);
#pragma unroll
                    for (int p = 0; p <= P; ++p) nxt[c][p] = qq[p];
                }
            }
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                const int t = t0 + c;
                uint4 qm = cur[c][0];
                uint4 x = make_uint4(0, 0, 0, 0);
#pragma unroll
                for (int p = 0; p < P; ++p) {
                    uint4 qc = cur[c][1 + p];
                    x.x |= qc.x ^ rc[p].x; x.y |= qc.y ^ rc[p].y; x.z |= qc.z ^ rc[p].z; x.w |= qc.w ^ rc[p].w;
                }
                uint32_t m0 = qm.x & rm.x, m1 = qm.y & rm.y, m2 = qm.z & rm.z, m3 = qm.w & rm.w;
                if (ASM) {
                    nv[t] = bcnt_acc(m3, bcnt_acc(m2, bcnt_acc(m1, bcnt_acc(m0, nv[t]))));
                    nm[t] = bcnt_acc(x.w & m3, bcnt_acc(x.z & m2, bcnt_acc(x.y & m1, bcnt_acc(x.x & m0, nm[t]))));
                } else {
                    nv[t] += __popc(m0) + __popc(m1) + __popc(m2) + __popc(m3);
                    nm[t] += __popc(x.x & m0) + __popc(x.y & m1) + __popc(x.z & m2) + __popc(x.w & m3);
                }
            }
#pragma unroll
            for (int c = 0; c < CH; ++c)
#pragma unroll
                for (int p = 0; p <= P; ++p) cur[c][p] = nxt[c][p];
        }
    }
    if (MODE == 1) {
        const int lane = threadIdx.x & 63;
        // the segment index is wave-uniform: keep it (and the row offsets built on it) in SGPRs
        const int64_t seg = __builtin_amdgcn_readfirstlane((int)(slot >> 6));
        const int64_t n_seg = slots_pad >> 6;
#pragma unroll
        for (int t = 0; t < TQ; ++t) {
            if (q0 + t < nq) {  // wave-uniform
                // 0 <= d <= thr decided on the integers when the host supplied the table: mmax[valid] is
                // the largest mismatch count whose tabulated distance passes (d is monotone in mism)
                double d = -1.0;
                bool keep;
                if (mmax) {
                    keep = slot < n_slots && (int)nm[t] <= mmax[nv[t]];
                    if (keep) d = lut[(int64_t)nv[t] * (nv[t] + 1) / 2 + nm[t]];
                } else {
                    if (slot < n_slots) d = jc69_from_counts(nm[t], nv[t], L, overlap, lut);
                    keep = d >= 0 && d <= thr;
                }
                const unsigned long long m = __ballot(keep);
                if (keep) {
                    const int64_t o = (q0 + t) * slots_pad + seg * 64 + __popcll(m & ((1ull << lane) - 1ull));
                    seg_slot[o] = (int32_t)slot;
                    dist[o] = d;
                }
                if (lane == 0) seg_cnt[(q0 + t) * n_seg + seg] = __popcll(m);
            }
            __builtin_amdgcn_sched_barrier(0);  // keep the 32 epilogues from being interleaved (VGPR pressure)
        }
        return;
    }
    if (MODE == 2 && segmin_d) {
        const int lane = threadIdx.x & 63;
        const int64_t seg = slot >> 6;
        const int my_rep = slot < n_slots ? slot_rep[slot] : 0x7fffffff;
#pragma unroll
        for (int t = 0; t < TQ; ++t) {
            if (q0 + t < nq) {  // wave-uniform
                const int64_t o = (q0 + t) * slots_pad + slot;
                double d = -1.0;
                if (slot < n_slots) {
                    d = jc69_from_counts(nm[t], nv[t], L, overlap, lut);
                    dist[o] = d;
                }
                double kd = d >= 0 ? d : __longlong_as_double(0x7ff0000000000000LL);
                int ki = d >= 0 ? my_rep : 0x7fffffff;
                for (int sft = 32; sft > 0; sft >>= 1) {
                    const double d2 = __hiloint2double(__shfl_down(__double2hiint(kd), sft, 64), __shfl_down(__double2loint(kd), sft, 64));
                    const int i2 = __shfl_down(ki, sft, 64);
                    if (d2 < kd || (d2 == kd && i2 < ki)) { kd = d2; ki = i2; }
                }
                if (lane == 0) {
                    segmin_d[(q0 + t) * slots_pad + seg] = kd;
                    segmin_i[(q0 + t) * slots_pad + seg] = ki;
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    } else if (slot < n_slots) {
#pragma unroll
        for (int t = 0; t < TQ; ++t) {
            if (q0 + t < nq) {
                int64_t o = (q0 + t) * slots_pad + slot;
                if (dist) dist[o] = jc69_from_counts(nm[t], nv[t], L, overlap, lut);
                if (counts) counts[o] = (nm[t] << 16) | nv[t];
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    }
}

// ---------------------------------------------------------------------------------------------
// Matrix-core form of the same counts (ACGT fast path, tiled case).  The bit-plane kernel above is
// bound by VALU issue: 6 lane-ops per 32 sites and pair.  Counting is a bilinear form: give every
// site a vector t in {(1,1,1),(1,-1,-1),(-1,1,-1),(-1,-1,1)} (the corners of a tetrahedron, one per
// nucleotide; zero for a gap) and a validity flag v.  Then for a pair of rows
//     sum t_r . t_q = 3*match - mism        sum v_r * v_q = valid = match + mism
// so mism = (3*valid - sum t.t) / 4, exactly.  The values 0, +1, -1 are exact in fp4 (e2m1: 0x0, 0x2,
// 0xA) and the sums (< 2^24) are exact in the f32 accumulators, so the densest matrix-core format
// serves: one v_mfma_f32_32x32x64_f8f6f4 (fp4 operands, no scaling) covers 64 sites of one
// component for a 32 x 32 block of pairs.  The order of the 64 nibbles inside a K chunk is
// irrelevant as long as both operands use the same one; the one used here makes the expansion two
// ALU operations per 8 sites: dword j of a 32-site word holds sites j, j+4, j+8, ... in nibbles 0..7.
// Workgroup tile: 256 queries x 128 reference slots, eight wavefronts of 64 x 64 (2 x 2 MFMA tiles,
// two accumulator sets): every expanded reference nibble feeds 256 queries.  Queries arrive
// pre-expanded (2 bytes/site, k_expand_queries_f4); reference rows are expanded from their bit
// planes on the fly into LDS, behind the MFMAs.
#include "jc69_f4.h"

// query rows -> fp4 operand image: a 64-site block of a query is 128 bytes, component c (t1, t2, t3,
// v) at c * 32, its first word's four dwords then its second word's
// (src_row: row r of the image comes from raw row src_row[r] -- the reference image in slot order)
// compact (the images dist_gemm.hip reads): t1 and t2 only (that kernel derives t3 and the validity operand), and
// the bytes stored as its LDS image of a 256-row tile and 64-site step -- image row row0 + q is row rr = (row0 + q)
// & 255 of tile (row0 + q) >> 8; a tile-step is 1024 16-byte chunks, chunk c (component * 2 + word) of row rr at
// position 4 rr + (c ^ ((rr >> 2) & 3)): the kernel's DMA pieces are then whole contiguous kilobytes
__global__ __launch_bounds__(APPLES_TPB) void k_expand_queries_f4(const uint8_t *__restrict__ raw, int64_t n, int L, int NB,
                                                                  uint32_t *__restrict__ out, int64_t n_pad,
                                                                  const int32_t *__restrict__ src_row, int compact,
                                                                  int64_t row0) {
    const int64_t idx = (int64_t)blockIdx.x * APPLES_TPB + threadIdx.x;  // one thread per (query, block, word, dword)
    const int64_t total = n_pad * NB * 8;
    if (idx >= total) return;
    const int j = (int)(idx & 3), x = (int)((idx >> 2) & 1);
    const int64_t qb = idx >> 3;
    const int b = (int)(qb % NB);
    const int64_t q = qb / NB;
    uint32_t t1 = 0, t2 = 0, t3 = 0, v = 0;
    if (q < n) {
        const int64_t src = src_row ? src_row[q] : q;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int site = (b * 2 + x) * 32 + 4 * i + j;
            if (site >= L) break;
            const uint32_t c = raw[src * (int64_t)L + site];
            if (!(c == 'A' || c == 'C' || c == 'G' || c == 'T')) continue;  // a gap, or a byte beyond ACGT- (as k_pack_rows<2>: a gap here)
            const uint32_t code = (c >> 1) & 3u;  // as k_pack_rows<2>
            v |= 0x2u << (4 * i);
            t1 |= (0x2u | ((code & 2u) << 2)) << (4 * i);
            t2 |= (0x2u | ((code & 1u) << 3)) << (4 * i);
            t3 |= (0x2u | ((((code >> 1) ^ code) & 1u) << 3)) << (4 * i);
        }
    }
    if (compact) {
        const int64_t ar = row0 + q, tile = ar >> 8;
        const int rr = (int)(ar & 255), sw = (rr >> 2) & 3;
        uint32_t *o = out + ((tile * NB + b) * 1024 + rr * 4) * 4 + j;
        o[((0 + x) ^ sw) * 4] = t1; o[((2 + x) ^ sw) * 4] = t2;
        return;
    }
    uint32_t *o = out + qb * 32 + x * 4 + j;
    o[0] = t1; o[8] = t2; o[16] = t3; o[24] = v;
}

__device__ __forceinline__ uint32_t comp4(const uint4 &v, int x) { return x == 0 ? v.x : (x == 1 ? v.y : (x == 2 ? v.z : v.w)); }

#define MF_TPB 512
#define MF_QT 256  // queries per workgroup tile
template <int MODE>
__global__ __launch_bounds__(MF_TPB, 2) void k_jc69_mfma(const uint4 *__restrict__ refp, const uint8_t *__restrict__ qf4,
                                                          double *__restrict__ dist, uint32_t *__restrict__ counts,
                                                          int64_t n_slots, int64_t slots_pad, int G, int64_t nq,
                                                          int L, double overlap, const double *__restrict__ lut,
                                                          double thr, int32_t *__restrict__ seg_slot,
                                                          int32_t *__restrict__ seg_cnt, const int32_t *__restrict__ mmax) {
    // two generations of the tile images: the next 64-site block is expanded while this one is multiplied
    __shared__ __attribute__((aligned(16))) uint8_t Aq[2][MF_QT * MF_RS];  // queries of the tile, one block
    __shared__ __attribute__((aligned(16))) uint8_t Br[2][128 * MF_RS];    // reference slots of the tile, one block
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wq = wv >> 1, wr = wv & 1;
    const int64_t r0 = (int64_t)blockIdx.x * 128, q0 = (int64_t)blockIdx.y * MF_QT;
    const int NB = 2 * G;  // 64-site blocks: two per group of four plane words
    // loader roles.  Expansion: every thread takes 16 sites of one reference row (16 rows per
    // wavefront; sixteen neighbouring lanes = 8 rows x 2 quarters hit 32 distinct banks at the
    // 144-byte row stride).  Query copy: 64 of a query block's 128 bytes.
    const int lrow = wv * 16 + (lane & 7) + 8 * ((lane >> 5) & 1), lquarter = ((lane >> 3) & 1) + 2 * ((lane >> 4) & 1);
    const int lq = tid >> 1, lhalf = tid & 1;
    const uint8_t *qsrc = qf4 + ((q0 + lq) * (int64_t)NB) * 128 + lhalf * 64;
    const uint4 *rsrc = refp + r0 + lrow;
    v16f_t s1[2][2], s2[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int x = 0; x < 16; ++x) { s1[i][j][x] = 0.f; s2[i][j][x] = 0.f; }
    // plane words of the group being expanded (pm, p0, p1) and of the one after it (pmN, ...)
    uint4 pm, p0, p1, pmN, p0N, p1N, qa, qb, qc, qd;
    auto fetch_planes = [&](int g, uint4 &m, uint4 &c0, uint4 &c1) {
        const uint4 *rp = rsrc + ((int64_t)(g < G ? g : G - 1) * 3) * slots_pad;
        m = rp[0]; c0 = rp[slots_pad]; c1 = rp[2 * slots_pad];
    };
    auto fetch_query = [&](int b) {  // the query image has NB blocks; what is staged beyond them is never multiplied
        const uint4 *qs = reinterpret_cast<const uint4 *>(qsrc + (int64_t)(b < NB ? b : NB - 1) * 128);
        qa = qs[0]; qb = qs[1]; qc = qs[2]; qd = qs[3];
    };
    auto stage_queries = [&](int b) {  // registers -> query image of generation b & 1
        uint4 *ad = reinterpret_cast<uint4 *>(Aq[b & 1] + lq * MF_RS + lhalf * 64);
        ad[0] = qa; ad[1] = qb; ad[2] = qc; ad[3] = qd;
    };
    auto stage_refs = [&](int b, int y) {  // planes -> reference image of generation b & 1 (y = b & 1: words 2y, 2y+1 of the group)
        const bool second = lquarter >> 1;
        const uint32_t m = second ? comp4(pm, 2 * y + 1) : comp4(pm, 2 * y);
        const uint32_t c0 = second ? comp4(p0, 2 * y + 1) : comp4(p0, 2 * y);
        const uint32_t c1 = second ? comp4(p1, 2 * y + 1) : comp4(p1, 2 * y);
        expand_quarter(m, c0, c1, Br[b & 1] + lrow * MF_RS, lquarter);
    };
    const int fr = lane & 31, fh = lane >> 5;
    // operand fragments: components {t1, t2} and {t3, v} live in two register sets that are refilled
    // half a block ahead of their use, so no MFMA ever waits for LDS after the barrier
    v4i_t a01[2][2], b01[2][2], a23[2][2], b23[2][2];
    auto load_frags = [&](int b, int cbase, v4i_t (&fa)[2][2], v4i_t (&fb)[2][2]) {
        const uint8_t *A = Aq[b & 1], *B = Br[b & 1];
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                fa[c][i] = *reinterpret_cast<const v4i_t *>(A + (wq * 64 + i * 32 + fr) * MF_RS + (cbase + c) * 32 + fh * 16);
                fb[c][i] = *reinterpret_cast<const v4i_t *>(B + (wr * 64 + i * 32 + fr) * MF_RS + (cbase + c) * 32 + fh * 16);
            }
    };
    // Schedule of block b (one barrier per block, in the middle of its MFMA stream):
    //   first half : MFMAs t1, t2 of b | query image of b+1 (stores first) | fragments t3, v of b | loads of query block b+2
    //   barrier    : image b+1 complete, image b read by everybody
    //   second half: MFMAs t3, v of b  | fragments t1, t2 of b+1 (first) | reference image of b+2 (into b's buffer)
    // LDS completes in order behind one counter: what the barrier needs (the stores of the first half)
    // goes out first, what the next half-block needs (fragments) next, the expansion's stores last.
    // Block b+2 lies in the next group: its planes were fetched a whole group earlier.
    fetch_planes(0, pm, p0, p1);
    fetch_query(0);
    stage_queries(0);
    stage_refs(0, 0);
    stage_refs(1, 1);
    fetch_planes(1, pm, p0, p1);
    fetch_query(1);
    __syncthreads();
    load_frags(0, 0, a01, b01);
#ifdef MF_NO_FRAGS
    load_frags(0, 2, a23, b23);
#endif
    for (int g = 0; g < G; ++g) {
#pragma unroll
        for (int y = 0; y < 2; ++y) {
            const int b = g * 2 + y;
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) s1[i][j] = mfma_f4(a01[c][i], b01[c][j], s1[i][j]);
#ifndef MF_NO_QSTAGE
            stage_queries(b + 1);
#endif
#ifndef MF_NO_FRAGS
            load_frags(b, 2, a23, b23);
#endif
#ifndef MF_NO_QFETCH
            fetch_query(b + 2);
#endif
            if (y == 0) fetch_planes(g + 2, pmN, p0N, p1N);  // expanded during the next group
#pragma unroll
            for (int k = 0; k < 2; ++k) {  // the four wide stores
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x200, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {  // the eight fragment reads, the loads
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
            }
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) s1[i][j] = mfma_f4(a23[0][i], b23[0][j], s1[i][j]);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) s2[i][j] = mfma_f4(a23[1][i], b23[1][j], s2[i][j]);
#ifndef MF_NO_FRAGS
            load_frags(b + 1, 0, a01, b01);
#endif
#ifndef MF_NO_RSTAGE
            stage_refs(b + 2, y);
#endif
            if (y == 1) { pm = pmN; p0 = p0N; p1 = p1N; }
#pragma unroll
            for (int k = 0; k < 4; ++k) {  // the eight fragment reads first, expansion arithmetic beside them
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {  // then the rest of the expansion and its four stores
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
                __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
            }
        }
    }
    // C layout of the 32x32 tiles: column (reference slot) = lane & 31, row (query) = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
    if (MODE == 0) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int x = 0; x < 16; ++x) {
                    const int64_t q = q0 + wq * 64 + i * 32 + (x & 3) + 8 * (x >> 2) + 4 * fh;
                    const int64_t slot = r0 + wr * 64 + j * 32 + fr;
                    if (q < nq && slot < n_slots) {
                        const uint32_t valid = (uint32_t)(int)s2[i][j][x];
                        const uint32_t mism = (uint32_t)((3 * (int)s2[i][j][x] - (int)s1[i][j][x]) >> 2);
                        const int64_t o = q * slots_pad + slot;
                        if (dist) dist[o] = jc69_from_counts(mism, valid, L, overlap, lut);
                        if (counts) counts[o] = (mism << 16) | valid;
                    }
                }
        return;
    }
#ifdef MF_SKIP_EPILOGUE
    {   // timing experiment: main loop only (every accumulator stays live)
        int acc = 0;
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int x = 0; x < 16; ++x) acc += (int)s1[i][j][x] ^ (int)s2[i][j][x];
        if (acc == 0x7fffffff) dist[0] = 1.0;
        return;
    }
#endif
    // MODE 1: threshold test + per-segment compaction, the format k_select_fast reads.  This wavefront's
    // 64 reference slots are one segment: column tile j = its lower or upper half.  The test is the
    // integer one, mism <= mmax[valid], carried out on the accumulators as they are: 3 valid - sum t.t
    // = 4 mism exactly, against 4 mmax[valid] as a float table in the now idle tile memory (-4 where
    // no count passes: valid = 0, too little overlap).  A survivor is one 32-bit word in seg_slot:
    // position in the segment (6 bits) | valid (13) | mism (13); k_select_fast looks the distance up.
    // No scattered table reads and one store per survivor here.
    __syncthreads();
    float *mm_lds = reinterpret_cast<float *>(&Aq[0][0]);
    for (int i = tid; i <= L; i += MF_TPB) mm_lds[i] = (float)(4 * mmax[i]);
    __syncthreads();
    const int64_t seg = (r0 + wr * 64) >> 6;
    const int64_t n_seg = slots_pad >> 6;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        uint32_t keepbits = 0;
#pragma unroll
        for (int x = 0; x < 16; ++x)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const float valid = s2[i][j][x];
                const float mism4 = __builtin_fmaf(valid, 3.f, -s1[i][j][x]);
                keepbits |= (mism4 <= mm_lds[(int)valid] ? 1u : 0u) << (x * 2 + j);
            }
        if (__ballot(keepbits != 0) == 0ull) continue;  // no survivor among these 32 queries x 64 slots
        const int64_t qbase = q0 + wq * 64 + i * 32 + 4 * fh;  // this lane half's first query of the 32
        // (rows past this launch's queries may be real queries of the next sub-batch: not ours to write)
        const int rem = (int)(nq - qbase < 32 ? nq - qbase : 32);
        int32_t *row0 = seg_slot + qbase * slots_pad + seg * 64;
        int32_t *cnt0 = seg_cnt + qbase * n_seg + seg;
        const uint32_t below = (1u << fr) - 1u;
#pragma unroll
        for (int x = 0; x < 16; ++x) {
            const int cx = (x & 3) + 8 * (x >> 2);  // this register's query, relative to qbase
            const bool in = cx < rem;
            const bool k0 = ((keepbits >> (x * 2)) & 1u) && in, k1 = ((keepbits >> (x * 2 + 1)) & 1u) && in;
            const unsigned long long b0 = __ballot(k0), b1 = __ballot(k1);
            if ((b0 | b1) == 0) continue;  // nobody in either half's segment survives: the counts stay at their preset zero
            // this lane half's query: slots 0..31 of the segment from tile 0, 32..63 from tile 1
            const uint32_t lo = (uint32_t)(b0 >> (32 * fh)), hi = (uint32_t)(b1 >> (32 * fh));
            int32_t *row = row0 + (int64_t)cx * slots_pad;
            if (k0) {
                const int valid = (int)s2[i][0][x], mism = (3 * valid - (int)s1[i][0][x]) >> 2;
                row[__popc(lo & below)] = (int32_t)(((uint32_t)fr << 26) | ((uint32_t)valid << 13) | (uint32_t)mism);
            }
            if (k1) {
                const int valid = (int)s2[i][1][x], mism = (3 * valid - (int)s1[i][1][x]) >> 2;
                row[__popc(lo) + __popc(hi & below)] = (int32_t)(((uint32_t)(32 + fr) << 26) | ((uint32_t)valid << 13) | (uint32_t)mism);
            }
            if (fr == 0 && in) cnt0[(int64_t)cx * n_seg] = __popc(lo) + __popc(hi);
        }
    }
}

template <int P, int MODE>
static void launch_jc69_tile(apples_ctx *ctx, const QueryBlock &qb, int64_t q0, int64_t nq, int tile, double *d_dist,
                             uint32_t *d_counts, int32_t *seg_slot, int32_t *seg_cnt, const int32_t *qlist,
                             const int32_t *qcount, double *segmin_d = nullptr, int32_t *segmin_i = nullptr) {
    const DevAlign &a = ctx->aln;
    // (P = 8 beside 2-plane forms: the 8-plane copies of a context with bytes beyond ACGT-, exact8_rows)
    const uint4 *refp = (P == 8 && a.planes == 2) ? a.packed8 : a.packed;
    const uint4 *qp = ((P == 8 && qb.planes == 2) ? qb.packed8 : qb.packed) + q0 * a.G * (P + 1);  // q0 is a multiple of 32: whole 16-query tiles
    const double *lut = ctx->jc_lut;
    dim3 block(APPLES_TPB);
    const bool use_asm = !knob_on(ctx, "APPLES_NO_BCNT_ASM");  // tuning knob (default: accumulate form)
    const bool no_mmax = knob_on(ctx, "APPLES_NO_MMAX");
    const int32_t *mmax = no_mmax ? nullptr : (ctx->jc_mmax_true ? ctx->jc_mmax_true : ctx->jc_mmax);  // (the rule itself: these kernels count exactly)
#define LAUNCH(TQ)                                                                                                   \
    if (use_asm) LAUNCH2(TQ, true); else LAUNCH2(TQ, false)
#define LAUNCH2(TQ, A)                                                                                               \
    hipLaunchKernelGGL((k_jc69<P, TQ, MODE, A>), dim3((unsigned)(a.slots_pad / APPLES_TPB),                          \
                       (unsigned)(MODE == 2 ? std::min<int64_t>((nq + TQ - 1) / TQ, 64) : (nq + TQ - 1) / TQ)),         \
                       block, 0, ctx->stream, refp, qp, d_dist, d_counts, a.n_rows, a.slots_pad, a.G, nq, a.L,   \
                       ctx->params.overlap_frac, lut, ctx->params.filt_threshold, seg_slot, seg_cnt, qlist, qcount, mmax,    \
                       a.slot_rep, segmin_d, segmin_i)
    if (tile >= 32) LAUNCH(32);
    else if (tile >= 16) LAUNCH(16);
    else if (tile >= 8) LAUNCH(8);
    else if (tile >= 4) LAUNCH(4);
    else LAUNCH(1);
#undef LAUNCH
#undef LAUNCH2
}

// the fused pass runs on the matrix cores and leaves packed (position, valid, mism) words instead of
// slots and distances
bool fused_counts_format(const apples_ctx *ctx, const QueryBlock &qb) {
    const bool no_mmax = knob_on(ctx, "APPLES_NO_MMAX");
    return qb.qf4 && !qb.exact8 && ctx->aln.planes == 2 && ctx->jc_lut && ctx->jc_mmax && !no_mmax && ctx->aln.L < 8192;  // 13-bit counts
}

// the bit-plane kernels read the 8-plane forms for this block: the context's own layout, or the copies a context with bytes
// beyond ACGT- keeps beside its 2-plane rows (exact full rows; a block whose queries carry such bytes: its fused pass too)
bool exact8_rows(const apples_ctx *ctx, const QueryBlock &qb) {
    return ctx->aln.planes == 8 || (ctx->aln.packed8 != nullptr && qb.packed8 != nullptr);
}

bool dist_mfma_enabled(const apples_ctx *ctx) {
    const bool off = knob_on(ctx, "APPLES_NO_DIST_MFMA");  // diagnostic knob: bit-plane VALU kernel everywhere
    return !off;
}

int launch_expand_queries_f4(apples_ctx *ctx, const uint8_t *d_raw, int64_t n, uint8_t *d_out, int64_t n_pad,
                             hipStream_t st, const int32_t *d_src_row, int64_t row0) {
    const DevAlign &a = ctx->aln;
    if (!st) st = ctx->stream;
    const int64_t total = n_pad * a.G * 2 * 8;
    hipLaunchKernelGGL(k_expand_queries_f4, dim3((unsigned)((total + APPLES_TPB - 1) / APPLES_TPB)), dim3(APPLES_TPB), 0,
                       st, d_raw, n, a.L, a.G * 2, reinterpret_cast<uint32_t *>(d_out), n_pad, d_src_row, a.ref_f4 != nullptr ? 1 : 0, row0);
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}

template <int MODE>
static int launch_mfma(apples_ctx *ctx, const QueryBlock &qb, int64_t q0, int64_t nq, double *d_dist, uint32_t *d_counts,
                       int32_t *seg_slot, int32_t *seg_cnt) {
    const DevAlign &a = ctx->aln;
    const bool no_mmax = knob_on(ctx, "APPLES_NO_MMAX");
    dim3 grid((unsigned)(a.slots_pad / 128), (unsigned)((nq + MF_QT - 1) / MF_QT));
    hipLaunchKernelGGL((k_jc69_mfma<MODE>), grid, dim3(MF_TPB), 0, ctx->stream, a.packed,
                       qb.qf4 + q0 * (int64_t)a.G * 256, d_dist, d_counts, a.n_rows, a.slots_pad, a.G, nq, a.L,
                       ctx->params.overlap_frac, ctx->jc_lut, ctx->params.filt_threshold, seg_slot, seg_cnt,
                       no_mmax ? nullptr : ctx->jc_mmax);
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}

// ---- clustered references: panels for the fused selection by representatives (select.hip k_select_clusters)
__global__ __launch_bounds__(APPLES_TPB) void k_gather_reps(const uint4 *__restrict__ packed, int64_t slots_pad,
                                                            const int32_t *__restrict__ rep_slot, int64_t n_reps,
                                                            int64_t reps_pad, uint4 *__restrict__ out) {
    const int64_t j = (int64_t)blockIdx.x * APPLES_TPB + threadIdx.x;
    const int64_t gp = blockIdx.y;  // (word group, plane)
    if (j < n_reps) out[gp * reps_pad + j] = packed[gp * slots_pad + rep_slot[j]];
}

// member rows cluster by cluster, a cluster's members interleaved per plane word: element (gp, position in the cluster) of
// cluster c at (rep_moff[c] * GP + gp * size(c) + position): the lanes of a wavefront that hold consecutive members of one
// cluster read consecutive 16-byte words (k_select_clusters)
__global__ __launch_bounds__(APPLES_TPB) void k_cluster_major(const uint4 *__restrict__ packed, int64_t slots_pad, int64_t n_mem,
                                                              const int32_t *__restrict__ mem_slot, const int32_t *__restrict__ slot_rep,
                                                              const int32_t *__restrict__ rep_moff, int GP, uint4 *__restrict__ out) {
    const int64_t m = (int64_t)blockIdx.x * APPLES_TPB + threadIdx.x;
    const int gp = blockIdx.y;
    if (m >= n_mem) return;
    const int slot = mem_slot[m];
    const int c = slot_rep[slot];
    const int64_t b = rep_moff[c], sz = rep_moff[c + 1] - b;
    out[b * GP + (int64_t)gp * sz + (m - b)] = packed[(int64_t)gp * slots_pad + slot];
}

// representative rows in representative order (the matrix-core pass runs on them alone) and every member row
// once more cluster by cluster (k_cluster_major); ACGT- contexts (2 code planes) only
int launch_build_cluster_panels(apples_ctx *ctx) {
    DevAlign &a = ctx->aln;
    const int GP = a.G * 3;
    a.reps_pad = (a.n_reps + 255) / 256 * 256;
    if (a.rep_packed) (void)hipFree(a.rep_packed);
    if (a.packed_rm) (void)hipFree(a.packed_rm);
    a.rep_packed = a.packed_rm = nullptr;
    HIP_TRY(ctx, hipMalloc((void **)&a.rep_packed, (size_t)GP * a.reps_pad * sizeof(uint4)));
    HIP_TRY(ctx, hipMemsetAsync(a.rep_packed, 0, (size_t)GP * a.reps_pad * sizeof(uint4), ctx->stream));
    HIP_TRY(ctx, hipMalloc((void **)&a.packed_rm, (size_t)GP * a.slots_pad * sizeof(uint4)));
    hipLaunchKernelGGL(k_gather_reps, dim3((unsigned)((a.n_reps + APPLES_TPB - 1) / APPLES_TPB), (unsigned)GP), dim3(APPLES_TPB), 0,
                       ctx->stream, a.packed, a.slots_pad, a.rep_slot, a.n_reps, a.reps_pad, a.rep_packed);
    hipLaunchKernelGGL(k_cluster_major, dim3((unsigned)((a.n_refs + APPLES_TPB - 1) / APPLES_TPB), (unsigned)GP), dim3(APPLES_TPB), 0,
                       ctx->stream, a.packed, a.slots_pad, a.n_refs, a.mem_slot, a.slot_rep, a.rep_moff, GP, a.packed_rm);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

// the fused matrix-core pass over the representatives alone: survivors (0 <= d <= threshold) per 64-representative
// segment, in the packed (position, valid, mism) format, rows of reps_pad entries
int launch_counts_reps(apples_ctx *ctx, const QueryBlock &qb, int64_t q0, int64_t nq, int32_t *seg_slot, int32_t *seg_cnt) {
    if (nq == 0) return 0;
    const DevAlign &a = ctx->aln;
    dim3 grid((unsigned)(a.reps_pad / 128), (unsigned)((nq + MF_QT - 1) / MF_QT));
    hipLaunchKernelGGL((k_jc69_mfma<1>), grid, dim3(MF_TPB), 0, ctx->stream, a.rep_packed,
                       qb.qf4 + q0 * (int64_t)a.G * 256, (double *)nullptr, (uint32_t *)nullptr, a.n_reps, a.reps_pad, a.G, nq,
                       a.L, ctx->params.overlap_frac, ctx->jc_lut, ctx->params.filt_threshold, seg_slot, seg_cnt, ctx->jc_mmax);
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}

int launch_counts(apples_ctx *ctx, const QueryBlock &qb, int64_t q0, int64_t nq, int tile, double *d_dist,
                  uint32_t *d_counts) {
    if (nq == 0) return 0;
    // Full rows stay with the bit-plane kernel: with 8 bytes out per pair and a table lookup per pair the
    // matrix-core form is slower here (0.64 against 0.58 ms at C2 size).  APPLES_DIST_MFMA_ROWS=1 routes
    // tiles of 16 and more queries to it anyway (tests compare its counts with the bytewise definition).
    if (qb.qf4 && !ctx->aln.ref_f4 && ctx->aln.planes == 2 && !ctx->aln.packed8 && tile >= 16 && knob_on(ctx, "APPLES_DIST_MFMA_ROWS"))
        return launch_mfma<0>(ctx, qb, q0, nq, d_dist, d_counts, nullptr, nullptr);
    if (!exact8_rows(ctx, qb)) launch_jc69_tile<2, 0>(ctx, qb, q0, nq, tile, d_dist, d_counts, nullptr, nullptr, nullptr, nullptr);
    else launch_jc69_tile<8, 0>(ctx, qb, q0, nq, tile, d_dist, d_counts, nullptr, nullptr, nullptr, nullptr);
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}

// fused threshold compaction (MODE 1): see k_select_fast for the consumer
int launch_counts_fused(apples_ctx *ctx, const QueryBlock &qb, int64_t q0, int64_t nq, int tile, double *seg_d,
                        int32_t *seg_slot, int32_t *seg_cnt) {
    if (nq == 0) return 0;
    if (fused_counts_format(ctx, qb))
        return dist_gemm_usable(ctx) ? launch_counts_gemm(ctx, qb, q0, nq, seg_slot, seg_cnt)
                                     : launch_mfma<1>(ctx, qb, q0, nq, seg_d, nullptr, seg_slot, seg_cnt);
    if (!exact8_rows(ctx, qb)) launch_jc69_tile<2, 1>(ctx, qb, q0, nq, tile, seg_d, nullptr, seg_slot, seg_cnt, nullptr, nullptr);
    else launch_jc69_tile<8, 1>(ctx, qb, q0, nq, tile, seg_d, nullptr, seg_slot, seg_cnt, nullptr, nullptr);
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}

// full rows for a device-side list of queries of the block starting at q0 (MODE 2); the grid covers
// nq_max list entries and tiles beyond *qcount exit at once
int launch_counts_listed(apples_ctx *ctx, const QueryBlock &qb, int64_t q0, int64_t nq_max, const int32_t *qlist,
                         const int32_t *qcount, double *d_dist, double *segmin_d, int32_t *segmin_i) {
    if (nq_max == 0) return 0;
    if (!exact8_rows(ctx, qb)) launch_jc69_tile<2, 2>(ctx, qb, q0, nq_max, 8, d_dist, nullptr, nullptr, nullptr, qlist, qcount, segmin_d, segmin_i);
    else launch_jc69_tile<8, 2>(ctx, qb, q0, nq_max, 8, d_dist, nullptr, nullptr, nullptr, qlist, qcount, segmin_d, segmin_i);
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}

// scoredist (apples/distance.py:681-715): valid as above; tot = sum over sites of T[q_s][r_s] with
// T = BLOSUM45 dissimilarities extended by a zero gap row/column; r = 1 - tot/valid; r <= 0 -> -1;
// else -ln(r)*1.3.  The 21x21 fp64 table sits in LDS; for one query site all lanes index the same
// 21-entry row with their own residue, i.e. 21 consecutive 8-byte words: conflict-free ds_read_b64.
// Sites are summed left to right in fp64 (the reference's order is whatever BLAS ddot does, so
// this path is tolerance-checked, not bit-checked).
// Workgroup = 256 reference slots x TQ queries; blockIdx.x walks the query tiles, blockIdx.y the slot blocks, so that the
// workgroups in flight together share one 140 KB slice of the reference (L2) and stream the queries.
// MODE 0: full rows for queries q0.. ; MODE 2: full rows for the queries named by qlist[0..*qcount) (top-up path), row r
// of dist belongs to qlist[r]; MODE 1: fused threshold compaction per 64-slot segment (the format k_select_fast reads, as
// k_jc69 MODE 1), with an exact early exit: the table is >= 0, so the partial sums only grow, and d <= thr <=> tot <=
// c valid with c = 1 - exp(-thr / 1.3).  valid comes first (gap masks only); a query of the tile whose 64 partial sums
// all exceed c valid (1 + 1e-9) is dropped from the wavefront's site loop (a wave-uniform mask), which ends when none is
// left.  (The default fused route is dist_sd.hip's: this one serves wide thresholds, clustered references' representatives
// excepted, and APPLES_NO_SD_GEMM.)
template <int TQ, int MODE>
__global__ __launch_bounds__(APPLES_TPB) void k_scoredist(const uint8_t *__restrict__ refa, const uint16_t *__restrict__ refm,
                                                          const uint8_t *__restrict__ qa, const uint16_t *__restrict__ qm,
                                                          const double *__restrict__ table, double *__restrict__ dist,
                                                          uint32_t *__restrict__ counts, int64_t n_slots,
                                                          int64_t slots_pad, int Lpad, int L, int64_t nq, double overlap,
                                                          double thr, double cfrac, int32_t *__restrict__ seg_slot,
                                                          int32_t *__restrict__ seg_cnt, const int32_t *__restrict__ qlist,
                                                          const int32_t *__restrict__ qcount) {
    __shared__ double T[21 * 21];
    for (int i = threadIdx.x; i < 21 * 21; i += APPLES_TPB) T[i] = table[i];
    __syncthreads();
    const char *Tb = reinterpret_cast<const char *>(T);
    const int64_t slot = (int64_t)blockIdx.y * APPLES_TPB + threadIdx.x;
    const int n16 = Lpad / 16;
    if (MODE == 2) nq = *qcount;
    // listed mode: the list length lives on the device, so a bounded grid.x loops over the tiles
    for (int64_t q0 = (int64_t)blockIdx.x * TQ; q0 < nq; q0 += (MODE == 2 ? (int64_t)gridDim.x * TQ : nq)) {
    int64_t qi[TQ];
#pragma unroll
    for (int t = 0; t < TQ; ++t) qi[t] = (MODE == 2) ? (int64_t)qlist[(q0 + t < nq) ? q0 + t : q0] : q0 + t;
    double tot[TQ];
    uint32_t nv[TQ];
#pragma unroll
    for (int t = 0; t < TQ; ++t) { tot[t] = 0.0; nv[t] = 0; }
    uint32_t alive = (1u << TQ) - 1u;  // wave-uniform
    double cut[TQ];
    if (MODE == 1) {
        for (int s16 = 0; s16 < n16; ++s16) {
            const uint32_t rmask = refm[(int64_t)s16 * slots_pad + slot];
#pragma unroll
            for (int t = 0; t < TQ; ++t) nv[t] += __popc(rmask & (uint32_t)qm[qi[t] * (int64_t)n16 + s16]);
        }
#pragma unroll
        for (int t = 0; t < TQ; ++t) {
            const bool ok = slot < n_slots && nv[t] != 0 && !((double)nv[t] / (double)L < overlap);
            cut[t] = ok ? (double)nv[t] * cfrac : -1.0;
            if (q0 + t >= nq || __ballot(ok) == 0ull) alive &= ~(1u << t);
        }
    }
    for (int s16 = 0; s16 < n16 && alive; ++s16) {
        // this row's 16 residues as table column byte offsets (index * 8), unpacked once per block of
        // sites and reused for all TQ queries
        const uint4 rw = *reinterpret_cast<const uint4 *>(refa + ((int64_t)s16 * slots_pad + slot) * 16);
        uint32_t r8[16];
        const uint32_t rr[4] = {rw.x, rw.y, rw.z, rw.w};
#pragma unroll
        for (int k = 0; k < 16; ++k) r8[k] = (rr[k >> 2] >> (8 * (k & 3))) & 0xffu;
        if (MODE != 1) {
            const uint32_t rmask = refm[(int64_t)s16 * slots_pad + slot];
            // wave-uniform: the queries' residues select table rows (scalar arithmetic)
            uint32_t qv[TQ][4];
#pragma unroll
            for (int t = 0; t < TQ; ++t) {
                const uint4 qw = *reinterpret_cast<const uint4 *>(qa + qi[t] * (int64_t)Lpad + s16 * 16);
                qv[t][0] = qw.x; qv[t][1] = qw.y; qv[t][2] = qw.z; qv[t][3] = qw.w;
                nv[t] += __popc(rmask & (uint32_t)qm[qi[t] * (int64_t)n16 + s16]);
            }
            // site-major: the TQ table reads of one site are independent, so they are in flight together;
            // each query still sums its sites left to right
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                double v[TQ];
#pragma unroll
                for (int t = 0; t < TQ; ++t) {
                    const uint32_t qrow = ((qv[t][k >> 2] >> (8 * (k & 3))) & 0xffu) * 168u;  // 21 columns * 8 bytes
                    v[t] = *reinterpret_cast<const double *>(Tb + qrow + r8[k]);               // gap row/column = +0.0
                }
#pragma unroll
                for (int t = 0; t < TQ; ++t) tot[t] += v[t];
            }
        } else {
            // query-major over the queries still alive: a query's 16 lookups are independent of one another
#pragma unroll
            for (int t = 0; t < TQ; ++t) {
                if (!((alive >> t) & 1u)) continue;  // (wave-uniform)
                const uint4 qw = *reinterpret_cast<const uint4 *>(qa + qi[t] * (int64_t)Lpad + s16 * 16);
                const uint32_t qq[4] = {qw.x, qw.y, qw.z, qw.w};
                double v[16];
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    const uint32_t qrow = ((qq[k >> 2] >> (8 * (k & 3))) & 0xffu) * 168u;
                    v[k] = *reinterpret_cast<const double *>(Tb + qrow + r8[k]);
                }
#pragma unroll
                for (int k = 0; k < 16; ++k) tot[t] += v[k];
                if (__ballot(tot[t] <= cut[t]) == 0ull) alive &= ~(1u << t);
            }
        }
    }
    const int lane = threadIdx.x & 63;
    const int64_t seg = slot >> 6, n_seg = slots_pad >> 6;
#pragma unroll
    for (int t = 0; t < TQ; ++t) {
        if (q0 + t < nq) {  // wave-uniform
            const int64_t o = (q0 + t) * slots_pad + slot;
            const uint32_t valid = nv[t];
            double d;
            if (valid == 0 || (double)valid / (double)L < overlap) d = -1.0;
            else {
                double r1 = 1 - tot[t] / (double)valid;
                if (0 >= r1) d = -1.0;
                else d = -log_libm(r1) * 1.3;
            }
            if (MODE == 1) {
                // (a query dropped from the site loop has no pair left that can pass; the pairs of the others whose sums
                // stopped growing when they exceeded their cut fail the test below on the partial sum already)
                const bool keep = ((alive >> t) & 1u) && slot < n_slots && tot[t] <= cut[t] && d >= 0 && d <= thr;
                const unsigned long long m = __ballot(keep);
                if (keep) {
                    const int64_t oo = (q0 + t) * slots_pad + seg * 64 + __popcll(m & ((1ull << lane) - 1ull));
                    seg_slot[oo] = (int32_t)slot;
                    dist[oo] = d;
                }
                if (lane == 0) seg_cnt[(q0 + t) * n_seg + seg] = __popcll(m);
            } else if (slot < n_slots) {
                if (dist) dist[o] = d;
                if (counts) counts[o] = valid;
            }
        }
    }
    }
}

static double sd_cut_fraction(double thr) { return (1.0 - std::exp(-thr / 1.3)) * (1.0 + 1e-9); }

int launch_scoredist(apples_ctx *ctx, const QueryBlock &qb, int64_t q0, int64_t nq, double *d_dist,
                     uint32_t *d_counts) {
    if (nq == 0) return 0;
    const DevAlign &a = ctx->aln;
    int Lpad = (a.L + 15) / 16 * 16;
    constexpr int TQ = 8;
    hipLaunchKernelGGL((k_scoredist<TQ, 0>), dim3((unsigned)((nq + TQ - 1) / TQ), (unsigned)(a.slots_pad / APPLES_TPB)),
                       dim3(APPLES_TPB), 0, ctx->stream, a.aa_idx, a.aa_mask, qb.aa_idx + q0 * Lpad,
                       qb.aa_mask + q0 * (Lpad / 16), ctx->blosum, d_dist, d_counts, a.n_rows, a.slots_pad, Lpad, a.L, nq,
                       ctx->params.overlap_frac, 0.0, 0.0, (int32_t *)nullptr, (int32_t *)nullptr, (const int32_t *)nullptr,
                       (const int32_t *)nullptr);
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}

// fused threshold compaction (MODE 1: seg_d/seg_slot/seg_cnt as launch_counts_fused leaves them for k_select_fast)
int launch_scoredist_fused(apples_ctx *ctx, const QueryBlock &qb, int64_t q0, int64_t nq, double *seg_d, int32_t *seg_slot,
                           int32_t *seg_cnt, double *) {
    if (nq == 0) return 0;
    const DevAlign &a = ctx->aln;
    int Lpad = (a.L + 15) / 16 * 16;
    constexpr int TQ = 8;
    hipLaunchKernelGGL((k_scoredist<TQ, 1>), dim3((unsigned)((nq + TQ - 1) / TQ), (unsigned)(a.slots_pad / APPLES_TPB)),
                       dim3(APPLES_TPB), 0, ctx->stream, a.aa_idx, a.aa_mask, qb.aa_idx + q0 * Lpad,
                       qb.aa_mask + q0 * (Lpad / 16), ctx->blosum, seg_d, (uint32_t *)nullptr, a.n_rows, a.slots_pad, Lpad, a.L, nq,
                       ctx->params.overlap_frac, ctx->params.filt_threshold, sd_cut_fraction(ctx->params.filt_threshold), seg_slot,
                       seg_cnt, (const int32_t *)nullptr, (const int32_t *)nullptr);
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}

// ---- clustered references, scoredist (the default route of `-p`): the distances to the representatives alone, then the
// survivors in the form k_select_clusters reads
__global__ __launch_bounds__(APPLES_TPB) void k_gather_reps_aa(const uint8_t *__restrict__ idx, const uint16_t *__restrict__ mask,
                                                               int64_t slots_pad, const int32_t *__restrict__ rep_slot,
                                                               int64_t n_reps, int64_t reps_pad, uint8_t *__restrict__ oidx,
                                                               uint16_t *__restrict__ omask) {
    const int64_t j = (int64_t)blockIdx.x * APPLES_TPB + threadIdx.x;
    const int64_t s16 = blockIdx.y;
    if (j >= n_reps) return;
    const int64_t slot = rep_slot[j];
    reinterpret_cast<uint4 *>(oidx)[s16 * reps_pad + j] = reinterpret_cast<const uint4 *>(idx)[s16 * slots_pad + slot];
    omask[s16 * reps_pad + j] = mask[s16 * slots_pad + slot];
}

int launch_build_cluster_panels_aa(apples_ctx *ctx) {
    DevAlign &a = ctx->aln;
    const int64_t n16 = (a.L + 15) / 16;
    a.reps_pad = (a.n_reps + 255) / 256 * 256;
    if (a.aa_rep_idx) (void)hipFree(a.aa_rep_idx);
    if (a.aa_rep_mask) (void)hipFree(a.aa_rep_mask);
    a.aa_rep_idx = nullptr; a.aa_rep_mask = nullptr;
    HIP_TRY(ctx, hipMalloc((void **)&a.aa_rep_idx, (size_t)n16 * a.reps_pad * 16));
    HIP_TRY(ctx, hipMemsetAsync(a.aa_rep_idx, 160, (size_t)n16 * a.reps_pad * 16, ctx->stream));  // (padding rows: gaps)
    HIP_TRY(ctx, hipMalloc((void **)&a.aa_rep_mask, (size_t)n16 * a.reps_pad * 2));
    HIP_TRY(ctx, hipMemsetAsync(a.aa_rep_mask, 0, (size_t)n16 * a.reps_pad * 2, ctx->stream));
    hipLaunchKernelGGL(k_gather_reps_aa, dim3((unsigned)((a.n_reps + APPLES_TPB - 1) / APPLES_TPB), (unsigned)n16), dim3(APPLES_TPB), 0,
                       ctx->stream, a.aa_idx, a.aa_mask, a.slots_pad, a.rep_slot, a.n_reps, a.reps_pad, a.aa_rep_idx, a.aa_rep_mask);
    // ... and every member row once more cluster by cluster (mem_slot's order): a wavefront of k_cluster_dist_sd holds consecutive
    // members of one cluster in its lanes, and from the slot-ordered rows every lane's 16 bytes were a sector of their own
    if (a.aa_cm_idx) (void)hipFree(a.aa_cm_idx);
    if (a.aa_cm_mask) (void)hipFree(a.aa_cm_mask);
    a.aa_cm_idx = nullptr; a.aa_cm_mask = nullptr;
    HIP_TRY(ctx, hipMalloc((void **)&a.aa_cm_idx, (size_t)n16 * a.slots_pad * 16));
    HIP_TRY(ctx, hipMemsetAsync(a.aa_cm_idx, 160, (size_t)n16 * a.slots_pad * 16, ctx->stream));
    HIP_TRY(ctx, hipMalloc((void **)&a.aa_cm_mask, (size_t)n16 * a.slots_pad * 2));
    HIP_TRY(ctx, hipMemsetAsync(a.aa_cm_mask, 0, (size_t)n16 * a.slots_pad * 2, ctx->stream));
    hipLaunchKernelGGL(k_gather_reps_aa, dim3((unsigned)((a.n_refs + APPLES_TPB - 1) / APPLES_TPB), (unsigned)n16), dim3(APPLES_TPB), 0,
                       ctx->stream, a.aa_idx, a.aa_mask, a.slots_pad, a.mem_slot, a.n_refs, a.slots_pad, a.aa_cm_idx, a.aa_cm_mask);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

// the representatives with 0 <= d <= thr of every query, per 64-representative segment: position << 26 (the packed word of the
// JC69 passes without its counts: k_select_clusters takes the distance from the row) and the segment's count
__global__ __launch_bounds__(APPLES_TPB) void k_sd_rep_survivors(const double *__restrict__ rep_d, int64_t n_reps, int64_t reps_pad,
                                                                 double thr, int32_t *__restrict__ seg_slot,
                                                                 int32_t *__restrict__ seg_cnt) {
    const int64_t q = blockIdx.x;
    const int64_t j = (int64_t)blockIdx.y * APPLES_TPB + threadIdx.x;  // (the grid covers reps_pad)
    const int lane = threadIdx.x & 63;
    const double d = j < n_reps ? rep_d[q * reps_pad + j] : -1.0;
    const bool keep = d >= 0 && d <= thr;
    const unsigned long long m = __ballot(keep);
    const int64_t seg = j >> 6;
    if (keep) seg_slot[q * reps_pad + seg * 64 + __popcll(m & ((1ull << lane) - 1ull))] = (int32_t)((uint32_t)lane << 26);
    if (lane == 0) seg_cnt[q * (reps_pad >> 6) + seg] = __popcll(m);
}

int launch_scoredist_reps(apples_ctx *ctx, const QueryBlock &qb, int64_t q0, int64_t nq, double *rep_d, int32_t *seg_slot,
                          int32_t *seg_cnt) {
    if (nq == 0) return 0;
    const DevAlign &a = ctx->aln;
    int Lpad = (a.L + 15) / 16 * 16;
    constexpr int TQ = 8;
    hipLaunchKernelGGL((k_scoredist<TQ, 0>), dim3((unsigned)((nq + TQ - 1) / TQ), (unsigned)(a.reps_pad / APPLES_TPB)),
                       dim3(APPLES_TPB), 0, ctx->stream, a.aa_rep_idx, a.aa_rep_mask, qb.aa_idx + q0 * Lpad,
                       qb.aa_mask + q0 * (Lpad / 16), ctx->blosum, rep_d, (uint32_t *)nullptr, a.n_reps, a.reps_pad, Lpad, a.L, nq,
                       ctx->params.overlap_frac, 0.0, 0.0, (int32_t *)nullptr, (int32_t *)nullptr, (const int32_t *)nullptr,
                       (const int32_t *)nullptr);
    hipLaunchKernelGGL(k_sd_rep_survivors, dim3((unsigned)nq, (unsigned)(a.reps_pad / APPLES_TPB)), dim3(APPLES_TPB), 0, ctx->stream,
                       rep_d, a.n_reps, a.reps_pad, ctx->params.filt_threshold, seg_slot, seg_cnt);
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}

// full rows for a device-side list of queries of the block starting at q0 (MODE 2): row r of d_dist = query qlist[r]; the grid
// covers at most nq_max list entries and tiles beyond *qcount exit at once
int launch_scoredist_listed(apples_ctx *ctx, const QueryBlock &qb, int64_t q0, int64_t nq_max, const int32_t *qlist,
                            const int32_t *qcount, double *d_dist) {
    if (nq_max == 0) return 0;
    const DevAlign &a = ctx->aln;
    int Lpad = (a.L + 15) / 16 * 16;
    constexpr int TQ = 8;
    const int64_t tiles = std::min<int64_t>((nq_max + TQ - 1) / TQ, 4096);
    hipLaunchKernelGGL((k_scoredist<TQ, 2>), dim3((unsigned)tiles, (unsigned)(a.slots_pad / APPLES_TPB)),
                       dim3(APPLES_TPB), 0, ctx->stream, a.aa_idx, a.aa_mask, qb.aa_idx + q0 * Lpad,
                       qb.aa_mask + q0 * (Lpad / 16), ctx->blosum, d_dist, (uint32_t *)nullptr, a.n_rows, a.slots_pad, Lpad, a.L,
                       nq_max, ctx->params.overlap_frac, 0.0, 0.0, (int32_t *)nullptr, (int32_t *)nullptr, qlist, qcount);
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}

// ---- bytes beyond ACGT- in the reference rows of a matrix-core context (DevAlign::ex_off) ------------------------------------
// The fused pass counted them as gaps.  For a pair (query of letters and gaps, row) every site where the row holds such a byte and
// the query a letter is a valid site and a mismatch (apples/distance.py:733-737: the bytes differ), so the pair's counts are
// (valid + k, mism + k) with k = the row's such sites under the query's letters -- which can only make the pair fail: every true
// survivor is among the pass's survivors (its table, ctx->jc_mmax, is loosened by ex_max valid sites at its lower end, where the
// overlap rule -V could let a pair in that the smaller count kept out: set_params).  A workgroup per query walks its survivors,
// recounts the ones on such rows from the query's raw bytes, tests 0 <= d <= threshold on the integers (the rule's own table) and
// rewrites the packed word -- valid 0 for a pair that fails: the table says -1 there and k_select_fast (SelectArgs::seg_surv) drops it.
namespace {
#ifndef WAVE
#define WAVE 64
#endif
__global__ __launch_bounds__(APPLES_TPB) void k_exotic_fix(int32_t *__restrict__ seg_slot, const int32_t *__restrict__ seg_cnt, int64_t stride,
                                                          const uint8_t *__restrict__ qraw, int L, const int32_t *__restrict__ ex_off,
                                                          const uint16_t *__restrict__ ex_site, const int32_t *__restrict__ mmax,
                                                          int retest_all, int32_t *__restrict__ n_surv) {
    __shared__ int sh_cnt[APPLES_TPB / WAVE];
    const int64_t q = blockIdx.x;
    const int64_t n_seg = stride >> 6;
    const uint8_t *row = qraw + q * (int64_t)L;
    int kept = 0;
    for (int64_t s = threadIdx.x; s < n_seg; s += APPLES_TPB) {
        const int c = seg_cnt[q * n_seg + s];
        int32_t *w = seg_slot + q * stride + s * 64;
        for (int k = 0; k < c; ++k) {
            const uint32_t pk = (uint32_t)w[k];
            const int64_t slot = s * 64 + (pk >> 26);
            int valid = (int)((pk >> 13) & 0x1fffu), mism = (int)(pk & 0x1fffu);
            const int e0 = ex_off ? ex_off[slot] : 0, e1 = ex_off ? ex_off[slot + 1] : 0;
            if (e1 > e0) {
                int add = 0;
                for (int e = e0; e < e1; ++e) {
                    const uint8_t b = row[ex_site[e]];
                    add += (b == 'A' || b == 'C' || b == 'G' || b == 'T') ? 1 : 0;
                }
                valid += add; mism += add;
            } else if (!retest_all) { ++kept; continue; }
            if (mism <= mmax[valid]) { ++kept; w[k] = (int32_t)((pk & 0xfc000000u) | ((uint32_t)valid << 13) | (uint32_t)mism); }
            else w[k] = (int32_t)(pk & 0xfc000000u);
        }
    }
    for (int o = WAVE / 2; o > 0; o >>= 1) kept += __shfl_down(kept, o, WAVE);
    if ((threadIdx.x & (WAVE - 1)) == 0) sh_cnt[threadIdx.x / WAVE] = kept;
    __syncthreads();
    if (threadIdx.x == 0) {
        int t = 0;
        for (int i = 0; i < APPLES_TPB / WAVE; ++i) t += sh_cnt[i];
        n_surv[q] = t;
    }
}
}  // namespace

int launch_exotic_fix(apples_ctx *ctx, const QueryBlock &qb, int64_t q0, int64_t nq, int32_t *seg_slot, const int32_t *seg_cnt,
                      int32_t *n_surv) {
    if (nq == 0) return 0;
    const DevAlign &a = ctx->aln;
    hipLaunchKernelGGL(k_exotic_fix, dim3((unsigned)nq), dim3(APPLES_TPB), 0, ctx->stream, seg_slot, seg_cnt, a.slots_pad,
                       qb.raw + q0 * (int64_t)a.L, a.L, a.ex_off, a.ex_site, ctx->jc_mmax_true ? ctx->jc_mmax_true : ctx->jc_mmax,
                       ctx->jc_mmax_true ? 1 : 0, n_surv);
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}

// ---- diagnostic: the kernels' logarithm on an array (include/apples_hip.h: apples_device_log) ------------------------------
namespace {
__global__ void k_log_array(const double *__restrict__ x, double *__restrict__ out, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = log_libm(x[i]);
}
}  // namespace

extern "C" int apples_device_log(int device, const double *x, int64_t n, double *out) {
    if (n <= 0) return 0;
    double *d_x = nullptr, *d_o = nullptr;
    auto fail = [&](const char *what, hipError_t e) {
        g_create_error = std::string("apples_device_log: ") + what + ": " + hipGetErrorString(e);
        if (d_x) (void)hipFree(d_x);
        if (d_o) (void)hipFree(d_o);
        return 1;
    };
    hipError_t e;
    if ((e = hipSetDevice(device)) != hipSuccess) return fail("hipSetDevice", e);
    if ((e = hipMalloc(&d_x, (size_t)n * 8)) != hipSuccess) return fail("hipMalloc", e);
    if ((e = hipMalloc(&d_o, (size_t)n * 8)) != hipSuccess) return fail("hipMalloc", e);
    if ((e = hipMemcpy(d_x, x, (size_t)n * 8, hipMemcpyHostToDevice)) != hipSuccess) return fail("hipMemcpy", e);
    hipLaunchKernelGGL(k_log_array, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, nullptr, d_x, d_o, n);
    if ((e = hipGetLastError()) != hipSuccess) return fail("launch", e);
    if ((e = hipMemcpy(out, d_o, (size_t)n * 8, hipMemcpyDeviceToHost)) != hipSuccess) return fail("hipMemcpy", e);
    (void)hipFree(d_x);
    (void)hipFree(d_o);
    return 0;
}
