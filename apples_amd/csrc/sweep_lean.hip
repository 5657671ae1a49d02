// Least-squares placement sweep, lean form for big binary trees (the C3 route): the level loop of sweep.hip with
// the merged level lists, cut into three passes per query and with no tree records in it.
//
//   apples/Subtree.py:23-43 (validate_edges)        -> pass A: the level lists, by merging (structure only, integers)
//   apples/OLS.py:12-44 ... (all_S_values)          -> pass B: S tuples, deepest level first
//   apples/OLS.py:46-128 ..., util.py:6-54          -> pass C: R tuples, 2x2 solve, residual, running arg-min, top down
//   apples/Algorithm.py:62-101 (placement)          -> wavefront arg-min, the placement struct
//
// What sweep.hip's level step gathers per internal node and pass -- a 64-byte tree record, 25.6 MB of them at 200 k
// leaves, out of the Infinity Cache at best -- is replaced by one 16-byte gather per swept node in pass A: {parent,
// edge length} of every list key (`pe`, 6.4 MB).  A list entry then carries everything the later passes need of
// its (at most two) valid children: descriptors, node ids and edge lengths.  Passes B and C read their levels
// front to back out of arrays (one array per field: every load of a level is a contiguous run per wavefront), and a
// child's tuple from the level below, whose entries are in the same order as their parents: near-sequential too.
// A node's R tuple is stored already lifted over its own edge (the parent knows that length), so a node never needs
// its own tree constants.  Levels of at most 64 nodes hand their tuples to the next level through LDS.
//
// Team = one wavefront per query (four per workgroup, no s_barrier).  Queries with many observed leaves are routed to
// sweep.hip's workgroup-sized teams as before; so are trees with polytomies, the HYBRID criterion and per-edge
// inspection (the launcher decides).  Arithmetic: sweep_math.h, shared with sweep.hip -- same expressions in the same
// order, so placements are bit-identical to the level loop's.
#include <algorithm>
#include <cstdlib>

#include "common.h"

#include "sweep_math.h"

namespace {

struct LeanShared {
    double pow[384 + 256];                    // libm pow tables (sweep_math.h)
    double2 stage[APPLES_TPB / WAVE][3][WAVE];  // per wavefront: the tuples of a level of at most 64 nodes
    int mk[APPLES_TPB / WAVE][2][WAVE];       // the two key windows of a merge step
    int w[APPLES_TPB / WAVE];
};

// per-team scratch: one array per field, `cap1` entries each (cap1 a multiple of 4)
struct LeanTeam {
    int32_t *K;     // node id
    int2 *D;        // descriptors of the first and second valid child (> 0: entry index + 1, <= -2: observed leaf -(j+2), 0: none)
    int2 *N;        // their node ids
    double2 *E;     // their edge lengths
    double2 *T0, *T1, *T2;  // the node's tuple: S after pass B, lift(R) after pass C reached its parent
};

__device__ __forceinline__ LeanTeam lean_team(void *base, int64_t team, int64_t cap1) {
    char *p = reinterpret_cast<char *>(base) + team * cap1 * LEAN_BYTES_PER_NODE;
    LeanTeam t;
    t.T0 = reinterpret_cast<double2 *>(p); p += cap1 * 16;
    t.T1 = reinterpret_cast<double2 *>(p); p += cap1 * 16;
    t.T2 = reinterpret_cast<double2 *>(p); p += cap1 * 16;
    t.E = reinterpret_cast<double2 *>(p); p += cap1 * 16;
    t.D = reinterpret_cast<int2 *>(p); p += cap1 * 8;
    t.N = reinterpret_cast<int2 *>(p); p += cap1 * 8;
    t.K = reinterpret_cast<int32_t *>(p);
    return t;
}

__device__ __forceinline__ double shfl_down_f64(double v, int delta) {
    return __hiloint2double(__shfl_down(__double2hiint(v), delta, WAVE), __shfl_down(__double2loint(v), delta, WAVE));
}

// Pass A, one level: merge by node id the parents of this level's internal nodes (K[base .. base + nA), sorted) and of
// its observed leaves (o_node[lo .. lo + nB), sorted): the next level's list, sorted (sweep.hip:merge_parents -- the
// same merge-path step of 64 keys through two LDS windows; a binary tree's runs have at most two keys and a step
// whose last key opens a run leaves it to the next step).  The entry of a parent names its valid children and carries
// their node ids and edge lengths.  Returns the number of entries written from next_base on.
__device__ __forceinline__ int lean_merge(const LeanTeam &t, int base, int nA, const int32_t *__restrict__ o_node, int lo, int nB,
                                          int next_base, const int4 *__restrict__ pe, int *mk_a, int *mk_b, int lane) {
    int out = 0, ia = 0, ib = 0;
    const unsigned long long below = (1ull << lane) - 1ull;
    while (ia < nA || ib < nB) {
        const int rem = (nA - ia) + (nB - ib);
        const int wa = min(nA - ia, WAVE), wb = min(nB - ib, WAVE);
        mk_a[lane] = lane < wa ? t.K[base + ia + lane] : 0x7fffffff;
        mk_b[lane] = lane < wb ? o_node[lo + ib + lane] : 0x7fffffff;
        __builtin_amdgcn_wave_barrier();
        const int tot = min(wa + wb, WAVE);
        const bool active = lane < tot;
        int i_lo = max(0, lane - wb), i_hi = min(lane, wa);  // i = keys of the first window among the lane smallest
        while (i_lo < i_hi) {
            const int i = (i_lo + i_hi) >> 1;
            if (mk_a[i] < mk_b[lane - 1 - i]) i_lo = i + 1; else i_hi = i;
        }
        const int i = i_lo, j = lane - i_lo;
        const int ka = i < wa ? mk_a[i] : 0x7fffffff, kb = j < wb ? mk_b[j] : 0x7fffffff;
        const bool from_a = ka < kb;
        const int key = from_a ? ka : kb;
        const int desc = from_a ? base + ia + i + 1 : -(lo + ib + j) - 2;
        int par = -3;
        double e = 0;
        if (active) {
            const int4 r = pe[key];
            par = r.x;
            e = __hiloint2double(r.w, r.z);
        }
        const int prev = __shfl_up(par, 1, WAVE);
        const bool first = active && (lane == 0 || par != prev);
        const int last_first = __shfl(first ? 1 : 0, tot - 1, WAVE);
        const int use = (rem > tot && last_first && tot > 1) ? tot - 1 : tot;
        const int next_desc = __shfl_down(desc, 1, WAVE), next_key = __shfl_down(key, 1, WAVE);
        const int next_first = __shfl_down(first ? 1 : 0, 1, WAVE);
        const double next_e = shfl_down_f64(e, 1);
        const bool mine = first && lane < use;
        const bool two = lane + 1 < use && !next_first;
        const unsigned long long fm = __ballot(mine);
        if (mine) {
            const int at = next_base + out + __popcll(fm & below);
            t.K[at] = par;
            t.D[at] = make_int2(desc, two ? next_desc : 0);
            t.N[at] = make_int2(key, two ? next_key : -1);
            t.E[at] = make_double2(e, two ? next_e : 0.0);
        }
        const int ca = __popcll(__ballot(lane < use && from_a));
        ia += ca;
        ib += use - ca;
        out += __popcll(fm);
        __builtin_amdgcn_wave_barrier();
    }
    return out;
}

// a child's S tuple: an internal child's from the arrays (or from the LDS stage when its level is there), a leaf's
// rebuilt from its distance
template <int M>
__device__ __forceinline__ void kid_tuple(int kd, const LeanTeam &t, const double2 (*stage)[WAVE], bool staged, int stage_base,
                                          const double *__restrict__ o_dist, double *S) {
    if (kd > 0) {
        double2 a, b, c;
        if (staged) {
            const int p = kd - 1 - stage_base;
            a = stage[0][p]; b = stage[1][p]; c = stage[2][p];
        } else {
            a = t.T0[kd - 1]; b = t.T1[kd - 1]; c = t.T2[kd - 1];
        }
        S[0] = a.x; S[1] = a.y; S[2] = b.x; S[3] = b.y; S[4] = c.x; S[5] = c.y;
    } else {
        leaf_tuple<M>(o_dist[-kd - 2], S);
    }
}

template <int M>
__device__ void lean_team_loop(const SweepArgs &a, LeanShared &sh) {
    constexpr bool BME = (M == APPLES_BME);
    const int lane = threadIdx.x & (WAVE - 1);
    const int wave = threadIdx.x / WAVE;
    const double *lds_pow = sh.pow;
    double2 (*stage)[WAVE] = sh.stage[wave];
    int *mk_a = sh.mk[wave][0], *mk_b = sh.mk[wave][1];
    const DevTree &T = a.tree;
    const int4 *__restrict__ pe = reinterpret_cast<const int4 *>(T.pe);
    const int64_t cap = a.cap;
    const int64_t team = (int64_t)blockIdx.x * (APPLES_TPB / WAVE) + wave;
    const LeanTeam t = lean_team(a.lean, team, a.lean_cap1);
    int32_t *grp_off = a.grp_off + team * (T.height + 4);
    const int c0 = a.cls_count[0], c1 = a.cls_count[1], c2 = a.cls_count[2], c3 = a.cls_count[3];
    const int64_t n_work = (int64_t)c0 + c1 + c2 + c3;
    while (true) {
        // dynamic scheduling: one atomic add per query, broadcast to the wavefront
        int wq = 0;
        if (lane == 0) wq = atomicAdd(a.cursor, 1);
        const int64_t w = __shfl(wq, 0, WAVE);
        if (w >= n_work) break;
        int64_t q;
        if (w < c0) q = a.cls_list[w];
        else if (w < c0 + c1) q = a.cls_list[a.cls_stride + (w - c0)];
        else if (w < (int64_t)c0 + c1 + c2) q = a.cls_list[2 * a.cls_stride + (w - c0 - c1)];
        else q = a.cls_list[3 * a.cls_stride + (w - c0 - c1 - c2)];
        const int n = a.n_obs[q];
        if (n == 0) continue;
        const int32_t *o_node = a.obs_node + q * a.obs_cap;
        const double *o_dist = a.obs_dist + q * a.obs_cap;
        const int32_t *cg = a.cnt_gt + q * (int64_t)(T.height + 2);

        // ------------------------------------------------------------ pass A: the level lists (Subtree.py:23-43)
        const int lvl_first = T.level[o_node[0]];
        int lvl = lvl_first, base = 0, n_par = 0, G = 0, lca = -1;
        bool overflow = false;
        while (true) {
            const int lo = cg[lvl + 1], hi = cg[lvl];  // observed leaves of this level: obs[lo, hi)
            const int n_leaf = hi - lo;
            if (n_par + n_leaf == 1 && hi == n) {  // one node left in the frontier: the LCA (Subtree.py:36-43)
                lca = n_par == 1 ? t.K[base] : o_node[lo];
                break;
            }
            if ((int64_t)base + 2 * (int64_t)n_par + n_leaf > cap) { overflow = true; break; }
            if (lane == 0) grp_off[G] = base;
            const int next_base = base + n_par;
            const int merged = lean_merge(t, base, n_par, o_node, lo, n_leaf, next_base, pe, mk_a, mk_b, lane);
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            base = next_base;
            n_par = merged;
            ++G;
            --lvl;
        }
        if (overflow) {  // hand the query to the workgroup-sized teams with full-size scratch
            if (lane == 0) a.overflow_list[atomicAdd(a.overflow_count, 1)] = (int32_t)q;
            continue;
        }
        const int VI = base;     // internal valid nodes; the LCA's entry sits at index VI
        const int V = base + n;  // Subtree.num_nodes
        if (lane == 0) { grp_off[G] = VI; grp_off[G + 1] = VI + 1; }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        if (a.debug_phase == 1) continue;

        // ------------------------------------------------------------ pass B: S tuples, deepest level first (OLS.py:25-44)
        // (group g = the internal nodes g levels above the deepest observed leaf; group 0 is empty, group G the LCA,
        // whose own S nobody needs)
        bool prev_staged = false;
        int g_lo = grp_off[1];
        for (int g = 1; g < G; ++g) {
            const int g_hi = grp_off[g + 1];
            const int kid_base = grp_off[g - 1];
            const bool one_pass = g_hi - g_lo <= WAVE;
            const bool staged = prev_staged && one_pass;
            for (int idx = g_lo + lane; idx < g_hi; idx += WAVE) {
                const int2 d = t.D[idx];
                const double2 e = t.E[idx];
                const double coef = BME ? 1.0 / (double)(d.y != 0 ? 2 : 1) : 1.0;  // apples/BME.py:20
                double S[6], r[6], u[6];
                kid_tuple<M>(d.x, t, stage, staged, kid_base, o_dist, S);
                lift<M>(S, e.x, u);
#pragma unroll
                for (int x = 0; x < 6; ++x) r[x] = 0;
#pragma unroll
                for (int x = 0; x < 6; ++x) r[x] += BME ? coef * u[x] : u[x];
                if (d.y != 0) {
                    kid_tuple<M>(d.y, t, stage, staged, kid_base, o_dist, S);
                    lift<M>(S, e.y, u);
#pragma unroll
                    for (int x = 0; x < 6; ++x) r[x] += BME ? coef * u[x] : u[x];
                }
                t.T0[idx] = make_double2(r[0], r[1]);
                t.T1[idx] = make_double2(r[2], r[3]);
                t.T2[idx] = make_double2(r[4], r[5]);
                if (one_pass) {  // (every lane's reads of the stage precede this store in the instruction stream)
                    __builtin_amdgcn_wave_barrier();
                    stage[0][idx - g_lo] = make_double2(r[0], r[1]);
                    stage[1][idx - g_lo] = make_double2(r[2], r[3]);
                    stage[2][idx - g_lo] = make_double2(r[4], r[5]);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            prev_staged = one_pass && g_hi > g_lo;
            g_lo = g_hi;
        }
        if (a.debug_phase == 2) continue;

        // ------------------------------------------------------------ pass C, top down and parent-centric: a node forms
        // R for each valid child (all_R_values), solves it (placement_per_edge) and evaluates its residual
        // (error_per_edge); an internal child's tuple becomes lift(R) over its own edge
        double best_key = INF_D;
        int best_v = 0x7fffffff;
        double best_x1 = 0, best_x2 = 0, best_err = 0, best_e = 0;
        int best_int = 0;
        for (int g = G; g >= 1; --g) {
            const int g0 = grp_off[g], g1 = grp_off[g + 1];
            for (int idx = g0 + lane; idx < g1; idx += WAVE) {
                const bool is_lca = idx == VI;
                const int2 d = t.D[idx], nd = t.N[idx];
                const double2 e = t.E[idx];
                const int nk = d.y != 0 ? 2 : 1;
                // apples/BME.py:36-37: 1 / (nonroot + #valid siblings)
                const double coef = BME ? 1.0 / (double)((is_lca ? 0 : 1) + nk - 1) : 1.0;
                double plift[6];
                if (!is_lca) {  // this node's R, already lifted over its own edge by its parent
                    const double2 p0 = t.T0[idx], p1 = t.T1[idx], p2 = t.T2[idx];
                    plift[0] = p0.x; plift[1] = p0.y; plift[2] = p1.x; plift[3] = p1.y; plift[4] = p2.x; plift[5] = p2.y;
                }
                double Sk[6], Ss[6];  // the child in hand and its sibling
                kid_tuple<M>(d.x, t, stage, false, 0, o_dist, Sk);
                if (nk > 1) kid_tuple<M>(d.y, t, stage, false, 0, o_dist, Ss);
                double ek = e.x, es = e.y;
                int kd = d.x, ks = d.y, kn = nd.x, ksn = nd.y;
                // one child at a time (the two swap roles in between): a rolled loop keeps one 2x2 solve's worth of
                // temporaries live, which is what decides how many wavefronts a SIMD holds
#pragma unroll 1
                for (int z = 0; z < nk; ++z) {
                    double acc[6];
#pragma unroll
                    for (int x = 0; x < 6; ++x) acc[x] = 0;
                    if (nk > 1) {  // the one valid sibling (apples/OLS.py:59-69)
                        double u[6];
                        lift<M>(Ss, es, u);
#pragma unroll
                        for (int x = 0; x < 6; ++x) acc[x] += BME ? coef * u[x] : u[x];
                    }
                    if (!is_lca) {  // parent term last (apples/OLS.py:70-80)
#pragma unroll
                        for (int x = 0; x < 6; ++x) acc[x] += BME ? coef * plift[x] : plift[x];
                    }
                    const Sol r = solve_edge<M>(Sk, acc, ek, a.negative, lds_pow);
                    if (kd > 0) {  // what the child will add for each of its own children: lift(R) over its edge
                        double u[6];
                        lift<M>(acc, ek, u);
                        t.T0[kd - 1] = make_double2(u[0], u[1]);
                        t.T1[kd - 1] = make_double2(u[2], u[3]);
                        t.T2[kd - 1] = make_double2(u[4], u[5]);
                    }
                    const double key = (a.criterion == APPLES_ME) ? r.x1 : r.err;
                    if (key < best_key || (key == best_key && kn < best_v)) {
                        best_key = key; best_v = kn; best_x1 = r.x1; best_x2 = r.x2; best_err = r.err; best_int = r.x1_int; best_e = ek;
                    }
#pragma unroll
                    for (int x = 0; x < 6; ++x) { const double w = Sk[x]; Sk[x] = Ss[x]; Ss[x] = w; }
                    { const double w = ek; ek = es; es = w; }
                    { const int w = kd; kd = ks; ks = w; }
                    { const int w = kn; kn = ksn; ksn = w; }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        }

        // ------------------------------------------------------------ selection (apples/Algorithm.py:74-91)
        const int my_best = best_v;
        team_argmin<WAVE>(best_key, best_v, nullptr, nullptr);
        const int win = best_v;
        if (win == 0x7fffffff) {
            if (lane == 0) {
                apples_placement pl = a.out[q];
                pl.n_valid = V;
                pl.edge = -1;
                pl.flags |= APPLES_F_DEGENERATE | APPLES_F_PENDANT_INT;
                a.out[q] = pl;
            }
        } else if (my_best == win) {
            apples_placement pl = a.out[q];
            pl.n_valid = V;
            pl.edge = win;
            pl.error = best_err;
            pl.distal = best_e - best_x2;
            pl.pendant = best_x1;
            pl.flags = 0;
            if (best_int) pl.flags |= APPLES_F_PENDANT_INT;
            if (best_x1 == 0 && best_err > 0 && (best_x2 == 0 || best_x2 == best_e)) pl.flags |= APPLES_F_MISPLACED;
            a.out[q] = pl;
        }
        (void)lca;
    }
}

template <int M, int W>
__global__ __launch_bounds__(APPLES_TPB, W) void k_sweep_lean(SweepArgs a) {
    __shared__ LeanShared sh;
    for (int i = threadIdx.x; i < 384; i += APPLES_TPB) sh.pow[i] = (&kPowLogTab[0][0])[i];
    for (int i = threadIdx.x; i < 256; i += APPLES_TPB) sh.pow[384 + i] = __longlong_as_double((long long)kExpTab[i]);
    __syncthreads();
    lean_team_loop<M>(a, sh);
}

template <int W>
void launch_lean_w(const SweepArgs &a, dim3 grid, hipStream_t st) {
    const dim3 block(APPLES_TPB);
    switch (a.method) {
        case APPLES_FM: hipLaunchKernelGGL((k_sweep_lean<APPLES_FM, W>), grid, block, 0, st, a); break;
        case APPLES_BME: hipLaunchKernelGGL((k_sweep_lean<APPLES_BME, W>), grid, block, 0, st, a); break;
        case APPLES_BE: hipLaunchKernelGGL((k_sweep_lean<APPLES_BE, W>), grid, block, 0, st, a); break;
        default: hipLaunchKernelGGL((k_sweep_lean<APPLES_OLS, W>), grid, block, 0, st, a); break;
    }
}

}  // namespace

// wavefronts per SIMD the lean sweep is compiled for (registers: 2 -> 175 without spills, 3 -> 168 with a few spilled,
// 4 -> 128); the workspace sizes its team count from it.  APPLES_LEAN_WAVES: tuning knob.
int sweep_lean_waves() {
    static const int w = getenv("APPLES_LEAN_WAVES") ? std::min(4, std::max(2, atoi(getenv("APPLES_LEAN_WAVES")))) : 2;
    return w;
}

// wavefront-sized teams over the size-class queues of one device batch; `a.lean` etc. set by the caller
int launch_sweep_lean(apples_ctx *ctx, const SweepArgs &a, int64_t nq, int wgs, hipStream_t st) {
    if (nq == 0) return 0;
    const int64_t need = (nq + 3) / 4;
    const dim3 grid((unsigned)std::min<int64_t>(need, wgs));
    switch (sweep_lean_waves()) {
        case 4: launch_lean_w<4>(a, grid, st); break;
        case 3: launch_lean_w<3>(a, grid, st); break;
        default: launch_lean_w<2>(a, grid, st); break;
    }
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}
