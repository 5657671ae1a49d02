// Least-squares placement sweep: one persistent workgroup walks queries, one query at a time.
//
// Per query (apples/PoolQueryWorker.py:101-133):
//   Subtree            apples/Subtree.py:23-43   -> bottom-up level loop below (marks + LCA)
//   all_S_values       apples/OLS.py:12-44 ...   -> fused into the same bottom-up loop
//   all_R_values       apples/OLS.py:46-80 ...   -> top-down level loop
//   placement_per_edge apples/OLS.py:83-97 + apples/util.py:6-54   -> fused into the top-down loop
//   error_per_edge     apples/OLS.py:100-128 ... -> fused into the top-down loop
//   placement          apples/Algorithm.py:62-101 -> wavefront/LDS arg-min over the candidates
//   unroll_changes     apples/Subtree.py:72-76   -> map[] entries cleared
//
// The reference pops the deepest frontier node, marks it valid and pushes its parent until one
// node (the LCA) is left.  Level-synchronous form: the nodes at level l are the observed leaves
// of that level (contiguous in the level-sorted obs list) plus the parents claimed from level
// l+1; the loop stops when a level holds a single node and no shallower leaves remain.
//
// Bit parity: fp64, compiled with -ffp-contract=off; every sum is taken in the order of the
// cited source line, children/siblings in file order and the parent term last.
// `x ** 2` is libm pow in the reference and x*x here (<= 1 ulp apart).
// HBM-bound by design: ~60 fp64 flops against ~330 B per node; no MFMA.
#include "common.h"

#define WAVE 64
#define INF_D __longlong_as_double(0x7ff0000000000000LL)

// tuple slots, reference attribute names per method
//   OLS/BME: 0 S   1 Sd    2 Sd2    3 SDd     4 SD2   5 SD      (apples/OLS.py:27-33, BME.py:11-17)
//   FM     : 0 S   1 Sd_D  2 Sd_D2  3 Sd2_D2  4 S1_D  5 S1_D2   (apples/FM.py:21-27)
//   BE     : 0 S   1 Sd    2 Sd_D   3 Sd2_D   4 SD    5 S1_D    (apples/BE.py:11-17)
template <int M>
__device__ __forceinline__ void leaf_tuple(double D, double *t) {
    t[0] = 1; t[1] = 0; t[2] = 0; t[3] = 0;
    if (M == APPLES_OLS || M == APPLES_BME) { t[4] = D * D; t[5] = D; }
    else if (M == APPLES_FM) { t[4] = 1.0 / D; t[5] = 1.0 / (D * D); }
    else { t[4] = D; t[5] = 1.0 / D; }
}

// what a parent adds for a child (or sibling, or its own R) tuple s over the edge e
template <int M>
__device__ __forceinline__ void lift(const double *s, double e, double *t) {
    if (M == APPLES_OLS || M == APPLES_BME) {  // apples/OLS.py:36-44
        t[0] = s[0];
        t[1] = s[0] * e + s[1];
        t[2] = s[0] * e * e + s[2] + 2 * e * s[1];
        t[3] = e * s[5] + s[3];
        t[4] = s[4];
        t[5] = s[5];
    } else if (M == APPLES_FM) {  // apples/FM.py:31-40
        t[0] = s[0];
        t[1] = e * s[4] + s[1];
        t[2] = e * s[5] + s[2];
        t[3] = s[5] * e * e + s[3] + 2 * e * s[2];
        t[4] = s[4];
        t[5] = s[5];
    } else {  // apples/BE.py:20-30
        t[0] = s[0];
        t[1] = s[0] * e + s[1];
        t[2] = e * s[5] + s[2];
        t[3] = s[5] * e * e + s[3] + 2 * e * s[2];
        t[4] = s[4];
        t[5] = s[5];
    }
}

struct Sol {
    double x1, x2, x1n, x2n, err;
    int x1_int;
};

// placement_per_edge + util.solve2_2 + error_per_edge for one edge
template <int M>
__device__ __forceinline__ Sol solve_edge(const double *S, const double *R, double e, int negative) {
    // which tuple slots play which role (apples/OLS.py:90-96, FM.py:86-92, BE.py:61-67, BME.py:64-70)
    constexpr int IA = (M == APPLES_OLS || M == APPLES_BME) ? 0 : 5;                     // a_11 = R? + S?
    constexpr int IC = (M == APPLES_OLS || M == APPLES_BME) ? 5 : (M == APPLES_FM ? 4 : 0);  // RD / R1_D / R
    constexpr int IE = (M == APPLES_OLS || M == APPLES_BME) ? 0 : 5;                     // e * S / S1_D2 / S1_D
    constexpr int ID = (M == APPLES_OLS || M == APPLES_BME) ? 1 : 2;                     // Rd / Rd_D2 / Rd_D
    double a11 = R[IA] + S[IA];
    double a12 = R[IA] - S[IA];
    double a21 = a12, a22 = a11;
    double c1 = R[IC] + S[IC] - e * S[IE] - R[ID] - S[ID];
    double c2 = R[IC] - S[IC] + e * S[IE] - R[ID] + S[ID];
    // apples/util.py:26-50
    double det = 1 / (a11 * a22 - a12 * a21);
    Sol r;
    r.x1n = (a22 * c1 - a12 * c2) * det;
    r.x2n = (-a21 * c1 + a11 * c2) * det;
    r.x1 = r.x1n;
    r.x2 = r.x2n;
    r.x1_int = 0;
    if (!negative) {
        double x1n = r.x1n, x2n = r.x2n;
        if (x1n < 0 && x2n < 0) {
            r.x1 = 0; r.x1_int = 1;
            r.x2 = 0;
        } else if (x1n > 0 && x2n < 0) {
            double t = c1 * 1.0 / a11;
            if (0 > t) { r.x1 = 0; r.x1_int = 1; } else r.x1 = t;  // max(t, 0)
            r.x2 = 0;
        } else if (x1n < 0 && 0 <= x2n && x2n <= e) {
            r.x1 = 0; r.x1_int = 1;
            double u = c2 * 1.0 / a22;
            if (0 > u) u = 0;      // max(u, 0)
            r.x2 = (e < u) ? e : u;  // min(., e)
        } else if (x1n < 0 && x2n > e) {
            r.x1 = 0; r.x1_int = 1;
            r.x2 = e;
        } else if (x1n > 0 && x2n > e) {
            double t = (c1 * 1.0 - a12 * e) / a11;
            if (0 > t) { r.x1 = 0; r.x1_int = 1; } else r.x1 = t;
            r.x2 = e;
        }
    }
    // error_per_edge (apples/OLS.py:121-128, FM.py:117-124, BE.py:73-80, BME.py:76-83)
    constexpr int JA = (M == APPLES_FM) ? 0 : 4;
    constexpr int JB = (M == APPLES_OLS || M == APPLES_BME) ? 1 : 2;
    constexpr int JC = (M == APPLES_OLS || M == APPLES_BME) ? 0 : 5;
    constexpr int JD = (M == APPLES_OLS || M == APPLES_BME) ? 5 : (M == APPLES_FM ? 4 : 0);
    constexpr int JE = (M == APPLES_OLS || M == APPLES_BME) ? 3 : 1;
    constexpr int JF = (M == APPLES_OLS || M == APPLES_BME) ? 2 : 3;
    double x1 = r.x1, x2 = r.x2;
    double up = x1 + x2;      // path through the parent side
    double dn = e + x1 - x2;  // path through the child side
    double A = R[JA] + S[JA];
    double B = 2 * up * R[JB] + 2 * dn * S[JB];
    double C = (up * up) * R[JC] + (dn * dn) * S[JC];
    double Dd = -2 * up * R[JD] - 2 * dn * S[JD];
    double E = -2 * R[JE] - 2 * S[JE];
    double F = R[JF] + S[JF];
    r.err = A + B + C + Dd + E + F;
    return r;
}

__device__ __forceinline__ double shfl_down_f64s(double v, int delta) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __shfl_down(lo, delta, WAVE);
    hi = __shfl_down(hi, delta, WAVE);
    return __hiloint2double(hi, lo);
}

// block-wide lexicographic arg-min over (key, id); NaN keys never win
__device__ void block_argmin(double &d, int &i, double *shd, int *shi) {
    for (int o = WAVE / 2; o > 0; o >>= 1) {
        double d2 = shfl_down_f64s(d, o);
        int i2 = __shfl_down(i, o, WAVE);
        if (d2 < d || (d2 == d && i2 < i)) { d = d2; i = i2; }
    }
    int w = threadIdx.x / WAVE;
    __syncthreads();
    if ((threadIdx.x & (WAVE - 1)) == 0) { shd[w] = d; shi[w] = i; }
    __syncthreads();
    d = shd[0]; i = shi[0];
    for (int k = 1; k < APPLES_TPB / WAVE; ++k) {
        double d2 = shd[k]; int i2 = shi[k];
        if (d2 < d || (d2 == d && i2 < i)) { d = d2; i = i2; }
    }
}

template <int M>
__global__ __launch_bounds__(APPLES_TPB) void k_sweep(SweepArgs a, int64_t nq) {
    __shared__ int sh_cnt[3];
    __shared__ double sh_d[4];
    __shared__ int sh_i[4];
    const int tid = threadIdx.x;
    const DevTree &T = a.tree;
    const int64_t nn = T.n_nodes;
    int32_t *map = a.map + (int64_t)blockIdx.x * nn;
    int32_t *order = a.order + (int64_t)blockIdx.x * nn;
    int32_t *grp_off = a.grp_off + (int64_t)blockIdx.x * (T.height + 4);
    double *Sb = a.S + (int64_t)blockIdx.x * nn * 6;
    double *Rb = a.R + (int64_t)blockIdx.x * nn * 6;
    double *xe = a.xe ? a.xe + (int64_t)blockIdx.x * nn * 5 : nullptr;

    for (int64_t q = blockIdx.x; q < nq; q += gridDim.x) {
        const int n = a.n_obs[q];
        if (n == 0) continue;
        const int32_t *o_node = a.obs_node + q * a.obs_cap;
        const double *o_dist = a.obs_dist + q * a.obs_cap;
        const int32_t *cg = a.cnt_gt + q * (int64_t)(T.height + 2);

        // ------------------------------------------------------------ bottom-up: mark + S values
        int lvl = T.level[o_node[0]];
        int base = 0, n_par = 0, G = 0, lca = -1, lca_claimed = 0;
        if (tid < 3) sh_cnt[tid] = 0;
        __syncthreads();
        while (true) {
            const int lo = cg[lvl + 1], hi = cg[lvl];  // observed leaves of this level: obs[lo, hi)
            const int n_lvl = n_par + (hi - lo);
            if (n_lvl == 1 && hi == n) {  // one node left in the frontier: the LCA (Subtree.py:36-43)
                if (n_par == 1) { lca = order[base]; lca_claimed = 1; }
                else lca = o_node[lo];
                break;
            }
            if (tid == 0) { grp_off[G] = base; sh_cnt[(G + 1) % 3] = 0; }
            int *next_cnt = &sh_cnt[G % 3];
            for (int k = tid; k < n_lvl; k += APPLES_TPB) {
                const int idx = base + k;
                double acc[6];
                int v;
                if (k >= n_par) {
                    const int j = lo + (k - n_par);
                    v = o_node[j];
                    order[idx] = v;
                    map[v] = idx + 1;
                    leaf_tuple<M>(o_dist[j], acc);
                } else {
                    v = order[idx];
#pragma unroll
                    for (int c = 0; c < 6; ++c) acc[c] = 0;
                    const int c0 = T.child_off[v], c1 = T.child_off[v + 1];
                    double coef = 1.0;
                    if (M == APPLES_BME) {  // apples/BME.py:20
                        int nv = 0;
                        for (int ci = c0; ci < c1; ++ci) nv += map[T.child_idx[ci]] > 0;
                        coef = 1.0 / (double)nv;
                    }
                    for (int ci = c0; ci < c1; ++ci) {
                        const int c = T.child_idx[ci];
                        const int mc = map[c];
                        if (mc > 0) {
                            double s[6], t[6];
                            const double *sp = Sb + (int64_t)(mc - 1) * 6;
#pragma unroll
                            for (int x = 0; x < 6; ++x) s[x] = sp[x];
                            lift<M>(s, T.edge_len[c], t);
#pragma unroll
                            for (int x = 0; x < 6; ++x) acc[x] += (M == APPLES_BME) ? coef * t[x] : t[x];
                        }
                    }
                }
                double *sp = Sb + (int64_t)idx * 6;
#pragma unroll
                for (int x = 0; x < 6; ++x) sp[x] = acc[x];
                const int p = T.parent[v];
                if (p >= 0 && atomicCAS(&map[p], 0, -1) == 0) {
                    const int nidx = base + n_lvl + atomicAdd(next_cnt, 1);
                    order[nidx] = p;
                    map[p] = nidx + 1;
                }
            }
            __syncthreads();
            n_par = *next_cnt;
            base += n_lvl;
            ++G;
            --lvl;
        }
        const int V = base;  // Subtree.num_nodes
        if (tid == 0) grp_off[G] = V;
        __syncthreads();

        // ------------------------------------------------------------ top-down: R values, solve, residual
        double best_key = INF_D;
        int best_v = 0x7fffffff;
        for (int g = G - 1; g >= 0; --g) {
            const int g0 = grp_off[g], g1 = grp_off[g + 1];
            for (int idx = g0 + tid; idx < g1; idx += APPLES_TPB) {
                const int v = order[idx];
                const int p = T.parent[v];
                double acc[6];
#pragma unroll
                for (int c = 0; c < 6; ++c) acc[c] = 0;
                const int c0 = T.child_off[p], c1 = T.child_off[p + 1];
                double coef = 1.0;
                if (M == APPLES_BME) {  // apples/BME.py:36-37
                    int ns = (p != lca) ? 1 : 0;
                    for (int ci = c0; ci < c1; ++ci) {
                        const int c = T.child_idx[ci];
                        ns += (c != v) && (map[c] > 0);
                    }
                    coef = 1.0 / (double)ns;
                }
                for (int ci = c0; ci < c1; ++ci) {
                    const int c = T.child_idx[ci];
                    if (c == v) continue;
                    const int mc = map[c];
                    if (mc > 0) {
                        double s[6], t[6];
                        const double *sp = Sb + (int64_t)(mc - 1) * 6;
#pragma unroll
                        for (int x = 0; x < 6; ++x) s[x] = sp[x];
                        lift<M>(s, T.edge_len[c], t);
#pragma unroll
                        for (int x = 0; x < 6; ++x) acc[x] += (M == APPLES_BME) ? coef * t[x] : t[x];
                    }
                }
                if (p != lca) {  // parent is valid: add its R lifted over its edge (apples/OLS.py:70-80)
                    double s[6], t[6];
                    const double *rp = Rb + (int64_t)(map[p] - 1) * 6;
#pragma unroll
                    for (int x = 0; x < 6; ++x) s[x] = rp[x];
                    lift<M>(s, T.edge_len[p], t);
#pragma unroll
                    for (int x = 0; x < 6; ++x) acc[x] += (M == APPLES_BME) ? coef * t[x] : t[x];
                }
                double *rp = Rb + (int64_t)idx * 6;
#pragma unroll
                for (int x = 0; x < 6; ++x) rp[x] = acc[x];
                double s[6];
                const double *sp = Sb + (int64_t)idx * 6;
#pragma unroll
                for (int x = 0; x < 6; ++x) s[x] = sp[x];
                Sol r = solve_edge<M>(s, acc, T.edge_len[v], a.negative);
                if (a.keep_edges) {
                    double *xp = xe + (int64_t)idx * 5;
                    xp[0] = r.x1; xp[1] = r.x2; xp[2] = r.x1n; xp[3] = r.x2n; xp[4] = r.err;
                }
                const double key = (a.criterion == APPLES_ME) ? r.x1 : r.err;
                if (key < best_key || (key == best_key && v < best_v)) { best_key = key; best_v = v; }
            }
            __syncthreads();
        }

        // ------------------------------------------------------------ selection (apples/Algorithm.py:74-91)
        int win;
        if (a.criterion == APPLES_HYBRID) {
            // nsmallest(floor(log2(num_nodes))) by error (stable = ties to the smaller edge_index),
            // then the first minimum of x_1 among them in that order
            const int kk = 31 - __clz(V);
            double last_e = -INF_D;
            int last_v = -1;
            double bx = INF_D;
            win = -1;
            for (int r = 0; r < kk; ++r) {
                double ke = INF_D;
                int kv = 0x7fffffff;
                for (int idx = tid; idx < V; idx += APPLES_TPB) {
                    const double e = xe[(int64_t)idx * 5 + 4];
                    const int v = order[idx];
                    const bool after = (e > last_e) || (e == last_e && v > last_v);
                    if (after && (e < ke || (e == ke && v < kv))) { ke = e; kv = v; }
                }
                block_argmin(ke, kv, sh_d, sh_i);
                if (kv == 0x7fffffff) break;
                last_e = ke; last_v = kv;
                const double x1 = xe[(int64_t)(map[kv] - 1) * 5 + 0];
                if (win < 0 || x1 < bx) { bx = x1; win = kv; }
            }
        } else {
            block_argmin(best_key, best_v, sh_d, sh_i);
            win = best_v;
        }

        if (tid == 0) {
            apples_placement pl = a.out[q];
            pl.n_valid = V;
            if (win < 0 || win == 0x7fffffff) {
                pl.edge = -1;
                pl.flags |= APPLES_F_DEGENERATE | APPLES_F_PENDANT_INT;
            } else {
                const int idx = map[win] - 1;
                double s[6], r6[6];
#pragma unroll
                for (int x = 0; x < 6; ++x) { s[x] = Sb[(int64_t)idx * 6 + x]; r6[x] = Rb[(int64_t)idx * 6 + x]; }
                const double e = T.edge_len[win];
                Sol r = solve_edge<M>(s, r6, e, a.negative);
                pl.edge = win;
                pl.error = r.err;
                pl.distal = e - r.x2;
                pl.pendant = r.x1;
                pl.flags = 0;
                if (r.x1_int) pl.flags |= APPLES_F_PENDANT_INT;
                if (r.x1 == 0 && r.err > 0 && (r.x2 == 0 || r.x2 == e)) pl.flags |= APPLES_F_MISPLACED;
            }
            a.out[q] = pl;
            grp_off[T.height + 3] = lca;
        }
        __syncthreads();
        // ------------------------------------------------------------ unroll_changes (Subtree.py:72-76)
        for (int idx = tid; idx < V + lca_claimed; idx += APPLES_TPB) map[order[idx]] = 0;
        __syncthreads();
    }
}

int launch_sweep(apples_ctx *ctx, const SweepArgs &a, int64_t nq, int wgs) {
    if (nq == 0) return 0;
    dim3 grid((unsigned)(nq < wgs ? nq : wgs)), block(APPLES_TPB);
    switch (a.method) {
        case APPLES_FM: hipLaunchKernelGGL(k_sweep<APPLES_FM>, grid, block, 0, ctx->stream, a, nq); break;
        case APPLES_BME: hipLaunchKernelGGL(k_sweep<APPLES_BME>, grid, block, 0, ctx->stream, a, nq); break;
        case APPLES_BE: hipLaunchKernelGGL(k_sweep<APPLES_BE>, grid, block, 0, ctx->stream, a, nq); break;
        default: hipLaunchKernelGGL(k_sweep<APPLES_OLS>, grid, block, 0, ctx->stream, a, nq); break;
    }
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}
