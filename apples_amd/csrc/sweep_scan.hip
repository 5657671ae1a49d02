// Least-squares placement sweep, scan formulation (the default; sweep.hip's level loop over tree records
// remains for trees too deep for the per-leaf ancestor tables).
//
// Per query (apples/PoolQueryWorker.py:101-133), as in sweep.hip:
//   Subtree            apples/Subtree.py:23-43      -> phases 0-1 (which nodes, where)
//   all_S_values       apples/OLS.py:12-44 ...      -> phase 2 (bottom-up)
//   all_R_values       apples/OLS.py:46-80 ...      -> phase 3 (top-down)
//   placement_per_edge apples/OLS.py:83-97, apples/util.py:6-54, error_per_edge apples/OLS.py:100-128 -> phase 3
//   placement          apples/Algorithm.py:62-101   -> team arg-min
//
// The level loop of sweep.hip finds a node's children through a bitmap or a hash-like map and gathers
// their 64-byte records: every level step is a chain of scattered gathers, and the chip's gather rate
// out of L2 (profiles/r02_gather_probe.txt: ~150 G records/s) bounds the kernel.  Here the observed
// leaves arrive SORTED BY NODE ID (= left-to-right post-order number, apples/util.py:57-69) and every
// index comes from ballots and prefix counts over them (tests/sweep_scan_model.py is the executable
// model, checked against the oracle on CPU):
//   lev[i]  level of leaf i;  lca[i]  level of the lowest common ancestor of leaves i-1 and i
//   top     = min lca = level of the subtree's root (the reference's Subtree.root, never a candidate)
//   leaf i owns the ancestors of itself at the levels m with max(lca[i], top) < m <= lev[i]
//   within a level the owned nodes, in leaf order, are sorted by node id: position = prefix count;
//   a node's children are consecutive positions of the level below, in file order;
//   the parent of the node leaf i owns at level m = the node at m-1 owned by the last leaf j <= i that owns one.
// Entries (one per subtree node) live in per-team component arrays S[6], R[6], e, node, parent, first
// child, child count; the bottom-up and top-down passes read and write them sequentially (a wavefront's
// load instruction covers 512 contiguous bytes), the sums run in the reference's order (children and
// siblings in file order, parent term last).  No atomics, no map, no bitmap, no tree pointers: big trees
// and small trees run the same code.
#include <algorithm>
#include <cstdlib>

#include "sweep_math.h"

#define ENT_S 0
#define ENT_R 6
#define ENT_NF 13  // double components per entry: S[6], R[6], edge length
#define ENT_ANC 0  // row of the ancestor table (DevTree::anc); the fill pass replaces it by the node id
#define ENT_NODE 0
#define ENT_PAR 1
#define ENT_KID0 2
#define ENT_NK 3
#define ENT_LEAF 4 // index of the observed leaf this entry is, or -1 (its tuple is rebuilt from the distance)
#define ENT_NI 5   // int components per entry

struct ScanShared {
    double pow[384 + 256];                  // libm pow tables (sweep_math.h)
    double d[4];
    int i[4];
    int w[APPLES_TPB / WAVE];
    int cnt[APPLES_TPB / WAVE][4];          // cross-wavefront prefix scratch
    int lvl_off[APPLES_TPB / WAVE][260];    // per team: first entry of level m (wavefront teams use their row; a workgroup team row 0)
    uint16_t leaves[4 * SCAN_LDS_LEAVES_SMALL];  // per-leaf (level | lca level << 8): 4 x 2048 or 1 x 8192
    int32_t loff[4 * SCAN_LDS_LEAVES_SMALL];     // per-leaf row offset into the ancestor table
};

// team-wide exclusive prefix of two flags; totals in *t0, *t1.  Must be reached by the whole team.
template <int TEAM>
__device__ __forceinline__ void team_prefix2(bool f0, bool f1, int &p0, int &p1, int &t0, int &t1, int (*cnt)[4], int tid) {
    const unsigned long long m0 = __ballot(f0), m1 = __ballot(f1);
    const int lane = tid & (WAVE - 1);
    const unsigned long long below = (1ull << lane) - 1ull;
    p0 = __popcll(m0 & below);
    p1 = __popcll(m1 & below);
    t0 = __popcll(m0);
    t1 = __popcll(m1);
    if (TEAM == WAVE) return;
    const int w = tid / WAVE;
    __syncthreads();
    if (lane == 0) { cnt[w][0] = t0; cnt[w][1] = t1; }
    __syncthreads();
    int b0 = 0, b1 = 0, s0 = 0, s1 = 0;
#pragma unroll
    for (int k = 0; k < APPLES_TPB / WAVE; ++k) {
        if (k < w) { b0 += cnt[k][0]; b1 += cnt[k][1]; }
        s0 += cnt[k][0]; s1 += cnt[k][1];
    }
    p0 += b0; p1 += b1; t0 = s0; t1 = s1;
}

template <int TEAM>
__device__ __forceinline__ int team_min(int v, int *sh, int tid) {
    for (int o = WAVE / 2; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o, WAVE));
    if (TEAM == WAVE) return v;
    __syncthreads();
    if ((tid & (WAVE - 1)) == 0) sh[tid / WAVE] = v;
    __syncthreads();
    int r = sh[0];
#pragma unroll
    for (int k = 1; k < APPLES_TPB / WAVE; ++k) r = min(r, sh[k]);
    return r;
}

template <int TEAM>
__device__ __forceinline__ int team_sum(int v, int *sh, int tid) {
    for (int o = WAVE / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, WAVE);
    if (TEAM == WAVE) return v;
    __syncthreads();
    if ((tid & (WAVE - 1)) == 0) sh[tid / WAVE] = v;
    __syncthreads();
    int r = 0;
#pragma unroll
    for (int k = 0; k < APPLES_TPB / WAVE; ++k) r += sh[k];
    return r;
}

template <int M, int TEAM>
__device__ void scan_team(const ScanArgs &a, int64_t nq, ScanShared &sh) {
    constexpr int TEAMS_PER_WG = APPLES_TPB / TEAM;
    constexpr bool BME = (M == APPLES_BME);
    const int team_in_wg = threadIdx.x / TEAM;
    const int tid = threadIdx.x % TEAM;
    const int64_t team = (int64_t)blockIdx.x * TEAMS_PER_WG + team_in_wg;
    const int64_t cap = a.cap;
    double *ef = a.ent_f + team * ENT_NF * cap;
    int32_t *ei = a.ent_i + team * ENT_NI * cap;
    double *xe = a.xe ? a.xe + team * cap * XE_STRIDE : nullptr;
    int32_t *meta = a.meta + team * 4;
    int *lvl_off = sh.lvl_off[team_in_wg];
    const double *lds_pow = sh.pow;
    // per-leaf state: LDS when the query's leaves fit the team's share, else the team's global area
    uint16_t *lds_leaves = sh.leaves + (TEAM == WAVE ? team_in_wg * SCAN_LDS_LEAVES_SMALL : 0);
    int32_t *lds_loff = sh.loff + (TEAM == WAVE ? team_in_wg * SCAN_LDS_LEAVES_SMALL : 0);
    const int lds_cap = TEAM == WAVE ? SCAN_LDS_LEAVES_SMALL : SCAN_LDS_LEAVES_BIG;
    uint16_t *glb_leaves = a.leaf_g ? a.leaf_g + team * a.leaf_cap * 3 : nullptr;  // 6 bytes per leaf: states, then offsets
    int32_t *glb_loff = a.leaf_g ? reinterpret_cast<int32_t *>(glb_leaves + a.leaf_cap) : nullptr;
    // entry components: tuples as three arrays of double2 (a wavefront's load covers 1 KiB), then the edge length
    double2 *ef2 = reinterpret_cast<double2 *>(ef);
    double *ee = ef + (int64_t)12 * cap;
    auto Fe = [&](int64_t e) -> double & { return ee[e]; };
    auto I = [&](int comp, int64_t e) -> int32_t & { return ei[(int64_t)comp * cap + e]; };
    auto loadT = [&](int comp, int64_t e, double *t) {  // comp = ENT_S or ENT_R
        const double2 p0 = ef2[(int64_t)(comp / 2) * cap + e], p1 = ef2[(int64_t)(comp / 2 + 1) * cap + e], p2 = ef2[(int64_t)(comp / 2 + 2) * cap + e];
        t[0] = p0.x; t[1] = p0.y; t[2] = p1.x; t[3] = p1.y; t[4] = p2.x; t[5] = p2.y;
    };
    auto storeT = [&](int comp, int64_t e, const double *t) {
        ef2[(int64_t)(comp / 2) * cap + e] = make_double2(t[0], t[1]);
        ef2[(int64_t)(comp / 2 + 1) * cap + e] = make_double2(t[2], t[3]);
        ef2[(int64_t)(comp / 2 + 2) * cap + e] = make_double2(t[4], t[5]);
    };

    // work queue (as sweep.hip): size-class lists, a device-side list, or 0..nq-1
    int c0 = 0, c1 = 0, c2 = 0, c3 = 0;
    int64_t n_work;
    if (a.cls_list) {
        c0 = a.cls_count[0]; c1 = a.cls_count[1]; c2 = a.cls_count[2]; c3 = a.cls_count[3];
        n_work = (int64_t)c0 + c1 + c2 + c3;
    } else {
        n_work = a.work_count ? *a.work_count : nq;
    }
    unsigned long long t_prev = a.prof ? wall_clock64() : 0, t_acc[6] = {0, 0, 0, 0, 0, 0};
    auto stamp = [&](int k) {
        if (a.prof) { const unsigned long long t = wall_clock64(); t_acc[k] += t - t_prev; t_prev = t; }
    };
    while (true) {
        if (tid == 0) sh.w[team_in_wg] = atomicAdd(a.cursor, 1);
        team_sync<TEAM>();
        const int64_t w = sh.w[team_in_wg];
        team_sync<TEAM>();
        if (w >= n_work) break;
        int64_t q;
        if (a.cls_list) {
            if (w < c0) q = a.cls_list[w];
            else if (w < c0 + c1) q = a.cls_list[a.cls_stride + (w - c0)];
            else if (w < (int64_t)c0 + c1 + c2) q = a.cls_list[2 * a.cls_stride + (w - c0 - c1)];
            else q = a.cls_list[3 * a.cls_stride + (w - c0 - c1 - c2)];
        } else {
            q = a.work_list ? a.work_list[w] : w;
        }
        const int n = a.n_obs[q];
        if (n == 0) continue;
        if (TEAM == WAVE && !a.work_list && !a.cls_list && n > a.big_threshold) continue;  // listed for a workgroup-sized team
        const int32_t *o_node = a.obs_node + row_start(a.row_off, q, a.obs_cap);
        const double *o_dist = a.obs_dist + row_start(a.row_off, q, a.obs_cap);
        if (n < 2) continue;
        stamp(0);
        // (a wavefront-sized team always has its leaves in LDS: plain ds_read/ds_write; a workgroup-sized team
        // may have to use its global area, through generic pointers)
        uint16_t *lst = (TEAM == WAVE || n <= lds_cap) ? lds_leaves : glb_leaves;
        int32_t *loff = (TEAM == WAVE || n <= lds_cap) ? lds_loff : glb_loff;
        if ((TEAM == WAVE && n > lds_cap) || !lst) {  // no place for this query's leaf state in this launch: hand it on
            if (tid == 0 && a.overflow_list) a.overflow_list[atomicAdd(a.overflow_count, 1)] = (int32_t)q;
            team_sync<TEAM>();
            continue;
        }

        // ---------------------------------------------------------------- phase 0: lev, lca per leaf
        // lca level of leaves i-1 and i = minimum level on the Euler tour between their positions
        int my_min = 0x7fffffff, my_max = 0, my_sum = 0;
        for (int i = tid; i < n; i += TEAM) {
            const int4 li = a.leaf_info[o_node[i]];  // {table offset, level, Euler position}
            int lc = 255;
            if (i > 0) {
                const int l = a.leaf_info[o_node[i - 1]].z, r = li.z;
                const int k = 31 - __clz(r - l + 1);
                const uint8_t *row = a.rmq + (int64_t)k * a.euler_len;
                lc = min((int)row[l], (int)row[r - (1 << k) + 1]);
                my_min = min(my_min, lc);
                my_sum += li.y - lc;
            }
            my_max = max(my_max, li.y);
            lst[i] = (uint16_t)(li.y | (lc << 8));
            loff[i] = li.x;
        }
        const int top = team_min<TEAM>(my_min, sh.i, tid);
        const int Lmax = -team_min<TEAM>(-my_max, sh.i, tid);
        const int lev0 = a.leaf_info[o_node[0]].y;
        const int V = team_sum<TEAM>(my_sum, sh.i, tid) + (lev0 - top);  // Subtree.num_nodes
        if (tid == 0) lst[0] = (uint16_t)(lev0 | (top << 8));
        if ((int64_t)V > cap) {  // does not fit this team's entry arrays: a team with full-size scratch takes it
            if (tid == 0 && a.overflow_list) a.overflow_list[atomicAdd(a.overflow_count, 1)] = (int32_t)q;
            team_sync<TEAM>();
            continue;
        }
        team_sync<TEAM>();

        stamp(1);
        // ---------------------------------------------------------------- phase 1: entries, level by level from the top
        // (index arithmetic only: what an entry needs from the tree -- node id, edge length -- and from the
        // query -- its distance, if it is a leaf -- is fetched by the position-wise passes below, where the
        // loads of a whole level are in flight together)
        {
            int base = 0, base_above = 0;
            for (int m = top + 1; m <= Lmax; ++m) {
                int run = 0, above = 0;
                for (int i0 = 0; i0 < n; i0 += TEAM) {  // team-uniform trip count
                    const int i = i0 + tid;
                    const unsigned st = i < n ? lst[i] : 0xff00u;
                    const int lv = (int)(st & 0xffu), lc = (int)(st >> 8);
                    const bool own = lc < m && m <= lv;
                    const bool up = (m - 1 > top) && lc < m - 1 && m - 1 <= lv;
                    int p_own, p_up, t_own, t_up;
                    team_prefix2<TEAM>(own, up, p_own, p_up, t_own, t_up, sh.cnt, tid);
                    if (own) {
                        const int64_t E = base + run + p_own;
                        I(ENT_ANC, E) = loff[i] + m;
                        // inclusive count of owners at m-1 up to this leaf, minus one
                        I(ENT_PAR, E) = (m - 1 > top) ? base_above + above + p_up + (up ? 1 : 0) - 1 : -1;
                        I(ENT_KID0, E) = -1;
                        I(ENT_NK, E) = 0;
                        I(ENT_LEAF, E) = lv == m ? i : -1;
                    }
                    run += t_own;
                    above += t_up;
                }
                if (tid == 0) lvl_off[m - top] = base;
                base_above = base;
                base += run;
            }
            if (tid == 0) lvl_off[Lmax + 1 - top] = base;  // = V
        }
        team_sync<TEAM>();
        // ---------------------------------------------------------------- fill: what the entries need from outside
        // node id and edge length from the ancestor table, the tuple of a leaf from its distance.  No level
        // order here: every gather of the pass is in flight at once, and the level passes below read plain arrays
        for (int64_t c = tid; c < V; c += TEAM) {
            const AncRec an = a.anc[I(ENT_ANC, c)];
            const int lf = I(ENT_LEAF, c);
            Fe(c) = an.e;
            I(ENT_NODE, c) = an.node;
            if (lf >= 0) {
                double t[6];
                leaf_tuple<M>(o_dist[lf], t);
                storeT(ENT_S, c, t);
            }
        }
        team_sync<TEAM>();

        stamp(2);
        // ---------------------------------------------------------------- phase 2: bottom-up S (all_S_values)
        // position-wise over the children of level m: the first child of every parent sums the run of
        // children that share it, in order, and stores the parent's tuple.  Everything a lane needs sits at
        // c-1 .. c+2: the loads do not depend on each other
        for (int m = Lmax; m >= top + 2; --m) {
            const int k_lo = lvl_off[m - top], k_hi = lvl_off[m + 1 - top];
            for (int c = k_lo + tid; c < k_hi; c += TEAM) {
                const int P = I(ENT_PAR, c);
                const int Pm = c > k_lo ? I(ENT_PAR, c - 1) : -2;
                const int P1 = c + 1 < k_hi ? I(ENT_PAR, c + 1) : -3;
                const int P2 = c + 2 < k_hi ? I(ENT_PAR, c + 2) : -3;
                double s0[6], s1[6];
                loadT(ENT_S, c, s0);
                const double e0 = Fe(c);
                const int cn = c + 1 < k_hi ? c + 1 : c;  // (speculative: the next entry is the second child more often than not)
                loadT(ENT_S, cn, s1);
                const double e1 = Fe(cn);
                if (P != Pm) {
                    int len = 1 + (P1 == P ? 1 : 0);
                    if (P1 == P && P2 == P) {  // polytomy
                        len = 3;
                        while (c + len < k_hi && I(ENT_PAR, c + len) == P) ++len;
                    }
                    const double coef = BME ? 1.0 / (double)len : 1.0;  // apples/BME.py:20
                    double acc[6], t[6];
#pragma unroll
                    for (int x = 0; x < 6; ++x) acc[x] = 0;
                    lift<M>(s0, e0, t);
#pragma unroll
                    for (int x = 0; x < 6; ++x) acc[x] += BME ? coef * t[x] : t[x];
                    if (len >= 2) {
                        lift<M>(s1, e1, t);
#pragma unroll
                        for (int x = 0; x < 6; ++x) acc[x] += BME ? coef * t[x] : t[x];
                    }
                    for (int k = 2; k < len; ++k) {
                        double sk[6];
                        loadT(ENT_S, c + k, sk);
                        lift<M>(sk, Fe(c + k), t);
#pragma unroll
                        for (int x = 0; x < 6; ++x) acc[x] += BME ? coef * t[x] : t[x];
                    }
                    storeT(ENT_S, P, acc);
                    I(ENT_KID0, P) = c;
                    I(ENT_NK, P) = len;
                }
            }
            team_sync<TEAM>();
        }

        stamp(3);
        // ---------------------------------------------------------------- phase 3: top-down R, solve, residual
        double best_key = INF_D;
        int best_v = 0x7fffffff;
        Sol best_sol;
        double best_e = 0;
        best_sol.x1 = best_sol.x2 = best_sol.err = 0; best_sol.x1_int = 0; best_sol.x1n = best_sol.x2n = 0;
        for (int m = top + 1; m <= Lmax; ++m) {
            const int k_lo = lvl_off[m - top], k_hi = lvl_off[m + 1 - top];
            for (int c = k_lo + tid; c < k_hi; c += TEAM) {
                // first round of loads: own entry and both neighbours (a binary node's sibling is one of them)
                const int P = I(ENT_PAR, c);
                const int cm = c > k_lo ? c - 1 : c, cp = c + 1 < k_hi ? c + 1 : c;
                double sv[6], sm[6], sp[6];
                loadT(ENT_S, c, sv);
                loadT(ENT_S, cm, sm);
                loadT(ENT_S, cp, sp);
                const double e = Fe(c), em = Fe(cm), ep = Fe(cp);
                const int node = I(ENT_NODE, c);
                const int my_nk = I(ENT_NK, c);
                // second round: the parent
                int kid0 = k_lo, nk = k_hi - k_lo;  // children of the subtree's root: the whole level
                double rv[6], eP = 0;
                if (P >= 0) {
                    kid0 = I(ENT_KID0, P);
                    nk = I(ENT_NK, P);
                    loadT(ENT_R, P, rv);
                    eP = Fe(P);
                }
                // apples/BME.py:36-37: 1 / (nonroot + #valid siblings)
                const double coef = BME ? 1.0 / (double)((P >= 0 ? 1 : 0) + nk - 1) : 1.0;
                double acc[6], t[6];
#pragma unroll
                for (int x = 0; x < 6; ++x) acc[x] = 0;
                if (nk == 2) {  // the one valid sibling (apples/OLS.py:59-69)
                    if (kid0 == c) lift<M>(sp, ep, t); else lift<M>(sm, em, t);
#pragma unroll
                    for (int x = 0; x < 6; ++x) acc[x] += BME ? coef * t[x] : t[x];
                } else if (nk > 2) {  // polytomy: valid siblings in file order
                    for (int s2 = kid0; s2 < kid0 + nk; ++s2) {
                        if (s2 == c) continue;
                        double sk[6];
                        loadT(ENT_S, s2, sk);
                        lift<M>(sk, Fe(s2), t);
#pragma unroll
                        for (int x = 0; x < 6; ++x) acc[x] += BME ? coef * t[x] : t[x];
                    }
                }
                if (P >= 0) {  // parent term last (apples/OLS.py:70-80)
                    lift<M>(rv, eP, t);
#pragma unroll
                    for (int x = 0; x < 6; ++x) acc[x] += BME ? coef * t[x] : t[x];
                }
                if (my_nk > 0) storeT(ENT_R, c, acc);
                const Sol r = solve_edge<M>(sv, acc, e, a.negative, lds_pow);
                if (a.keep_edges) {
                    double *xp = xe + (int64_t)c * XE_STRIDE;
                    xp[0] = r.x1; xp[1] = r.x2; xp[2] = r.x1n; xp[3] = r.x2n; xp[4] = r.err;
#pragma unroll
                    for (int x = 0; x < 6; ++x) { xp[5 + x] = acc[x]; xp[11 + x] = sv[x]; }
                    xp[17] = r.x1_int ? -(double)(node + 1) : (double)(node + 1);  // sign: x_1 is the clamped int 0; magnitude: node id + 1
                }
                const double key = (a.criterion == APPLES_ME) ? r.x1 : r.err;
                if (key < best_key || (key == best_key && node < best_v)) {
                    best_key = key; best_v = node; best_sol = r; best_e = e;
                }
            }
            team_sync<TEAM>();
        }

        stamp(4);
        // ---------------------------------------------------------------- selection (apples/Algorithm.py:74-91)
        int win;
        const int my_best = best_v;
        if (a.criterion == APPLES_HYBRID) {
            // nsmallest(floor(log2(num_nodes))) by error (stable = ties to the smaller edge_index), then the
            // first minimum of x_1 among them in that order
            const int kk = 31 - __clz(V);
            double last_e = -INF_D;
            int last_v = -1;
            double bx = INF_D;
            int win_slot = -1;
            win = -1;
            for (int r = 0; r < kk; ++r) {
                double ke = INF_D;
                int kv = 0x7fffffff, kslot = -1;
                for (int i = tid; i < V; i += TEAM) {
                    const int v = I(ENT_NODE, i);
                    const double e = xe[(int64_t)i * XE_STRIDE + 4];
                    const bool after = (e > last_e) || (e == last_e && v > last_v);
                    if (after && (e < ke || (e == ke && v < kv))) { ke = e; kv = v; kslot = i; }
                }
                const int mine = kv;
                team_argmin<TEAM>(ke, kv, sh.d, sh.i);
                if (kv == 0x7fffffff) break;
                // the entry of the round's winner: node ids are unique within a query
                const unsigned long long holder = __ballot(mine == kv);
                int slot;
                if (TEAM == WAVE) {
                    slot = __shfl(kslot, __ffsll((long long)holder) - 1, WAVE);
                } else {
                    __syncthreads();
                    if (mine == kv) sh.w[0] = kslot;
                    __syncthreads();
                    slot = sh.w[0];
                    __syncthreads();
                }
                last_e = ke; last_v = kv;
                const double x1 = xe[(int64_t)slot * XE_STRIDE + 0];
                if (win < 0 || x1 < bx) { bx = x1; win = kv; win_slot = slot; }
            }
            if (win >= 0 && tid == 0) {
                const double *xp = xe + (int64_t)win_slot * XE_STRIDE;
                best_sol.x1 = xp[0]; best_sol.x2 = xp[1]; best_sol.err = xp[4];
                best_sol.x1_int = xp[17] < 0.0;
                best_e = Fe(win_slot);
            }
        } else {
            team_argmin<TEAM>(best_key, best_v, sh.d, sh.i);
            win = best_v;
        }
        const bool writer = (a.criterion == APPLES_HYBRID) ? (tid == 0) : (my_best == win && win != 0x7fffffff);
        if (win < 0 || win == 0x7fffffff) {
            if (tid == 0) {
                apples_placement pl = a.out[q];
                pl.n_valid = V;
                pl.edge = -1;
                pl.flags |= APPLES_F_DEGENERATE | APPLES_F_PENDANT_INT;
                a.out[q] = pl;
            }
        } else if (writer) {
            apples_placement pl = a.out[q];
            pl.n_valid = V;
            pl.edge = win;
            pl.error = best_sol.err;
            pl.distal = best_e - best_sol.x2;
            pl.pendant = best_sol.x1;
            pl.flags = 0;
            if (best_sol.x1_int) pl.flags |= APPLES_F_PENDANT_INT;
            if (best_sol.x1 == 0 && best_sol.err > 0 && (best_sol.x2 == 0 || best_sol.x2 == best_e)) pl.flags |= APPLES_F_MISPLACED;
            a.out[q] = pl;
        }
        if (tid == 0) {
            meta[0] = V;
            meta[1] = a.anc[loff[0] + top].node;  // the subtree's root (Subtree.root)
            meta[2] = top;
        }
        team_sync<TEAM>();
        stamp(5);
        if (a.prof && tid == 0) atomicAdd(&a.prof[6], 1ull);
    }
    if (a.prof && tid == 0)
        for (int k = 0; k < 6; ++k) atomicAdd(&a.prof[k], t_acc[k]);
}

__device__ __forceinline__ void scan_shared_init(ScanShared &sh) {
    for (int i = threadIdx.x; i < 384; i += APPLES_TPB) sh.pow[i] = (&kPowLogTab[0][0])[i];
    for (int i = threadIdx.x; i < 256; i += APPLES_TPB) sh.pow[384 + i] = __longlong_as_double((long long)kExpTab[i]);
    __syncthreads();
}

#ifndef APPLES_SCAN_WAVES
#define APPLES_SCAN_WAVES 2
#endif

template <int M, int TEAM>
__global__ __launch_bounds__(APPLES_TPB, APPLES_SCAN_WAVES) void k_scan(ScanArgs a, int64_t nq) {
    __shared__ ScanShared sh;
    scan_shared_init(sh);
    scan_team<M, TEAM>(a, nq, sh);
}

// One launch for a whole batch, as k_sweep_mixed: the first n_big workgroups first serve, as
// workgroup-sized teams with full-size scratch, the queries routed to them (many observed leaves), then
// every workgroup splits into four wavefront-sized teams that drain the size-class queues.
template <int M>
__global__ __launch_bounds__(APPLES_TPB, APPLES_SCAN_WAVES) void k_scan_mixed(ScanArgs small, ScanArgs big, int64_t nq, int n_big) {
    __shared__ ScanShared sh;
    scan_shared_init(sh);
    if ((int)blockIdx.x < n_big) {
        scan_team<M, APPLES_TPB>(big, nq, sh);
        __syncthreads();
    }
    scan_team<M, WAVE>(small, nq, sh);
}

int launch_scan_mixed(apples_ctx *ctx, const ScanArgs &small, const ScanArgs &big, int64_t nq, int wgs, int n_big, hipStream_t st) {
    if (nq == 0) return 0;
    const int64_t need = (nq + 3) / 4;
    dim3 grid((unsigned)std::min<int64_t>(std::max<int64_t>(need, std::min<int64_t>(n_big, nq)), wgs)), block(APPLES_TPB);
    if (n_big > (int)grid.x) n_big = (int)grid.x;
    switch (small.method) {
        case APPLES_FM: hipLaunchKernelGGL((k_scan_mixed<APPLES_FM>), grid, block, 0, st, small, big, nq, n_big); break;
        case APPLES_BME: hipLaunchKernelGGL((k_scan_mixed<APPLES_BME>), grid, block, 0, st, small, big, nq, n_big); break;
        case APPLES_BE: hipLaunchKernelGGL((k_scan_mixed<APPLES_BE>), grid, block, 0, st, small, big, nq, n_big); break;
        default: hipLaunchKernelGGL((k_scan_mixed<APPLES_OLS>), grid, block, 0, st, small, big, nq, n_big); break;
    }
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}

template <int TEAM>
static void launch_scan_team(const ScanArgs &a, int64_t nq, int wgs, hipStream_t st) {
    dim3 grid((unsigned)wgs), block(APPLES_TPB);
    switch (a.method) {
        case APPLES_FM: hipLaunchKernelGGL((k_scan<APPLES_FM, TEAM>), grid, block, 0, st, a, nq); break;
        case APPLES_BME: hipLaunchKernelGGL((k_scan<APPLES_BME, TEAM>), grid, block, 0, st, a, nq); break;
        case APPLES_BE: hipLaunchKernelGGL((k_scan<APPLES_BE, TEAM>), grid, block, 0, st, a, nq); break;
        default: hipLaunchKernelGGL((k_scan<APPLES_OLS, TEAM>), grid, block, 0, st, a, nq); break;
    }
}

int launch_scan(apples_ctx *ctx, const ScanArgs &a, int64_t nq, int wgs, int team, hipStream_t st) {
    if (nq == 0) return 0;
    if (!st) st = ctx->stream;
    if (team == 64) {
        const int64_t need = (nq + 3) / 4;
        launch_scan_team<64>(a, nq, (int)(need < wgs ? need : wgs), st);
    } else {
        launch_scan_team<256>(a, nq, (int)(nq < wgs ? nq : wgs), st);
    }
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}
