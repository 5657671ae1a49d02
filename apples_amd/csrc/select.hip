// Observed-set selection: which references a query keeps (one workgroup per query).
//
// Alignment input, apples/Reference.py:138-154: valid representatives are popped in ascending
// (distance, index) order; a cluster is expanded while d <= threshold OR fewer than `baseobs`
// valid member distances have been collected; the first representative failing both stops the
// walk.  Data-parallel restatement: accept every representative with 0 <= d <= thr; if that
// leaves obs_num < baseobs, keep accepting the smallest remaining (d, i) until it does not.
// The accepted set is therefore {d <= thr} U {(d, i) <= cut}, and only `cut` has to be stored.
//
// Distance-table input, apples/PoolQueryWorker.py:44-59: stable sort by distance, names not in
// the tree and negatives skipped, the `baseobs` nearest always kept, later ones only if <= thr:
// the same rule with every column its own cluster and i = column index.
//
// Then (PoolQueryWorker.py:63-98): drop the query's own entry, first zero distance in dict
// order -> exact placement, <= 2 distances -> unplaceable.  Survivors are written with an
// ordered compaction in slot order; slots are sorted by tree level, deepest first, so the sweep
// kernel receives its leaves grouped by level together with the per-level offsets cnt_gt.
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "libm_log.h"
#include "jc69_f4.h"

#define WAVE 64

__device__ __forceinline__ bool key_lt(double d1, int i1, double d2, int i2) { return d1 < d2 || (d1 == d2 && i1 < i2); }
__device__ __forceinline__ bool key_le(double d1, int i1, double d2, int i2) { return d1 < d2 || (d1 == d2 && i1 <= i2); }

__device__ __forceinline__ double shfl_down_f64(double v, int delta) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __shfl_down(lo, delta, WAVE);
    hi = __shfl_down(hi, delta, WAVE);
    return __hiloint2double(hi, lo);
}

// block-wide sum of an int; result valid in every thread (NW = the workgroup's wavefronts; sh holds NW entries)
template <int NW = APPLES_TPB / WAVE>
__device__ int block_sum(int v, int *sh) {
    for (int o = WAVE / 2; o > 0; o >>= 1) v += __shfl_down(v, o, WAVE);
    int w = threadIdx.x / WAVE;
    __syncthreads();
    if ((threadIdx.x & (WAVE - 1)) == 0) sh[w] = v;
    __syncthreads();
    int s = 0;
    for (int k = 0; k < NW; ++k) s += sh[k];
    return s;
}

// block-wide lexicographic arg-min over (d, i, j); result valid in every thread
template <int NW = APPLES_TPB / WAVE>
__device__ void block_argmin3(double &d, int &i, int &j, double *shd, int *shi, int *shj) {
    for (int o = WAVE / 2; o > 0; o >>= 1) {
        double d2 = shfl_down_f64(d, o);
        int i2 = __shfl_down(i, o, WAVE);
        int j2 = __shfl_down(j, o, WAVE);
        if (d2 < d || (d2 == d && (i2 < i || (i2 == i && j2 < j)))) { d = d2; i = i2; j = j2; }
    }
    int w = threadIdx.x / WAVE;
    __syncthreads();
    if ((threadIdx.x & (WAVE - 1)) == 0) { shd[w] = d; shi[w] = i; shj[w] = j; }
    __syncthreads();
    d = shd[0]; i = shi[0]; j = shj[0];
    for (int k = 1; k < NW; ++k) {
        double d2 = shd[k]; int i2 = shi[k], j2 = shj[k];
        if (d2 < d || (d2 == d && (i2 < i || (i2 == i && j2 < j)))) { d = d2; i = i2; j = j2; }
    }
}

// exclusive prefix sum of a 0/1 flag across the block; returns this thread's offset, total in *tot
template <int NW = APPLES_TPB / WAVE>
__device__ int block_excl_scan(int flag, int *sh, int *tot) {
    unsigned long long mask = __ballot(flag);
    int lane = threadIdx.x & (WAVE - 1), w = threadIdx.x / WAVE;
    int pre = __popcll(mask & ((1ull << lane) - 1ull));
    __syncthreads();
    if (lane == 0) sh[w] = __popcll(mask);
    __syncthreads();
    int base = 0, t = 0;
    for (int k = 0; k < NW; ++k) {
        if (k < w) base += sh[k];
        t += sh[k];
    }
    *tot = t;
    return base + pre;
}

// exclusive prefix sum of an int across the block; returns this thread's offset, total in *tot
template <int NW = APPLES_TPB / WAVE>
__device__ int block_excl_scan_int(int v, int *sh, int *tot) {
    int lane = threadIdx.x & (WAVE - 1), w = threadIdx.x / WAVE;
    int incl = v;
    for (int o = 1; o < WAVE; o <<= 1) {
        int t = __shfl_up(incl, o, WAVE);
        if (lane >= o) incl += t;
    }
    __syncthreads();
    if (lane == WAVE - 1) sh[w] = incl;
    __syncthreads();
    int base = 0, t = 0;
    for (int k = 0; k < NW; ++k) {
        if (k < w) base += sh[k];
        t += sh[k];
    }
    *tot = t;
    return base + incl - v;
}


// a placeable query joins the work list of its size class; the sweep's teams take queries from the
// largest class first (longest-processing-time order keeps the tail of the launch short)
__device__ __forceinline__ void enlist(const SelectArgs &a, int64_t q, int ne) {
    if (ne <= 0) return;
    if (ne > a.big_threshold && a.overflow_list) {
        if (a.route_classes) {  // by size, so that the longest jobs start first (counts at overflow_count[4..6])
            const int k = ne > 4 * a.big_threshold ? 0 : (ne > 2 * a.big_threshold ? 1 : 2);
            a.overflow_list[k * a.cls_stride + atomicAdd(&a.overflow_count[4 + k], 1)] = (int32_t)q;
        } else {
            a.overflow_list[atomicAdd(a.overflow_count, 1)] = (int32_t)q;
        }
        return;
    }
    if (!a.cls_list) return;
    const int k = ne > 1024 ? 0 : (ne > 512 ? 1 : (ne > 256 ? 2 : 3));
    a.cls_list[k * a.cls_stride + atomicAdd(&a.cls_count[k], 1)] = (int32_t)q;
    if (k == 0) {  // the largest class once more, four ways (lists 4..7, counts at [16..19]): sweep_lean.hip starts the longest jobs first
        const int u = ne > 8192 ? 0 : (ne > 4096 ? 1 : (ne > 2048 ? 2 : 3));
        a.cls_list[(4 + u) * a.cls_stride + atomicAdd(&a.cls_count[16 + u], 1)] = (int32_t)q;
    }
}

#define INF_D __longlong_as_double(0x7ff0000000000000LL)

#ifndef SELECT_E
#define SELECT_E 8  // slots per thread and round of the compaction pass (their loads are in flight together; 16: no faster)
#endif
// Ragged rows (common.h, Workspace::ragged): query q takes one of the big rows -- its flat member list is longer than a small row, or it
// leaves the fast phases for the top-up rule / the general selection, whose lists may grow to every leaf.  One thread per query.
// False: none left -- nothing is swept for the query (n_obs = 0) and row_fail is raised: run_block repeats the block with full rows.
// mark (phase 1 of k_select_clusters, before the query has joined any list): row_off = -1, the later phases leave the query alone.
// Later callers (phase 4, k_select) leave the row where it is: the query's items are in its clusters' lists by then and
// k_blocks_up, which may run beside them, reads the member distances of every item through row_off.
__device__ __forceinline__ bool row_make_big(const SelectArgs &a, int64_t q, bool mark) {
    if (!a.row_off || a.row_off[q] >= a.row_big_base) return true;
    if (a.row_off[q] < 0) return false;
    const int k = atomicAdd(a.row_big_cursor, 1);
    if (k >= a.row_big_n) {
        if (mark) a.row_off[q] = -1;
        *a.row_fail = 1;
        a.n_obs[q] = 0;
        return false;
    }
    a.row_off[q] = a.row_big_base + (int64_t)k * a.row_big_pitch;
    return true;
}

template <int TPB>
__global__ __launch_bounds__(TPB) void k_select(SelectArgs a) {
    constexpr int NW = TPB / WAVE;
    extern __shared__ double dyn_drep[];  // clustered rows: the row's distances to the representatives (a.rep_cache of them)
    __shared__ int sh_i[NW];
    __shared__ int sh_j[NW];
    __shared__ double sh_d[NW];
    // listed mode (top-up path): a fixed grid walks the device-side list; entry r names query
    // qlist[r], whose distances are row r
    const int64_t n_list = a.qcount ? (int64_t)*a.qcount : a.n_rows_plain;
    for (int64_t r = blockIdx.x; r < n_list; r += gridDim.x) {
    const int64_t q = a.qlist ? a.qlist[r] : r;
    const int tid = threadIdx.x;
    if (a.row_off) {  // (ragged rows: this selection's list may name every leaf)
        if (tid == 0) sh_i[0] = row_make_big(a, q, false) ? 1 : 0;
        __syncthreads();
        const int ok = sh_i[0];
        __syncthreads();
        if (!ok) continue;
    }
    const double *row = a.dist + (a.rows_by_query ? q : r) * a.stride;
    const int32_t *gather = a.gather;
    const int64_t nm = a.n_members;
    const int self = a.self_slot ? a.self_slot[q] : -1;
    const double thr = a.thr;
    const bool single = a.all_singleton != 0;
    const bool table = a.table_mode != 0;
#define DIST(s) (gather ? row[gather[(s)]] : row[(s)])
    // representative index used to break distance ties: heap of (d, i) at Reference.py:143, or the
    // column position for the stable sort at PoolQueryWorker.py:51
#define SLOT_KEYIDX(s) (a.slot_rep[(s)])  // table input: slot_rep holds the column of the slot
    // a member's key is its representative's distance: staged once per row (LDS) where the representatives fit, so that the
    // compaction pass below waits for one level of loads per slot and not for slot -> representative -> slot -> distance
#define DREP(ri) (a.rep_cache ? dyn_drep[(ri)] : DIST(a.rep_slot[(ri)]))
    if (a.rep_cache && !single) {
        for (int64_t ri = tid; ri < a.n_reps; ri += TPB) dyn_drep[ri] = DIST(a.rep_slot[ri]);
        __syncthreads();
    }

    // The compaction pass below also counts the observations inside the threshold (obs_num of
    // Reference.py:144-152: valid member distances of the clusters whose representative is within the
    // threshold); if they fall short of `-b` the top-up rule fixes the cut and the pass is repeated.
    int obs = 0;
    bool have_obs = false, topped = false;
    if (a.qhint && a.qhint[r] >= 0) {  // the listing kernel counted them already (k_select_clusters): straight to the top-up rule
        obs = a.qhint[r];
        have_obs = true;
    }
    double cut_d = -INF_D;
    int cut_i = -1;
    int base = 0, n_total = 0;
    double z_d = INF_D;
    int z_i = 0x7fffffff, z_p = 0x7fffffff;
    int64_t z_s = -1;  // the slot of that first zero (its node is looked up once, at the end)
    int32_t *o_node = a.obs_node + row_start(a.row_off, q, a.obs_cap);
    double *o_dist = a.obs_dist + row_start(a.row_off, q, a.obs_cap);
    int32_t *cg = a.cnt_gt ? a.cnt_gt + q * (int64_t)(a.height + 2) : nullptr;
    for (int round = 0; round < 2; ++round) {
    // ---- top-up: smallest (d, i) beyond the threshold until baseobs observations -------------------
    // Each thread first caches the KL smallest candidates of its own strided slice (one pass over
    // the row); every round then takes the block-wide minimum of the cache heads.  A thread whose
    // cache runs dry while its slice holds more candidates refills it with another pass over its
    // slice only.
    if (have_obs && !topped && obs < a.baseobs) {
        topped = true;
        constexpr int KL = 4;
        double cd[KL];
        int ci[KL];
        int head = 0, filled = 0;
        bool more = true;  // the slice may hold candidates beyond the cached ones
        const int64_t n_items = single ? nm : a.n_reps;
        auto refill = [&](double lo_d, int lo_i) {
            filled = 0;
            head = 0;
            int seen = 0;
            for (int64_t s = tid; s < n_items; s += TPB) {
                double d;
                int i;
                if (single) {
                    if (table && a.slot_node[s] < 0) continue;
                    d = DIST(s);
                    i = SLOT_KEYIDX(s);
                } else {
                    d = DREP(s);
                    i = (int)s;
                }
                if (!(d >= 0 && d > thr) || !key_lt(lo_d, lo_i, d, i)) continue;
                ++seen;
                // insertion into the sorted cache (ascending)
                int pos = filled < KL ? filled : KL;
                while (pos > 0 && key_lt(d, i, cd[pos - 1], ci[pos - 1])) --pos;
                if (pos < KL) {
                    for (int k = (filled < KL ? filled : KL - 1); k > pos; --k) { cd[k] = cd[k - 1]; ci[k] = ci[k - 1]; }
                    cd[pos] = d; ci[pos] = i;
                    if (filled < KL) ++filled;
                }
            }
            more = seen > filled;
        };
        refill(cut_d, cut_i);
        while (obs < a.baseobs) {
            if (head == filled && more) refill(cut_d, cut_i);
            double bd = head < filled ? cd[head] : INF_D;
            int bi = head < filled ? ci[head] : 0x7fffffff, bj = 0;
            const int mine = bi;
            block_argmin3<NW>(bd, bi, bj, sh_d, sh_i, sh_j);
            if (bi == 0x7fffffff) break;  // nothing left
            if (mine == bi && head < filled) ++head;
            cut_d = bd;
            cut_i = bi;
            if (single) obs += 1;
            else {
                int c = 0;
                for (int m = a.rep_moff[bi] + tid; m < a.rep_moff[bi + 1]; m += TPB) c += !(DIST(a.mem_slot[m]) < 0);
                obs += block_sum<NW>(c, sh_i);
            }
        }
    }

    // ---- pass B: ordered compaction of the observed leaves, SELECT_E consecutive slots per thread ----------
    base = 0;       // emitted so far
    n_total = 0;    // len(obs_dist) after the self entry is removed
    // first zero distance in dict order: min over (d_rep, rep index, member position)
    z_d = INF_D; z_i = 0x7fffffff; z_p = 0x7fffffff; z_s = -1;
    int thr_cnt = 0;  // observations inside the threshold (the obs_num the top-up rule looks at)
    constexpr int E = SELECT_E;
    static_assert(E % 4 == 0, "the compaction pass loads its slots four at a time");
    // a thread's E consecutive slots come in 16-byte pieces (a wavefront's load instruction then covers whole cache lines:
    // with one 4- or 8-byte element per lane at a stride of E elements the pass is bound by the rate of line requests, not by
    // latency -- 24 us per 8 192 slots); rows whose start is not 16-byte aligned and gathered rows take the plain loads
    const bool vec_ok = !gather && (reinterpret_cast<uintptr_t>(row) & 15) == 0;
    for (int64_t s0 = 0; s0 <= nm; s0 += (int64_t)TPB * E) {
        const int64_t sb = s0 + (int64_t)tid * E;
        int emit[E], node[E], v_rep[E], v_mp[E], v_lv[E];
        double dm[E];
        if (vec_ok && sb + E <= nm) {
#pragma unroll
            for (int k = 0; k < E / 4; ++k) {
                const int4 t = *reinterpret_cast<const int4 *>(a.slot_node + sb + 4 * k);
                node[4 * k] = t.x; node[4 * k + 1] = t.y; node[4 * k + 2] = t.z; node[4 * k + 3] = t.w;
                const int4 r = *reinterpret_cast<const int4 *>(a.slot_rep + sb + 4 * k);
                v_rep[4 * k] = r.x; v_rep[4 * k + 1] = r.y; v_rep[4 * k + 2] = r.z; v_rep[4 * k + 3] = r.w;
                int4 m = make_int4(0, 0, 0, 0), l = make_int4(0, 0, 0, 0);
                if (!single) m = *reinterpret_cast<const int4 *>(a.slot_mpos + sb + 4 * k);
                if (cg) l = *reinterpret_cast<const int4 *>(a.slot_level + sb + 4 * k);
                v_mp[4 * k] = m.x; v_mp[4 * k + 1] = m.y; v_mp[4 * k + 2] = m.z; v_mp[4 * k + 3] = m.w;
                v_lv[4 * k] = l.x; v_lv[4 * k + 1] = l.y; v_lv[4 * k + 2] = l.z; v_lv[4 * k + 3] = l.w;
            }
#pragma unroll
            for (int k = 0; k < E / 2; ++k) {
                const double2 t = *reinterpret_cast<const double2 *>(row + sb + 2 * k);
                dm[2 * k] = t.x; dm[2 * k + 1] = t.y;
            }
        } else {
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const int64_t s = sb + e;
                node[e] = -1; dm[e] = -1.0; v_rep[e] = 0; v_mp[e] = 0; v_lv[e] = -1;
                if (s < nm) {
                    node[e] = a.slot_node[s];
                    dm[e] = DIST(s);
                    v_rep[e] = a.slot_rep[s];
                    if (!single) v_mp[e] = a.slot_mpos[s];
                    if (cg) v_lv[e] = a.slot_level[s];
                }
            }
        }
        const int lv_before = (cg && sb > 0 && sb <= nm) ? a.slot_level[sb - 1] : a.height + 1;  // level of the slot before this thread's first
        int n_emit_l = 0;
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int64_t s = sb + e;
            emit[e] = 0;
            if (s < nm) {
                bool in_dict;
                double drep;
                int ri = v_rep[e], mp;  // (table input: slot_rep holds the column of the slot)
                if (single) {
                    drep = dm[e]; mp = 0;
                    const bool ok = (dm[e] >= 0) && !(table && node[e] < 0);
                    thr_cnt += ok && dm[e] <= thr;
                    in_dict = ok && (dm[e] <= thr || key_le(dm[e], ri, cut_d, cut_i));
                } else {
                    mp = v_mp[e];
                    drep = DREP(ri);
                    const bool member_ok = !(dm[e] < 0);
                    thr_cnt += (drep >= 0) && (drep <= thr) && member_ok;
                    in_dict = (drep >= 0) && (drep <= thr || key_le(drep, ri, cut_d, cut_i)) && member_ok;
                }
                if (in_dict && (int)s != self) {
                    n_total++;
                    if (dm[e] == 0 && (drep < z_d || (drep == z_d && (ri < z_i || (ri == z_i && mp < z_p))))) {
                        z_d = drep; z_i = ri; z_p = mp; z_s = s;
                    }
                    emit[e] = node[e] >= 0;
                }
            }
            n_emit_l += emit[e];
        }
        int tot;
        int pos = base + block_excl_scan_int<NW>(n_emit_l, sh_i, &tot);
        int lprev = lv_before;  // (carried along: no v_lv[e - 1], which at e = 0 would name an element before the array)
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int64_t s = sb + e;
            if (cg && s <= nm) {  // level boundaries (virtual end slot nm has level -1)
                const int lv = (s < nm) ? v_lv[e] : -1;
                for (int l = lv; l < lprev; ++l) cg[l + 1] = pos;
            }
            lprev = v_lv[e];
            if (emit[e]) { o_node[pos] = node[e]; o_dist[pos] = dm[e]; ++pos; }
        }
        base += tot;
    }
    if (!have_obs) {  // first round: was the threshold set large enough?
        obs = block_sum<NW>(thr_cnt, sh_i);
        have_obs = true;
        if (obs < a.baseobs) continue;  // no: apply the top-up rule and compact again
    }
    break;
    }  // round
    n_total = block_sum<NW>(n_total, sh_i);
    // pack (z_i, z_p) is not needed beyond ordering; carry the node through a second reduction
    double zd = z_d; int zi = z_i, zp = z_p;
    block_argmin3<NW>(zd, zi, zp, sh_d, sh_i, sh_j);
    __shared__ int sh_znode;
    if (tid == 0) sh_znode = -2;
    __syncthreads();
    if (z_s >= 0 && z_d == zd && z_i == zi && z_p == zp) sh_znode = a.slot_node[z_s];
    __syncthreads();

    if (tid == 0) {
        apples_placement p;
        p.edge = 0; p.flags = 0; p.error = 0.0; p.distal = 0.0; p.pendant = 0.0; p.n_obs = n_total; p.n_valid = 0;
        int n_emit = base;
        if (zi != 0x7fffffff) {
            p.flags = APPLES_F_EXACT | APPLES_F_PENDANT_INT;
            p.edge = sh_znode;
            if (sh_znode < 0) { p.flags |= APPLES_F_ZERO_NOT_IN_TREE; p.edge = -1; }
            n_emit = 0;
        } else if (n_total <= 2) {
            p.flags = APPLES_F_INSUFFICIENT | APPLES_F_PENDANT_INT;
            p.edge = -1;
            n_emit = 0;
        } else if (n_emit < 2) {
            p.flags = APPLES_F_DEGENERATE | APPLES_F_PENDANT_INT;
            p.edge = -1;
            n_emit = 0;
        }
        a.out[q] = p;
        a.n_obs[q] = n_emit;  // 0 = nothing for the sweep to do
        enlist(a, q, n_emit);
    }
    __syncthreads();  // shared scratch is reused by the next list entry
    }
#undef DIST
#undef DREP
#undef SLOT_KEYIDX
}

// Fast selection for all-singleton alignment input: consumes the segments the fused distance
// kernel wrote (k_jc69 MODE 1: per 64-slot segment, the entries with 0 <= d <= thr in slot order).
// If they number fewer than `baseobs` the query needs the top-up rule of Reference.py:144 and is
// pushed to the slow list (full rows + k_select).  Otherwise the observed dict is exactly those
// entries: drop the query's own row, find the first zero (smallest representative index, since all
// zero keys tie on d = 0), count, and compact the tree leaves in slot order (= level order).
template <int TPB>
__global__ __launch_bounds__(TPB) void k_select_fast(SelectArgs a) {
    constexpr int NW = TPB / WAVE;
    __shared__ int sh_i[8];
    __shared__ int sh_j[8];
    __shared__ double sh_d[8];
    __shared__ int sh_pref[TPB + 1];
    const int64_t q = blockIdx.x;
    const int tid = threadIdx.x;
    const int64_t n_seg = a.stride >> 6;
    const int32_t *cnt = a.seg_cnt + q * n_seg;
    const int32_t *sslot = a.seg_slot + q * a.stride;
    const double *sd = a.dist + q * a.stride;
    const int self = a.self_slot ? a.self_slot[q] : -1;
    // Rows of up to 16 384 segments (a million references): the exclusive prefix of the segment counts goes to
    // LDS once, and the survivors are then walked as one flat list -- four rounds of loads for a thousand
    // survivors instead of one per 256-segment chunk.  Longer rows take the chunked loop below.
    extern __shared__ int dyn_pref[];
    const bool flat = a.flat_pref != 0;
    int total;
    if (flat) {
        const int K = (int)((n_seg + TPB - 1) / TPB);
        const int64_t s_lo = (int64_t)tid * K, s_hi = s_lo + K < n_seg ? s_lo + K : n_seg;
        // (the counts come in coalesced and are summed out of LDS: a thread reading its K consecutive counts from memory is K
        // load instructions of 64 cache lines each)
        for (int64_t s = tid; s < n_seg; s += TPB) dyn_pref[s] = cnt[s];
        __syncthreads();
        int local = 0;
        for (int64_t s = s_lo; s < s_hi; ++s) { const int v = dyn_pref[s]; dyn_pref[s] = local; local += v; }
        const int at = block_excl_scan_int<NW>(local, sh_i, &total);
        for (int64_t s = s_lo; s < s_hi; ++s) dyn_pref[s] += at;
        if (tid == 0) dyn_pref[n_seg] = total;
    } else {
        int c = 0;
        for (int64_t s = tid; s < n_seg; s += TPB) c += cnt[s];
        total = block_sum<NW>(c, sh_i);
    }
    // (reference rows with bytes beyond ACGT-, a.ex_off / a.ex_mmax: a survivor may still fail once those sites are counted -- the
    // walk below recounts it -- so fewer than `baseobs` survivors is known for sure only at the end; with fewer than that to begin
    // with the answer is here already)
    if ((a.seg_surv ? a.seg_surv[q] : total) < a.baseobs) {
        __syncthreads();  // (seg_surv may be n_obs itself: every thread has read it)
        if (tid == 0) {
            a.slow_list[atomicAdd(a.slow_count, 1)] = (int32_t)q;
            a.n_obs[q] = 0;
        }
        return;
    }
    int32_t *o_node = a.obs_node + q * a.obs_cap;
    double *o_dist = a.obs_dist + q * a.obs_cap;
    int32_t *cg = a.cnt_gt ? a.cnt_gt + q * (int64_t)(a.height + 2) : nullptr;
    int base = 0, n_total = 0, n_drop = 0;
    int z_i = 0x7fffffff, z_node = -2;
    const bool exm = a.ex_mmax != nullptr;  // the survivors' counts once more where the row holds bytes beyond ACGT- (below)
    const uint8_t *qrow = exm ? a.q_raw + q * (int64_t)a.L : nullptr;
    // A reference row with bytes beyond ACGT-, which the matrix-core pass took for gaps: every such site under a letter of the
    // query is a valid site and a mismatch (apples/distance.py:733-737: the bytes differ; a query with such bytes itself is not
    // served here, api.hip:exotic_queries), so the pair's counts are (valid + k, mism + k) -- it can only fail now: 0 <= d <=
    // threshold once more on the integers, the rule's own table (the pass's was loosened at its lower end where -V could have
    // kept the smaller count out: then every survivor is tested again, ex_all).  False: the pair is no survivor.
    auto recount = [&](int slot, long long &valid, long long &mism) -> bool {
        const int e0 = a.ex_off ? a.ex_off[slot] : 0, e1 = a.ex_off ? a.ex_off[slot + 1] : 0;
        if (e1 == e0 && !a.ex_all) return true;
        int add = 0;
        for (int e = e0; e < e1; ++e) {
            const uint8_t b = qrow[a.ex_site[e]];
            add += (b == 'A' || b == 'C' || b == 'G' || b == 'T') ? 1 : 0;
        }
        valid += add; mism += add;
        return mism <= a.ex_mmax[valid];
    };
    // one survivor: entry `off` of segment `seg` -> (slot, distance), own row dropped, first zero noted
    auto take = [&](int64_t seg, int off, int &node, double &d, int &lv) -> int {
        const int64_t src = seg * 64 + off;
        int slot = sslot[src];
        if (a.seg_lut) {  // the matrix-core distance pass leaves position | valid | mism; same table, same bits
            const uint32_t pk = (uint32_t)slot;
            long long valid = (pk >> 13) & 0x1fffu, mism = pk & 0x1fffu;
            slot = (int)(seg * 64 + (pk >> 26));
            if (exm && !recount(slot, valid, mism)) { ++n_drop; return 0; }
            d = a.seg_lut[valid * (valid + 1) / 2 + mism];
        } else {
            d = sd[src];
        }
        if (slot == self || (a.seg_surv && d < 0)) return 0;
        ++n_total;
        node = a.slot_node[slot];
        lv = a.slot_level[slot];  // (beside the node: the per-level offsets below need no second pass over the list)
        if (d == 0) {
            const int ri = a.slot_rep[slot];
            if (ri < z_i) { z_i = ri; z_node = node; }
        }
        return node >= 0;
    };
    // ordered emission of up to 256 entries (one per thread) + the per-level offsets they imply (the sweep's cnt_gt:
    // cg[l + 1] = entries with a level above l = position of the first entry at level l or below; entries come in
    // level order, so an entry whose predecessor sits at a higher level writes the offsets in between)
    __shared__ int sh_lv[TPB + 1];
    int last_lv = a.height + 1;  // level of the entry before this round's first (block-uniform)
    auto put = [&](int emit, int node, double d, int lv) {
        int tot;
        const int r = block_excl_scan<NW>(emit, sh_j, &tot);
        if (emit) { o_node[base + r] = node; o_dist[base + r] = d; sh_lv[r + 1] = lv; }
        if (tid == 0) sh_lv[0] = last_lv;
        __syncthreads();
        if (emit && cg)
            for (int l = lv; l < sh_lv[r]; ++l) cg[l + 1] = base + r;
        if (tot > 0) last_lv = sh_lv[tot];
        base += tot;
        __syncthreads();
    };
    if (flat) {
        __syncthreads();
        // EF rounds of 256 survivors at a time: entry -> (slot, counts) -> (distance, node, level) are three dependent lookups, and
        // a workgroup's time is their latency; with the rounds' lookups in flight together a typical query (1 000 survivors) pays
        // for them once.  Addresses of lanes beyond the list are clamped to entry 0 so that every load is unconditional.
        constexpr int EF = 1024 / TPB;  // (a thousand survivors per outer round)
        for (int e0 = 0; e0 < total; e0 += TPB * EF) {
            int64_t seg_[EF];
            int raw_[EF], slot_[EF], node_[EF], lv_[EF];
            double d_[EF];
            bool in_[EF];
#pragma unroll
            for (int u = 0; u < EF; ++u) {
                const int e = e0 + u * TPB + tid;
                in_[u] = e < total;
                int64_t lo = 0, hi = n_seg + 1;  // last segment whose prefix <= e (empty segments share a prefix: the last one wins, and holds e)
                while (hi - lo > 1) {
                    const int64_t mid = (lo + hi) >> 1;
                    if (dyn_pref[mid] <= (in_[u] ? e : 0)) lo = mid; else hi = mid;
                }
                seg_[u] = lo;
                const int64_t src = in_[u] ? lo * 64 + (e - dyn_pref[lo]) : 0;
                raw_[u] = sslot[src];
                d_[u] = a.seg_lut ? 0.0 : sd[src];
            }
#pragma unroll
            for (int u = 0; u < EF; ++u) {
                slot_[u] = raw_[u];
                if (a.seg_lut) {  // the matrix-core distance pass leaves position | valid | mism; same table, same bits
                    const uint32_t pk = (uint32_t)raw_[u];
                    long long valid = (pk >> 13) & 0x1fffu, mism = pk & 0x1fffu;
                    slot_[u] = (int)(seg_[u] * 64 + (pk >> 26));
                    if (exm && in_[u] && !recount(slot_[u], valid, mism)) { in_[u] = false; ++n_drop; }
                    d_[u] = a.seg_lut[in_[u] ? valid * (valid + 1) / 2 + mism : 0];
                }
                const int sl = in_[u] ? slot_[u] : 0;
                node_[u] = a.slot_node[sl];
                lv_[u] = a.slot_level[sl];  // (beside the node: the per-level offsets below need no second pass over the list)
            }
#pragma unroll
            for (int u = 0; u < EF; ++u) {
                if (e0 + u * TPB >= total) break;  // (block-uniform)
                int emit = 0;
                if (in_[u] && slot_[u] != self && !(a.seg_surv && d_[u] < 0)) {  // own row dropped, first zero noted
                    ++n_total;
                    if (d_[u] == 0) {
                        const int ri = a.slot_rep[slot_[u]];
                        if (ri < z_i) { z_i = ri; z_node = node_[u]; }
                    }
                    emit = node_[u] >= 0;
                }
                put(emit, node_[u], d_[u], lv_[u]);
            }
        }
    } else {
    for (int64_t s0 = 0; s0 < n_seg; s0 += TPB) {
        const int64_t s = s0 + tid;
        const int my = s < n_seg ? cnt[s] : 0;
        int chunk_total;
        const int pre = block_excl_scan_int<NW>(my, sh_i, &chunk_total);
        sh_pref[tid] = pre;
        if (tid == 0) sh_pref[TPB] = chunk_total;
        __syncthreads();
        // flat loop over this chunk's entries, in order
        for (int e0 = 0; e0 < chunk_total; e0 += TPB) {
            const int e = e0 + tid;
            int emit = 0, node = -1, lv = 0;
            double d = 0;
            if (e < chunk_total) {
                int lo = 0, hi = TPB;  // last segment whose prefix <= e
                while (hi - lo > 1) {
                    int mid = (lo + hi) >> 1;
                    if (sh_pref[mid] <= e) lo = mid; else hi = mid;
                }
                emit = take(s0 + lo, e - sh_pref[lo], node, d, lv);
            }
            put(emit, node, d, lv);
        }
        __syncthreads();
    }
    }
    n_total = block_sum<NW>(n_total, sh_i);
    if (exm) {  // survivors that failed with their rows' other bytes counted: fewer than `baseobs` left after all -> the top-up rule's list
        __syncthreads();
        n_drop = block_sum<NW>(n_drop, sh_i);
        if (total - n_drop < a.baseobs) {
            if (tid == 0) {
                a.slow_list[atomicAdd(a.slow_count, 1)] = (int32_t)q;
                a.n_obs[q] = 0;
            }
            return;
        }
        __syncthreads();
    }
    double zd = 0; int zi = z_i, zp = 0;
    block_argmin3<NW>(zd, zi, zp, sh_d, sh_i, sh_j);
    __shared__ int sh_znode;
    if (tid == 0) sh_znode = -2;
    __syncthreads();
    if (z_i == zi && zi != 0x7fffffff) sh_znode = z_node;
    __syncthreads();
    const int n_emit = base;
    // the offsets below the last entry's level (everything sits above those levels)
    for (int l = -1 + tid; cg && l < last_lv; l += TPB) cg[l + 1] = n_emit;
    if (tid == 0) {
        apples_placement p;
        p.edge = 0; p.flags = 0; p.error = 0.0; p.distal = 0.0; p.pendant = 0.0; p.n_obs = n_total; p.n_valid = 0;
        int ne = n_emit;
        if (zi != 0x7fffffff) {
            p.flags = APPLES_F_EXACT | APPLES_F_PENDANT_INT;
            p.edge = sh_znode;
            if (sh_znode < 0) { p.flags |= APPLES_F_ZERO_NOT_IN_TREE; p.edge = -1; }
            ne = 0;
        } else if (n_total <= 2) {
            p.flags = APPLES_F_INSUFFICIENT | APPLES_F_PENDANT_INT;
            p.edge = -1;
            ne = 0;
        } else if (ne < 2) {
            p.flags = APPLES_F_DEGENERATE | APPLES_F_PENDANT_INT;
            p.edge = -1;
            ne = 0;
        }
        a.out[q] = p;
        a.n_obs[q] = ne;
        enlist(a, q, ne);
    }
}

// Fused selection for clustered references (the command line's default route, apples/Reference.py:117-157):
// the matrix-core distance pass ran over the REPRESENTATIVES only and left, per query, the ones with
// 0 <= d <= threshold (k_jc69_mfma<1> on the representative panel).  Those clusters are the accepted ones
// unless they hold fewer than `baseobs` valid member distances -- then the reference keeps walking its
// heap, and the query goes to the slow list (full rows + k_select), as in k_select_fast.  One workgroup
// per query: the accepted clusters' members are expanded to a flat list, every (query, member) distance comes
// from the member's words in the cluster-major panel (the pair counts of k_jc69, same table lookup), the member's
// slot is marked in an LDS bitmap; ranks of the bitmap give the slot-ordered (= level-ordered) observation list
// the sweep wants.
//
// Who computes the member distances (PHASE):
//   0  this kernel, a thread per (query, member): every query reads the words of all its members, 384 B each at L = 1000
//      -- 120 GB per 100 k queries against a 77 MB panel, bound by what the Infinity Cache delivers (2.6 ms per 14 300
//      queries); kept for APPLES_CLUSTER_BY_QUERY and as the reference form of the phases below;
//   1 -> k_cluster_tiles -> 2 -> k_cluster_dist -> 3  (default): cluster-major.  Phase 1 counts, per cluster, the queries that
//      accepted it; k_cluster_tiles turns the counts into list offsets and tiles of up to 64 queries; phase 2 writes the
//      lists (query, where the cluster's members start in the query's flat member list); k_cluster_dist takes a tile, keeps
//      a member's words in registers and the tile's query words in LDS, and writes the distances into the queries' rows;
//      phase 3 is phase 0 with the distance read from the row.  The panel is then read once per tile, not once per query.
//   4  the slow list (fewer than `baseobs` valid member distances inside the threshold): the query's distance to EVERY
//      representative, the top-up rule over them, the members of the clusters it accepts (by-query arithmetic), then as phase 3;
//      what it cannot hold goes on to full rows + k_select (launch_select_clusters_listed).
#ifndef CLUSTER_UNROLL
#define CLUSTER_UNROLL 1  // phase 0: member word groups in flight per lane (3 16-byte loads each); 38 / 42 / 48 / 43 ms per C3 pass
                          // with 1 / 2 / 4 / 8 (the loop's load schedule is fragile: the same source measured 41 ms per pass
                          // where an unrelated change had the compiler interleave the six loads with their uses)
#endif
// pair counts of one 128-site word group: rm / r0 / r1 = the member's mask and two code planes, qm / q0 / q1 the query's
__device__ __forceinline__ void cluster_count(const uint4 &rm, const uint4 &r0, const uint4 &r1, const uint4 &qm, const uint4 &q0,
                                              const uint4 &q1, uint32_t &nv, uint32_t &nmis) {
    const uint32_t m0_ = qm.x & rm.x, m1_ = qm.y & rm.y, m2_ = qm.z & rm.z, m3_ = qm.w & rm.w;
    nv += __popc(m0_) + __popc(m1_) + __popc(m2_) + __popc(m3_);
    nmis += __popc(((q0.x ^ r0.x) | (q1.x ^ r1.x)) & m0_) + __popc(((q0.y ^ r0.y) | (q1.y ^ r1.y)) & m1_) +
            __popc(((q0.z ^ r0.z) | (q1.z ^ r1.z)) & m2_) + __popc(((q0.w ^ r0.w) | (q1.w ^ r1.w)) & m3_);
}

//   CAP = accepted clusters a workgroup's lists hold: ACC_CAP for the launch over all queries; the queries beyond it (1 % at C3
//   size: observed sets of tens of thousands) are listed by phase 1 and go through the same phases once more with CAP = BIG_CAP,
//   a workgroup per list entry (LISTED; 60 KB of lists, so one workgroup per CU: a launch of their own keeps that occupancy
//   and their long rounds away from the other 99 %).
//   SHORT_ONLY (phase 3 of a context with clade blocks): the launch over all queries serves the short form alone -- 18 KB of dynamic
//   LDS instead of the bitmap's 33 at 200 000 leaves: five workgroups per CU where the general form has three, and the phase is a
//   chain of dependent look-ups (its time follows the occupancy: + 1.1 ms per pass with two per CU) -- and lists the queries that need
//   the general form (a.gen_list), which a LISTED launch takes right behind it.
template <int PHASE, int CAP = SELECT_CLUSTERS_ACC_CAP, bool LISTED = (PHASE == 4), int TPB = APPLES_TPB, bool SHORT_ONLY = false>
__global__ __launch_bounds__(TPB) void k_select_clusters(SelectArgs a) {
    constexpr int ACC_CAP = CAP, NW = TPB / WAVE;
    extern __shared__ unsigned long long dyn_bits[];  // [n_words] member bits in slot order, then uint16 [n_words]: the set bits before the
                                                      // word inside its thread's run of 16 words (sh_base: before the run)
    __shared__ int sh_base[TPB];
    __shared__ int sh_i[NW < 8 ? 8 : NW];
    __shared__ int sh_j[NW < 8 ? 8 : NW];
    __shared__ double sh_d[NW < 8 ? 8 : NW];
    __shared__ uint4 sh_q[(PHASE == 0 || PHASE == 4) ? 64 * 3 : 1];
    __shared__ double sh_T[PHASE == 4 ? 21 * 21 : 1];  // scoredist contexts: the table (phase 4 forms member distances itself)
    // (CAP beyond BIG_CAP, the third form: references of more than BIG_CAP clusters -- the offsets, which every member's binary
    // search reads, stay in LDS; the other three lists are rows of a scratch buffer in global memory, one set per workgroup)
    constexpr bool GL = CAP > SELECT_CLUSTERS_BIG_CAP;
    __shared__ int sh_rep_[GL ? 1 : ACC_CAP];
    __shared__ int sh_off[ACC_CAP + 1];
    __shared__ int sh_mb_[GL ? 1 : ACC_CAP];  // first member (index into mem_slot / the cluster-major panel) of every accepted cluster
    __shared__ int sh_sb_[(PHASE == 3 && !GL) ? ACC_CAP : 1];  // clade blocks: where the tuples of the cluster's blocks are for this query (first slot x 64 + lane), -1: no blocks
    int *const sh_rep = GL ? a.big_scr + (int64_t)blockIdx.x * 3 * ACC_CAP : sh_rep_;
    int *const sh_mb = GL ? sh_rep + ACC_CAP : sh_mb_;
    int *const sh_sb = GL ? sh_rep + 2 * ACC_CAP : sh_sb_;
    __shared__ int sh_znode, sh_nacc;
    if (PHASE == 4) {
        // phase 4: a workgroup per entry of the slow list, up to the grid (the launcher does not know the list's length and a
        // workgroup holds 80 KB of LDS: one grid's worth is served here, entries beyond it go on to the general route)
        const int n_list = *a.qcount;
        if (threadIdx.x == 0)
            for (int r2 = blockIdx.x + gridDim.x; r2 < n_list; r2 += gridDim.x) a.slow2_list[atomicAdd(a.slow2_count, 1)] = a.qlist[r2];
        if ((int)blockIdx.x >= n_list) return;
    }
    if (PHASE != 4 && LISTED) {
        const int n_list = *a.qcount < (int)gridDim.x ? *a.qcount : (int)gridDim.x;  // (the list is capped at the grid where it is written)
        if ((int)blockIdx.x >= n_list) return;
    }
    const int64_t q = LISTED ? a.qlist[blockIdx.x] : blockIdx.x;
    const int tid = threadIdx.x;
    const int G = a.G;
    const int64_t nm = a.n_members;
    // (with clade blocks the bitmap is over emission indices: tree-leaf slots and block roots in level order)
    const bool blk_space = PHASE == 3 && a.e_of_slot != nullptr;
    const int n_words = (int)(((blk_space ? a.n_e : nm) + 63) >> 6);
    uint16_t *pre = reinterpret_cast<uint16_t *>(dyn_bits + n_words);
    double *reprow = reinterpret_cast<double *>(dyn_bits + n_words + (n_words + 3) / 4);  // phase 4: the query's distance to every representative
    const int self = a.self_slot ? a.self_slot[q] : -1;
    if (a.row_off) {
        if (PHASE == 4) {  // (ragged rows: the top-up rule may add clusters up to every leaf)
            if (threadIdx.x == 0) sh_znode = row_make_big(a, q, false) ? 1 : 0;
            __syncthreads();
            if (!sh_znode) return;
            __syncthreads();
        } else if (PHASE != 1 && a.row_off[q] < 0) return;  // (phase 1 found no big row for it: row_make_big)
    }
    int32_t *o_node = a.obs_node + row_start(a.row_off, q, a.obs_cap);
    double *o_dist = a.obs_dist + row_start(a.row_off, q, a.obs_cap);
    double *tmp = a.tmp_d + row_start(a.row_off, q, a.stride);
    auto to_slow = [&](int known_obs) {  // known_obs: the observations inside the threshold if they were counted, else -1
        if (tid == 0) {
            const int at = atomicAdd(a.slow_count, 1);
            a.slow_list[at] = (int32_t)q;
            if (a.slow_hint) a.slow_hint[at] = known_obs;
            a.n_obs[q] = 0;
        }
    };
    auto forward = [&]() {  // phase 4 cannot serve the query: the general route (full rows + k_select) takes it
        if (tid == 0) a.slow2_list[atomicAdd(a.slow2_count, 1)] = (int32_t)q;
    };
    if (PHASE == 4 && (a.qhint[blockIdx.x] < 0 || !a.rep_cache)) { forward(); return; }
    const bool sd = PHASE == 4 && a.aa_idx != nullptr;  // scoredist context
    if ((PHASE == 0 || PHASE == 4) && !sd) {
        // the query's packed words (tile layout of pack.hip: [(q/16)*G + g][q%16][plane])
        for (int i = tid; i < G * 3; i += TPB) {
            const int g = i / 3, pl = i % 3;
            sh_q[i] = a.qpacked[(((q >> 4) * G + g) * 16 + (q & 15)) * 3 + pl];
        }
    }
    if (sd)
        for (int i = tid; i < 21 * 21; i += TPB) sh_T[i] = a.table[i];
    // scoredist of the query and the reference row in `slot`: the arithmetic of k_scoredist (dist.hip: fp64, sites left to right,
    // the table in LDS -- the same bits); the query's residues are the same for every lane (one table row per site)
    auto sd_dist = [&](int64_t slot) -> double {
        const char *Tb = reinterpret_cast<const char *>(sh_T);
        const int n16 = a.Lpad / 16;
        double tot = 0.0;
        uint32_t nv = 0;
        for (int s16 = 0; s16 < n16; ++s16) {
            const uint4 rw = *reinterpret_cast<const uint4 *>(a.aa_idx + ((int64_t)s16 * a.stride + slot) * 16);
            const uint32_t rmask = a.aa_mask[(int64_t)s16 * a.stride + slot];
            const uint4 qw = *reinterpret_cast<const uint4 *>(a.q_aa + q * (int64_t)a.Lpad + s16 * 16);
            nv += __popc(rmask & (uint32_t)a.q_aam[q * (int64_t)n16 + s16]);
            const uint32_t rr[4] = {rw.x, rw.y, rw.z, rw.w}, qq[4] = {qw.x, qw.y, qw.z, qw.w};
            double v[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const uint32_t r8 = (rr[k >> 2] >> (8 * (k & 3))) & 0xffu;
                const uint32_t qrow = ((qq[k >> 2] >> (8 * (k & 3))) & 0xffu) * 168u;
                v[k] = *reinterpret_cast<const double *>(Tb + qrow + r8);
            }
#pragma unroll
            for (int k = 0; k < 16; ++k) tot += v[k];
        }
        if (nv == 0 || (double)nv / (double)a.L < a.overlap) return -1.0;
        const double r1 = 1 - tot / (double)nv;
        if (0 >= r1) return -1.0;
        return -log_libm(r1) * 1.3;
    };
    if (!SHORT_ONLY && (PHASE == 0 || PHASE >= 3))
        for (int i = tid; i < n_words; i += TPB) dyn_bits[i] = 0;
    // a member's (or representative's) distance from its words: word (g, plane) at base[(g * 3 + plane) * stride] -- the
    // cluster-major panel with the cluster's size as stride (the lanes holding consecutive members read consecutive 16
    // bytes), or the representative panel with its padded length
    auto by_query = [&](const uint4 *row, int64_t sz) -> double {
        uint32_t nv = 0, nmis = 0;
        int g = 0;
        for (; g + CLUSTER_UNROLL <= G; g += CLUSTER_UNROLL) {  // 3 x CLUSTER_UNROLL loads in flight before the first is used
            uint4 w[3 * CLUSTER_UNROLL];
#pragma unroll
            for (int k = 0; k < 3 * CLUSTER_UNROLL; ++k) w[k] = row[(int64_t)(g * 3 + k) * sz];
#pragma unroll
            for (int k = 0; k < CLUSTER_UNROLL; ++k)
                cluster_count(w[3 * k], w[3 * k + 1], w[3 * k + 2], sh_q[(g + k) * 3], sh_q[(g + k) * 3 + 1], sh_q[(g + k) * 3 + 2], nv, nmis);
        }
        for (; g < G; ++g)
            cluster_count(row[(int64_t)(g * 3) * sz], row[(int64_t)(g * 3 + 1) * sz], row[(int64_t)(g * 3 + 2) * sz], sh_q[g * 3],
                          sh_q[g * 3 + 1], sh_q[g * 3 + 2], nv, nmis);
        return a.seg_lut[(int64_t)nv * (nv + 1) / 2 + nmis];
    };
    // ---- accepted representatives: the survivors of the representative pass, in representative order
    const int64_t n_seg = a.rep_stride >> 6;
    const int32_t *cnt = a.seg_cnt + q * n_seg;
    const int32_t *sslot = a.seg_slot + q * a.rep_stride;
    int base = 0;
    bool overflow = false;
    if (PHASE == 4) {
        // (phase 4 wants the distance to EVERY representative -- the top-up rule walks on beyond the threshold -- and computes
        // them itself: same pair counts, same table, so the ones inside the threshold are the survivors the other phases read)
        __syncthreads();  // sh_q
        for (int64_t j0 = 0; j0 < a.n_reps; j0 += TPB) {
            const int64_t j = j0 + tid;
            // (scoredist: the full rows of the representative pass hold them)
            const double d = j < a.n_reps ? (sd ? a.rep_dist[q * a.rep_stride + j] : by_query(a.rep_panel + j, a.rep_stride)) : -1.0;
            if (j < a.n_reps) reprow[j] = d;
            const int in = j < a.n_reps && d >= 0 && d <= a.thr;
            int tot;
            const int at = base + block_excl_scan(in, sh_i, &tot);
            if (in && at < ACC_CAP) sh_rep[at] = (int)j;
            base += tot;
        }
    } else
    for (int64_t s0 = 0; s0 < n_seg; s0 += TPB) {
        const int64_t s = s0 + tid;
        const int my = s < n_seg ? cnt[s] : 0;
        int chunk_total;
        const int at = base + block_excl_scan_int<NW>(my, sh_i, &chunk_total);
        if (at + my <= ACC_CAP) {
            for (int k = 0; k < my; ++k) {
                const uint32_t pk = (uint32_t)sslot[s * 64 + k];
                sh_rep[at + k] = (int)(s * 64 + (pk >> 26));
            }
        } else if (my > 0) {
            overflow = true;
        }
        base += chunk_total;
    }
    if (tid == 0) sh_nacc = base;
    __syncthreads();
    const int n_acc = sh_nacc;
    if (n_acc > ACC_CAP || __syncthreads_or(overflow ? 1 : 0)) {
        // (once, in the first phase: the later ones just leave the query alone)
        if (PHASE == 1 && !LISTED && a.big_list) {  // the second (third) form of the phases takes it
            int at = SELECT_CLUSTERS_BIG_LIST;
            if (tid == 0) {
                at = atomicAdd(a.big_count, 1);
                if (at < SELECT_CLUSTERS_BIG_LIST) a.big_list[at] = (int32_t)q;
            }
            if (tid == 0 && at >= SELECT_CLUSTERS_BIG_LIST) to_slow(-1);
        } else if (PHASE <= 1) {
            to_slow(-1);
        }
        if (PHASE == 4) forward();
        return;
    }
    if (PHASE == 1) {  // one more query for every accepted cluster
        if (a.row_off) {  // (ragged rows: a flat member list longer than a small row takes a big one, here, before anything is counted)
            int mine = 0;
            for (int k = tid; k < n_acc; k += TPB) mine += a.rep_moff[sh_rep[k] + 1] - a.rep_moff[sh_rep[k]];
            const int m_all = block_sum<NW>(mine, sh_i);
            if (m_all > a.row_small) {
                if (tid == 0) sh_znode = row_make_big(a, q, true) ? 1 : 0;
                __syncthreads();
                if (!sh_znode) return;
            }
        }
        for (int k = tid; k < n_acc; k += TPB) atomicAdd(&a.cl_count[sh_rep[k]], 1);
        return;
    }
    // ---- members of the accepted clusters as one flat list: offsets by cluster
    {
        int carry = 0;
        for (int k0 = 0; k0 < n_acc; k0 += TPB) {
            const int k = k0 + tid;
            const int mb0 = k < n_acc ? a.rep_moff[sh_rep[k]] : 0;
            const int sz = k < n_acc ? a.rep_moff[sh_rep[k] + 1] - mb0 : 0;
            int tot;
            const int off = carry + block_excl_scan_int<NW>(sz, sh_i, &tot);
            if (k < n_acc) { sh_off[k] = off; sh_mb[k] = mb0; }
            carry += tot;
        }
        if (tid == 0) sh_off[n_acc] = carry;
    }
    __syncthreads();
    if (PHASE == 2) {  // the query joins the lists of its clusters (in whatever order the additions land: every pair is on its own)
        if (a.q_items) {  // clade blocks: which items are this query's, in the order of its accepted clusters
            if (tid == 0) { sh_znode = atomicAdd(a.q_item_cursor, n_acc); a.q_items[q] = make_int2(sh_znode, n_acc); }
            __syncthreads();
        }
        for (int k = tid; k < n_acc; k += TPB) {
            const int c = sh_rep[k];
            const int pos = atomicAdd(&a.cl_fill[c], 1), item = a.cl_start[c] + pos;
            a.cl_items[item] = make_int2((int)q, sh_off[k]);
            if (a.q_items) {
                // where the tuples of the cluster's blocks will be for this query: tile pos / 64 of the cluster, lane pos % 64
                // (k_cluster_tiles: 1 + ns + ceil(size / 6) slots per tile from the cluster's base; -1: no blocks, no room)
                const int bb = a.cl_bbase[c], ns = a.rep_soff[c + 1] - a.rep_soff[c];
                const int ts = 1 + ns + (a.rep_moff[c + 1] - a.rep_moff[c] + 5) / 6, tb = bb + (pos >> 6) * ts;
                a.q_item[sh_znode + k] = item;
                a.item_sbase[item] = (bb >= 0 && (int64_t)(tb + ts) * 384 <= a.blk_pool_cap) ? tb * 64 + (pos & 63) : -1;
                a.item_bad[item] = 0;
            }
        }
        return;
    }
    int M = sh_off[n_acc];
    int n_acc_all = n_acc;  // (phase 4: grows with the clusters the top-up rule accepts)
    bool use_blk = false;   // this query's observation list names block roots (workgroup-uniform)
    if (PHASE == 3 && blk_space) {
        const int2 qi = a.q_items[q];
        int any = 0;
        for (int k = tid; k < n_acc; k += TPB) {
            int sb = -1;
            if (k < qi.y) {
                const int item = a.q_item[qi.x + k];
                sb = a.item_bad[item] ? -1 : a.item_sbase[item];
            }
            sh_sb[k] = sb;
            any |= sb >= 0;
        }
        use_blk = __syncthreads_or(any) != 0;
    }
    // ---- clade blocks, the short form of this phase.  When every accepted cluster that has blocks has them for this query, the
    // observation list is a hundred-odd entries -- the clusters' block roots and their few members outside every block -- and needs
    // neither the bitmap over every emission index nor a look at every member: a thread per accepted cluster writes its entries
    // into a short list in LDS (the bitmap's memory: it is not used), the entries are ranked by counting, and the per-level offsets
    // are counts over the list.  Anything else -- a cluster whose item goes without blocks, an exact match or the query's own row
    // among the members outside the blocks, more than SHORT_CAP entries, fewer than two -- takes the general form below.
#ifndef SELECT_SHORT_ONLY_CAP
#define SELECT_SHORT_ONLY_CAP 1024  // entries of the short form's list in the SHORT_ONLY launch (experiments: 512 = eight workgroups per CU)
#endif
    constexpr int SHORT_CAP = SHORT_ONLY ? SELECT_SHORT_ONLY_CAP : 1024;
    if (PHASE == 3 && !LISTED && use_blk) {
        double *f_val = reinterpret_cast<double *>(dyn_bits);
        int *f_key = reinterpret_cast<int *>(f_val + SHORT_CAP), *f_node = f_key + SHORT_CAP, *f_off = f_node + SHORT_CAP;  // f_off[ACC_CAP + 1]
        // entries each cluster may write (its blocks + its members outside them); a cluster with blocks must have them here
        int ok = 1, carry = 0;
        for (int k0 = 0; k0 < n_acc; k0 += TPB) {
            const int k = k0 + tid;
            int cnt_k = 0;
            if (k < n_acc) {
                const int c = sh_rep[k];
                if (a.rep_soff[c + 1] > a.rep_soff[c] && sh_sb[k] < 0) ok = 0;
                cnt_k = a.rep_boff[c + 1] - a.rep_boff[c] + a.rep_loff[c + 1] - a.rep_loff[c];
            }
            int tot;
            const int off = carry + block_excl_scan_int<NW>(cnt_k, sh_i, &tot);
            if (k < n_acc) f_off[k] = off;
            carry += tot;
        }
        const int n_slots_f = carry;
        if (__syncthreads_and(ok) && n_slots_f <= SHORT_CAP) {
            int obs_c = 0, tot_c = 0, extra_c = 0, slow = 0;
            for (int k = tid; k < n_acc; k += TPB) {
                const int c = sh_rep[k], sb = sh_sb[k], mb = sh_mb[k];
                int at = f_off[k];
                const int l0 = a.rep_loff[c], l1 = a.rep_loff[c + 1];
                const int in_blocks = (sh_off[k + 1] - sh_off[k]) - (l1 - l0);  // (every one of them observed: k_cluster_dist found none to drop)
                obs_c += in_blocks;
                tot_c += in_blocks;
                for (int b = a.rep_boff[c]; b < a.rep_boff[c + 1]; ++b, ++at) {
                    f_key[at] = a.e_of_blk[b];
                    f_node[at] = a.blk_root[b];
                    const long long pl = ((long long)(sb >> 6) + 1 + a.blk_rslot[b]) * 384 + (sb & 63);
                    f_val[at] = __longlong_as_double((long long)APPLES_BLOCK_BOX | pl);
                    extra_c += a.blk_nodes[b];
                }
                for (int l = l0; l < l1; ++l, ++at) {  // a member outside every block: Reference.py:150, PoolQueryWorker.py:63-75 one by one
                    const int mp = a.loose_mp[l], slot = a.mem_slot[mb + mp];
                    const double d = tmp[sh_off[k] + mp];
                    f_key[at] = 0x7fffffff;  // (no entry unless it is observed below)
                    if (d < 0) continue;
                    ++obs_c;
                    if (slot == self) continue;
                    ++tot_c;
                    if (d == 0) { slow = 1; continue; }  // an exact match: the general form keeps that book
                    const int node = a.slot_node[slot];
                    if (node < 0) continue;
                    f_key[at] = a.e_of_slot[slot];
                    f_node[at] = node;
                    f_val[at] = d;
                }
            }
            const int obs_s = block_sum<NW>(obs_c, sh_i);
            const int tot_s = block_sum<NW>(tot_c, sh_i);
            const int extra_s = block_sum<NW>(extra_c, sh_i);
            int ne_c = 0;
            for (int i = tid; i < n_slots_f; i += TPB) ne_c += f_key[i] != 0x7fffffff;
            const int ne = block_sum<NW>(ne_c, sh_i);
            if (!__syncthreads_or(slow) && ne >= 2) {
                if (obs_s < a.baseobs) { to_slow(obs_s); return; }  // the reference would pop further clusters (Reference.py:146)
                // rank by counting (the keys are distinct emission indices), then the list in level order
                for (int i = tid; i < n_slots_f; i += TPB) {
                    const int key = f_key[i];
                    if (key == 0x7fffffff) continue;
                    int r = 0;
                    for (int j = 0; j < n_slots_f; ++j) r += f_key[j] < key;
                    o_node[r] = f_node[i];
                    o_dist[r] = f_val[i];
                }
                int32_t *cgf = a.cnt_gt ? a.cnt_gt + q * (int64_t)(a.height + 2) : nullptr;
                for (int i = tid; cgf && i < a.height + 2; i += TPB) {  // entries above level i - 1: emission indices below lvl_e[i]
                    const int lim = a.lvl_e[i];
                    int r = 0;
                    for (int j = 0; j < n_slots_f; ++j) r += f_key[j] < lim;
                    cgf[i] = r;
                }
                if (tid == 0) {
                    apples_placement p;
                    p.edge = 0; p.flags = 0; p.error = 0.0; p.distal = 0.0; p.pendant = 0.0; p.n_obs = tot_s; p.n_valid = extra_s;
                    int nee = ne;
                    if (tot_s <= 2) {  // (PoolQueryWorker.py:97-98; cannot be with two entries of which one is a block, kept for the form's sake)
                        p.flags = APPLES_F_INSUFFICIENT | APPLES_F_PENDANT_INT;
                        p.edge = -1;
                        p.n_valid = 0;
                        nee = 0;
                    }
                    a.q_blk[q] = nee > 0 ? 1 : 0;
                    a.out[q] = p;
                    a.n_obs[q] = nee;
                    enlist(a, q, nee);
                }
                return;
            }
        }
        // the general form after all: the bitmap's memory as it expects it (the list's offsets, at least, were written into it)
        if (!SHORT_ONLY) {
            __syncthreads();
            for (int i = tid; i < n_words; i += TPB) dyn_bits[i] = 0;
            __syncthreads();
        }
    }
    if (SHORT_ONLY) {  // (this launch has no bitmap: the LISTED launch behind it takes the query)
        if (tid == 0) a.gen_list[atomicAdd(a.gen_count, 1)] = (int32_t)q;
        return;
    }
    if (PHASE == 4) {
        // ---- the members' distances (a thread per member), then Reference.py:144-152: while fewer than `-b` valid member
        // distances are in, the representative with the next smallest (distance, index) beyond the threshold brings its cluster
        int c = 0;
        for (int m = tid; m < M; m += TPB) {
            int lo = 0, hi = n_acc;
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (sh_off[mid] <= m) lo = mid; else hi = mid;
            }
            const double d = sd ? sd_dist(a.mem_slot[sh_mb[lo] + (m - sh_off[lo])])
                                : by_query(a.packed_rm + (int64_t)sh_mb[lo] * (G * 3) + (m - sh_off[lo]), sh_off[lo + 1] - sh_off[lo]);
            tmp[m] = d;
            c += !(d < 0);
        }
        int obs = block_sum<NW>(c, sh_i);
        double cut_d = -INF_D;
        int cut_i = -1;
        while (obs < a.baseobs) {
            double bd = INF_D;
            int bi = 0x7fffffff, bj = 0;
            for (int64_t j = tid; j < a.n_reps; j += TPB) {
                const double d = reprow[j];
                if (d > a.thr && key_lt(cut_d, cut_i, d, (int)j) && key_lt(d, (int)j, bd, bi)) { bd = d; bi = (int)j; }
            }
            block_argmin3<NW>(bd, bi, bj, sh_d, sh_i, sh_j);
            if (bi == 0x7fffffff) break;  // nothing left
            cut_d = bd;
            cut_i = bi;
            if (n_acc_all == ACC_CAP) { forward(); return; }
            const int mb0 = a.rep_moff[bi], sz = a.rep_moff[bi + 1] - mb0;
            __syncthreads();  // (the searches above are done with sh_off)
            if (tid == 0) { sh_rep[n_acc_all] = bi; sh_mb[n_acc_all] = mb0; sh_off[n_acc_all + 1] = M + sz; }
            c = 0;
            for (int mp = tid; mp < sz; mp += TPB) {
                const double d = sd ? sd_dist(a.mem_slot[mb0 + mp]) : by_query(a.packed_rm + (int64_t)mb0 * (G * 3) + mp, sz);
                tmp[M + mp] = d;
                c += !(d < 0);
            }
            obs += block_sum<NW>(c, sh_i);
            M += sz;
            ++n_acc_all;
        }
        __syncthreads();  // tmp and the list are complete
    }
    // ---- pass 1: one (query, member) distance per thread; E members per thread and round, so that the lookups of a round
    // (member -> slot -> node) are E in flight and not one after the other: the workgroup's time is their latency
    constexpr int E = PHASE == 0 ? 1 : 4;
    auto cluster_of = [&](int m) -> int {  // last accepted cluster whose offset <= m
        int lo = 0, hi = n_acc_all;
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (sh_off[mid] <= m) lo = mid; else hi = mid;
        }
        return lo;
    };
    int n_total = 0, obs_cnt = 0, blk_extra = 0;
    double z_d = INF_D;
    int z_i = 0x7fffffff, z_p = 0x7fffffff, z_node = -2;
    for (int m0 = 0; m0 < M; m0 += TPB * E) {
        int lo_[E], slot_[E], node_[E], mbk_[E];
        double d_[E];
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int m = m0 + e * TPB + tid;
            lo_[e] = -1; slot_[e] = 0; d_[e] = -1.0; mbk_[e] = -1;
            if (m < M) {
                const int lo = cluster_of(m), mp = m - sh_off[lo], mb = sh_mb[lo];
                lo_[e] = lo;
                // clade blocks: a member of a block whose tuples k_blocks_up has formed for this query -- every leaf of the block is
                // a tree leaf with a distance > 0 and none is the query's own row (else the cluster's item has no blocks): it is
                // counted, its block's first leaf marks the root, and nothing of its own is looked up
                if (PHASE == 3 && use_blk && sh_sb[lo] >= 0) mbk_[e] = a.mem_block[mb + mp];
                if (mbk_[e] >= 0) { d_[e] = 1.0; continue; }
                slot_[e] = a.mem_slot[mb + mp];  // (on its way with the member's distance / words below: neither waits for the other)
                if (PHASE >= 3) d_[e] = tmp[m];  // k_cluster_dist (phase 4: the block above) left it there
                else d_[e] = by_query(a.packed_rm + (int64_t)mb * (G * 3) + mp, sh_off[lo + 1] - sh_off[lo]);
            }
        }
#pragma unroll
        for (int e = 0; e < E; ++e)  // (the node of every listed member: cheaper asked for than waited for)
            node_[e] = lo_[e] < 0 ? -1 : (mbk_[e] >= 0 ? 0 : a.slot_node[slot_[e]]);
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int m = m0 + e * TPB + tid;
            if (lo_[e] < 0) continue;
            const int lo = lo_[e], slot = slot_[e];
            const double d = d_[e];
            double keep = -2.0;  // not emitted
            if (!(d < 0)) {      // Reference.py:150: `if not dm < 0`
                ++obs_cnt;
                if (slot != self || mbk_[e] >= 0) {
                    ++n_total;
                    const int node = node_[e];
                    if (d == 0) {
                        // (rare: the representative's own distance is looked up again among the survivors of its segment)
                        const int rep = sh_rep[lo], mp = m - sh_off[lo];
                        double drep = PHASE == 4 ? reprow[rep] : (a.rep_dist ? a.rep_dist[q * a.rep_stride + rep] : 0.0);
                        for (int k = 0; PHASE != 4 && !a.rep_dist && k < cnt[rep >> 6]; ++k) {
                            const uint32_t pk = (uint32_t)sslot[(int64_t)(rep >> 6) * 64 + k];
                            if ((int)(pk >> 26) == (rep & 63)) {
                                const long long valid = (pk >> 13) & 0x1fffu, mism = pk & 0x1fffu;
                                drep = a.seg_lut[valid * (valid + 1) / 2 + mism];
                            }
                        }
                        if (drep < z_d || (drep == z_d && (rep < z_i || (rep == z_i && mp < z_p)))) {
                            z_d = drep; z_i = rep; z_p = mp; z_node = node;
                        }
                    }
                    if (node >= 0) {
                        keep = d;
                        if (mbk_[e] >= 0) {  // the block's root stands for its leaves: its first leaf marks it
                            if (mbk_[e] & 1) {
                                const int es = a.e_of_blk[mbk_[e] >> 1];
                                atomicOr(&dyn_bits[es >> 6], 1ull << (es & 63));
                            }
                        } else {
                            const int es = blk_space ? a.e_of_slot[slot] : slot;
                            atomicOr(&dyn_bits[es >> 6], 1ull << (es & 63));
                        }
                    }
                }
            }
            if (mbk_[e] < 0) tmp[m] = keep;  // (a block's leaf: its distance stays as it is)
        }
    }
    const int obs = block_sum<NW>(obs_cnt, sh_i);
    if (PHASE != 4 && obs < a.baseobs) { to_slow(obs); return; }  // the reference would pop further clusters (Reference.py:146): phase 4 did
    // ---- ranks of the slot bitmap: a thread counts its run of 16 words (of 32 beyond 16 TPB words: 262 144 slots at 256
    // threads), one scan over the threads
    int n_emit;
    const int rsh = n_words > TPB * 16 ? 5 : 4;
    auto ranks = [&]() {
        int c = 0;
        for (int k = 0; k < (1 << rsh); ++k) {
            const int w = (tid << rsh) + k;
            if (w < n_words) { pre[w] = (uint16_t)c; c += __popcll(dyn_bits[w]); }
        }
        sh_base[tid] = block_excl_scan_int<NW>(c, sh_i, &n_emit);
        __syncthreads();
    };
    ranks();
    if (PHASE == 3 && use_blk && n_emit < 2) {
        // every observed leaf inside one block (or nothing left to observe): the block's root would be the only entry of the list
        // and the induced subtree's root lies inside the block -- this query goes without blocks, its leaves one by one
        use_blk = false;
        for (int i = tid; i < n_words; i += TPB) dyn_bits[i] = 0;
        __syncthreads();
        for (int m = tid; m < M; m += TPB) {
            if (!(tmp[m] >= 0)) continue;  // (pass 1 left the distance of what it keeps, -2 elsewhere)
            const int lo = cluster_of(m);
            const int es = a.e_of_slot[a.mem_slot[sh_mb[lo] + (m - sh_off[lo])]];
            atomicOr(&dyn_bits[es >> 6], 1ull << (es & 63));
        }
        __syncthreads();
        ranks();
    }
    auto rank_of = [&](int slot) -> int {  // emitted members in the slots below `slot`
        const int w = slot >> 6;
        if (w >= n_words) return n_emit;
        return sh_base[w >> rsh] + (int)pre[w] + __popcll(dyn_bits[w] & ((1ull << (slot & 63)) - 1ull));
    };
    // ---- pass 2: emission in slot order
    for (int m0 = 0; m0 < M; m0 += TPB * E) {
        int slot_[E], mbk_[E], lo_[E];
        double d_[E];
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int m = m0 + e * TPB + tid;
            slot_[e] = -1; d_[e] = -2.0; mbk_[e] = -1; lo_[e] = 0;
            if (m < M) {
                const int lo = cluster_of(m);
                lo_[e] = lo;
                if (PHASE == 3 && use_blk && sh_sb[lo] >= 0) mbk_[e] = a.mem_block[sh_mb[lo] + (m - sh_off[lo])];
                if (mbk_[e] >= 0) { d_[e] = (mbk_[e] & 1) ? 1.0 : -2.0; continue; }  // (the block's first leaf writes the block's entry)
                d_[e] = tmp[m];
                slot_[e] = a.mem_slot[sh_mb[lo] + (m - sh_off[lo])];
            }
        }
        int node_[E];
#pragma unroll
        for (int e = 0; e < E; ++e) node_[e] = (slot_[e] >= 0 && mbk_[e] < 0) ? a.slot_node[slot_[e]] : -1;
#pragma unroll
        for (int e = 0; e < E; ++e)
            if (d_[e] >= 0) {
                int es, node = node_[e];
                double dd = d_[e];
                if (mbk_[e] >= 0) {
                    if (!(mbk_[e] & 1)) continue;  // (the block's first leaf writes the block's entry)
                    const int b = mbk_[e] >> 1, sb = sh_sb[lo_[e]];
                    es = a.e_of_blk[b];
                    node = a.blk_root[b];
                    // the "distance" of a block root: where its tuple is (sweep_lean.hip:lean_is_block), a boxed index into the pool
                    const long long at = ((long long)(sb >> 6) + 1 + a.blk_rslot[b]) * 384 + (sb & 63);
                    dd = __longlong_as_double((long long)APPLES_BLOCK_BOX | at);
                    blk_extra += a.blk_nodes[b];  // (the nodes below the root, once)
                } else {
                    es = blk_space ? a.e_of_slot[slot_[e]] : slot_[e];
                }
                const int pos = rank_of(es);
                o_node[pos] = node;
                o_dist[pos] = dd;
            }
    }
    n_total = block_sum<NW>(n_total, sh_i);  // (its barriers also publish the emission)
    if (PHASE == 3 && blk_space) blk_extra = use_blk ? block_sum<NW>(blk_extra, sh_i) : 0;
    double zd = z_d; int zi = z_i, zp = z_p;
    block_argmin3<NW>(zd, zi, zp, sh_d, sh_i, sh_j);
    if (tid == 0) sh_znode = -2;
    __syncthreads();
    if (z_node != -2 && z_d == zd && z_i == zi && z_p == zp) sh_znode = z_node;
    __syncthreads();
    // per-level offsets into the level-sorted list (the sweep's cnt_gt): cg[l + 1] = entries above level l = the emitted members
    // in the slots above level l (slots are sorted by level, deepest first; lvl_slots[l + 1] = how many slots those are)
    int32_t *cg = a.cnt_gt ? a.cnt_gt + q * (int64_t)(a.height + 2) : nullptr;
    if (cg && blk_space) {
        for (int i = tid; i < a.height + 2; i += TPB) cg[i] = rank_of(a.lvl_e[i]);
    } else if (cg && a.lvl_slots) {
        for (int i = tid; i < a.height + 2; i += TPB) cg[i] = rank_of(a.lvl_slots[i]);
    } else {
        for (int i = tid; cg && i <= n_emit; i += TPB) {
            const int lv = (i < n_emit) ? a.node_level[o_node[i]] : -1;
            const int lprev = (i == 0) ? a.height + 1 : a.node_level[o_node[i - 1]];
            for (int l = lv; l < lprev; ++l) cg[l + 1] = i;
        }
    }
    if (tid == 0) {
        apples_placement p;
        p.edge = 0; p.flags = 0; p.error = 0.0; p.distal = 0.0; p.pendant = 0.0; p.n_obs = n_total; p.n_valid = 0;
        int ne = n_emit;
        if (zi != 0x7fffffff) {
            p.flags = APPLES_F_EXACT | APPLES_F_PENDANT_INT;
            p.edge = sh_znode;
            if (sh_znode < 0) { p.flags |= APPLES_F_ZERO_NOT_IN_TREE; p.edge = -1; }
            ne = 0;
        } else if (n_total <= 2) {
            p.flags = APPLES_F_INSUFFICIENT | APPLES_F_PENDANT_INT;
            p.edge = -1;
            ne = 0;
        } else if (ne < 2) {
            p.flags = APPLES_F_DEGENERATE | APPLES_F_PENDANT_INT;
            p.edge = -1;
            ne = 0;
        }
        // (n_valid: the nodes strictly inside the emitted blocks; the sweep adds its own count to it)
        if (PHASE == 3 && blk_space && ne > 0) { p.n_valid = blk_extra; a.q_blk[q] = use_blk ? 1 : 0; }
        a.out[q] = p;
        a.n_obs[q] = ne;
        enlist(a, q, ne);
    }
}

// Queries per tile of a cluster with `sz` members (k_cluster_dist): a workgroup's 256 threads are P = min(sz, 256) member
// lanes x QL = 256 / P query lanes, and a thread keeps up to 16 queries' counts in registers.
__device__ __forceinline__ int cluster_tile_queries(int sz) {
    const int P = sz < APPLES_TPB ? (sz > 0 ? sz : 1) : APPLES_TPB, QL = APPLES_TPB / P;
    return QL * 16 < 64 ? QL * 16 : 64;
}

// One workgroup: the clusters' query counts -> list offsets (cl_start), cleared fill cursors, and the tile table.
#define CL_TILES_TPB 1024  // (one workgroup walks all clusters: its time is rounds of two scans, so make the rounds few)
__global__ __launch_bounds__(CL_TILES_TPB) void k_cluster_tiles(SelectArgs a) {
    __shared__ int sh_i[CL_TILES_TPB / WAVE];
    const int tid = threadIdx.x;
    int item_base = 0, tile_base = 0, btile_base = 0, bslot_base = 0;
    for (int64_t c0 = 0; c0 < a.n_reps; c0 += CL_TILES_TPB) {
        // (clade blocks: the clusters in the order of their walks' lengths, longest first -- the tiles of the block kernels are
        // taken in table order, and a launch's floor is its longest walk: started first it lies beside the others, not behind them)
        const int64_t c = c0 + tid < a.n_reps ? (a.cl_order ? a.cl_order[c0 + tid] : c0 + tid) : a.n_reps;
        const int cnt = c < a.n_reps ? a.cl_count[c] : 0;
        const int T = c < a.n_reps ? (a.cl_mfma ? 64 : cluster_tile_queries(a.rep_moff[c + 1] - a.rep_moff[c])) : 1;  // (k_cluster_dist_mfma: 64 queries)
        const int nt = (cnt + T - 1) / T;
        int tot_i, tot_t;
        const int at_i = item_base + block_excl_scan_int<CL_TILES_TPB / WAVE>(cnt, sh_i, &tot_i);
        const int at_t = tile_base + block_excl_scan_int<CL_TILES_TPB / WAVE>(nt, sh_i, &tot_t);
        if (c < a.n_reps) {
            a.cl_start[c] = at_i;
            a.cl_fill[c] = 0;
            for (int k = 0; k < nt; ++k)
                if (at_t + k < a.cl_tiles_cap) a.cl_tiles[at_t + k] = make_int4((int)c, at_i + k * T, cnt - k * T < T ? cnt - k * T : T, 0);
        }
        item_base += tot_i;
        tile_base += tot_t;
        if (a.blk_tiles) {
            // clade blocks: the cluster's items once more in tiles of up to 64 (a lane of k_blocks_up / k_blocks_down = an item), each
            // with room in the pool for the tuples of the cluster's block-internal nodes (ns slots of 6 x 64 doubles); a tile the
            // pool cannot hold gets no storage (base -1: its items go without blocks)
            const int ns = c < a.n_reps ? a.rep_soff[c + 1] - a.rep_soff[c] : 0;
            const int nb = ns > 0 ? (cnt + 63) / 64 : 0;
            int tot_b, tot_s;
            const int at_b = btile_base + block_excl_scan_int<CL_TILES_TPB / WAVE>(nb, sh_i, &tot_b);
            // (slots per tile: the first holds the lanes' best edges inside the blocks, k_blocks_down -> k_blocks_finish; ns for the
            // tuples; then the members' distances once more as [member][lane], six rows to a slot: what the walks read whole rows of)
            const int ts = c < a.n_reps ? 1 + ns + (a.rep_moff[c + 1] - a.rep_moff[c] + 5) / 6 : 0;
            const int at_s = bslot_base + block_excl_scan_int<CL_TILES_TPB / WAVE>(nb * ts, sh_i, &tot_s);
            if (c < a.n_reps) a.cl_bbase[c] = ns > 0 ? at_s : -1;  // (phase 2 derives every item's place from it)
            for (int k = 0; k < nb; ++k)
                if (at_b + k < a.blk_tiles_cap)
                    a.blk_tiles[at_b + k] = make_int4((int)c, at_i + k * 64, cnt - k * 64 < 64 ? cnt - k * 64 : 64,
                                                      (int64_t)(at_s + (k + 1) * ts) * 384 <= a.blk_pool_cap ? at_s + k * ts : -1);
            btile_base += tot_b;
            bslot_base += tot_s;
        }
    }
    if (tid == 0) {
        a.cl_start[a.n_reps] = item_base;
        *a.cl_ntiles = tile_base < a.cl_tiles_cap ? tile_base : (int)a.cl_tiles_cap;  // (the cap is the proven upper bound: never hit)
        if (a.blk_tiles) *a.blk_ntiles = btile_base < a.blk_tiles_cap ? btile_base : (int)a.blk_tiles_cap;
    }
}

// The member distances of one tile = one cluster x up to 64 of the queries that accepted it.  Thread = (member lane,
// query lane); per chunk of CL_GC word groups the member's words are loaded once into registers (lanes along the members:
// consecutive 16 bytes of the cluster-major panel), the tile's query words are staged in LDS (a wavefront reads one or a few
// addresses: broadcasts), and the thread adds the pair counts of its up to 16 queries.  Same counts and same table lookup as
// phase 0 of k_select_clusters, so the same bits.  Persistent workgroups walk the tile table.
#define CL_GC 4
__global__ __launch_bounds__(APPLES_TPB) void k_cluster_dist(SelectArgs a) {
    __shared__ uint4 sh_qw[64][CL_GC * 3];
    __shared__ int sh_q[64], sh_o[64];
    const int tid = threadIdx.x;
    const int G = a.G;
    const int n_tiles = *a.cl_ntiles;
    for (int t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const int4 tile = a.cl_tiles[t];
        const int c = tile.x, nqt = tile.z;
        const int mb = a.rep_moff[c], sz = a.rep_moff[c + 1] - mb;
        __syncthreads();  // the previous tile's last readers of sh_q / sh_o / sh_qw
        if (tid < nqt) {
            const int2 it = a.cl_items[tile.y + tid];
            sh_q[tid] = it.x; sh_o[tid] = it.y;
        }
        for (int mc0 = 0; mc0 < sz; mc0 += APPLES_TPB) {  // (clusters beyond 256 members: in chunks)
            const int P = sz - mc0 < APPLES_TPB ? sz - mc0 : APPLES_TPB, QL = APPLES_TPB / P;
            const int ml = tid % P, jl = tid / P;
            const bool active = jl < QL;
            uint32_t nv[16], nmis[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) { nv[k] = 0; nmis[k] = 0; }
            const uint4 *row = a.packed_rm + (int64_t)mb * (G * 3) + mc0 + ml;
            for (int g0 = 0; g0 < G; g0 += CL_GC) {
                __syncthreads();  // the list above is written / the previous chunk's words are read
                for (int i = tid; i < nqt * (CL_GC * 3); i += APPLES_TPB) {
                    const int j = i / (CL_GC * 3), w = i % (CL_GC * 3), g = g0 + w / 3, pl = w % 3;
                    const int64_t q = sh_q[j];
                    sh_qw[j][w] = g < G ? a.qpacked[(((q >> 4) * G + g) * 16 + (q & 15)) * 3 + pl] : make_uint4(0, 0, 0, 0);
                }
                uint4 mw[CL_GC * 3];
                if (active) {
#pragma unroll
                    for (int w = 0; w < CL_GC * 3; ++w)
                        mw[w] = g0 + w / 3 < G ? row[(int64_t)((g0 + w / 3) * 3 + w % 3) * sz] : make_uint4(0, 0, 0, 0);
                }
                __syncthreads();
                if (active) {
#pragma unroll
                    for (int k = 0; k < 16; ++k) {
                        const int j = jl + k * QL;
                        if (j < nqt) {
#pragma unroll
                            for (int gg = 0; gg < CL_GC; ++gg)
                                cluster_count(mw[gg * 3], mw[gg * 3 + 1], mw[gg * 3 + 2], sh_qw[j][gg * 3], sh_qw[j][gg * 3 + 1],
                                              sh_qw[j][gg * 3 + 2], nv[k], nmis[k]);
                        }
                    }
                }
            }
            if (active) {
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    const int j = jl + k * QL;
#ifdef CL_EXP_NO_EPILOGUE
                    if (j < nqt && nv[k] == 0x7fffffffu) {
#else
                    if (j < nqt) {
#endif
                        const double d = a.seg_lut[(int64_t)nv[k] * (nv[k] + 1) / 2 + nmis[k]];
                        a.tmp_d[row_start(a.row_off, sh_q[j], a.stride) + sh_o[j] + mc0 + ml] = d;
                        // clade blocks: a member the reference drops, an exact match or the query's own row -- the item goes without
                        if (a.item_bad && (!(d > 0) || (a.self_slot && a.mem_slot[mb + mc0 + ml] == a.self_slot[sh_q[j]]))) a.item_bad[tile.y + j] = 1;
                    }
                }
            }
        }
    }
}

// The same on the matrix cores (the default): a wavefront = a tile of up to 64 queries x 64 of the cluster's members at a time, the
// pair counts as in k_jc69_mfma (dist.hip: acc1 = sum t.t over three components, acc2 = valid; mism = (3 valid - acc1) / 4, exact
// in the f32 accumulators) -- four v_mfma_f32_32x32x64_f8f6f4 per 32 x 32 block of pairs and 64-site block where k_cluster_dist
// spends ~ 240 vector instructions per pair and keeps the vector pipes 70 % busy (profiles/r05_blk_sq_counters.txt).  Both
// operands come as bit planes (3 bits per site: the queries' packed rows, the cluster-major member panel; lanes along the rows) and
// are expanded to fp4 nibbles into the wavefront's own LDS images (expand_quarter, the same encoding on both sides; no workgroup
// barrier: a workgroup is one wavefront).  (Fed
// from the queries' pre-expanded fp4 images instead -- 16 bits per site -- the kernel was bound by L2 traffic and slower than the
// bit counts: 3.9 against 1.3 ms per batch.)  A 32 x 32 block with no query or no member in it is skipped.  Same counts, same table
// look-up: the same bits as k_cluster_dist.
#ifndef CLM_WAVES
#define CLM_WAVES 2  // wavefronts per SIMD k_cluster_dist_mfma is compiled for
#endif
__global__ __launch_bounds__(WAVE, CLM_WAVES) void k_cluster_dist_mfma(SelectArgs a) {
    // the images of one 64-site block, piece-major: piece (component c, quarter k) of row r -- 8 bytes, 16 sites -- at [c * 4 + k][r].  A lane
    // expands its own row (it holds the row's plane words), so a store instruction writes 64 consecutive pieces and a fragment read
    // 32 consecutive ones: no bank conflicts either way (row-major images as in k_jc69_mfma cost this kernel four-way conflicts on
    // every store of the expansion, and the LDS was what it waited for)
    // Only t1 and t2 go through LDS: t3 is t1 with the sign flipped where t2 is negative and the validity operand is t1 with the sign
    // bits cleared -- two bit operations per fragment register (as dist_gemm.hip), half the LDS traffic, which is what bounds the loop
    __shared__ uint2 Aq[8][64];  // the tile's queries
    __shared__ uint2 Bm[8][64];  // 64 members
    __shared__ int sh_q[64], sh_o[64], sh_self[64];
    const int lane = threadIdx.x;
    const int G = a.G;
    const int n_tiles = *a.cl_ntiles;
    auto expand_row = [&](uint2 (*img)[64], uint32_t m, uint32_t c0, uint32_t c1, int half) {  // one 32-site word: quarters 2 half, 2 half + 1 (expand_quarter's arithmetic)
        const uint32_t K2 = 0x22222222u, K8 = 0x88888888u;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const uint32_t mm = m >> (2 * k), a0 = c0 >> (2 * k), a1 = c1 >> (2 * k);
            const uint32_t v0 = (mm << 1) & K2, v1 = mm & K2;
            const int q = 2 * half + k;
            img[0 + q][lane] = make_uint2(((a1 << 3) & K8) | v0, ((a1 << 2) & K8) | v1);
            img[4 + q][lane] = make_uint2(((a0 << 3) & K8) | v0, ((a0 << 2) & K8) | v1);
        }
    };
    const int fr = lane & 31, fh = lane >> 5;
    for (int t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const int4 tile = a.cl_tiles[t];
        const int c = tile.x, nqt = tile.z;
        const int mb = a.rep_moff[c], sz = a.rep_moff[c + 1] - mb;
        __builtin_amdgcn_wave_barrier();  // (the previous tile's readers of sh_q / sh_o)
        {
            const int2 it = a.cl_items[tile.y + (lane < nqt ? lane : 0)];
            sh_q[lane] = it.x; sh_o[lane] = it.y;
            sh_self[lane] = a.self_slot ? a.self_slot[it.x] : -2;  // (-2: no member's slot; my_slot is -1 without the check)
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        const int64_t myq = sh_q[lane];
        const uint4 *qsrc = a.qpacked + ((myq >> 4) * G * 16 + (myq & 15)) * 3;  // group g, plane pl at ((g * 16) * 3 + pl) beyond this
        const bool qi1 = nqt > 32;
        for (int mc0 = 0; mc0 < sz; mc0 += 64) {
            const int msz = sz - mc0 < 64 ? sz - mc0 : 64;
            const bool mj1 = msz > 32;
            const int ml = lane < msz ? lane : msz - 1;  // (lanes beyond the chunk repeat its last member: computed, not written)
            const uint4 *rsrc = a.packed_rm + (int64_t)mb * (G * 3) + mc0 + ml;
            v16f_t s1[2][2], s2[2][2];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int x = 0; x < 16; ++x) { s1[i][j][x] = 0.f; s2[i][j][x] = 0.f; }
            uint4 qm, q0, q1, pm, p0, p1;
            auto fetch = [&](int g, uint4 &am, uint4 &a0, uint4 &a1, uint4 &bm, uint4 &b0, uint4 &b1) {
                const uint4 *qp = qsrc + (int64_t)g * 48;
                am = qp[0]; a0 = qp[1]; a1 = qp[2];
                const uint4 *rp = rsrc + (int64_t)(g * 3) * sz;
                bm = rp[0]; b0 = rp[sz]; b1 = rp[2 * (int64_t)sz];
            };
            fetch(0, qm, q0, q1, pm, p0, p1);
            for (int g = 0; g < G; ++g) {
#pragma unroll
                for (int y = 0; y < 2; ++y) {
                    // planes -> the images (the previous block's fragment reads are behind us: one wavefront, program order)
                    expand_row(Aq, y ? qm.z : qm.x, y ? q0.z : q0.x, y ? q1.z : q1.x, 0);
                    expand_row(Aq, y ? qm.w : qm.y, y ? q0.w : q0.y, y ? q1.w : q1.y, 1);
                    expand_row(Bm, y ? pm.z : pm.x, y ? p0.z : p0.x, y ? p1.z : p1.x, 0);
                    expand_row(Bm, y ? pm.w : pm.y, y ? p0.w : p0.y, y ? p1.w : p1.y, 1);
                    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                    {
                        v4i_t fa[2][2], fb[2][2];  // [t1 | t2][row block]
#pragma unroll
                        for (int cc = 0; cc < 2; ++cc)
#pragma unroll
                            for (int i = 0; i < 2; ++i) {  // sites 32 fh .. 32 fh + 31 of the block: pieces 2 fh and 2 fh + 1
                                const uint2 a0 = Aq[cc * 4 + 2 * fh][i * 32 + fr], a1 = Aq[cc * 4 + 2 * fh + 1][i * 32 + fr];
                                const uint2 b0 = Bm[cc * 4 + 2 * fh][i * 32 + fr], b1 = Bm[cc * 4 + 2 * fh + 1][i * 32 + fr];
                                fa[cc][i] = v4i_t{(int)a0.x, (int)a0.y, (int)a1.x, (int)a1.y};
                                fb[cc][i] = v4i_t{(int)b0.x, (int)b0.y, (int)b1.x, (int)b1.y};
                            }
#pragma unroll
                        for (int cc = 0; cc < 4; ++cc) {  // t1, t2, then t3 and v in t2's and t1's registers
                            if (cc == 2) {
#pragma unroll
                                for (int i = 0; i < 2; ++i)
#pragma unroll
                                    for (int e = 0; e < 4; ++e) {
                                        fa[1][i][e] = fa[0][i][e] ^ (fa[1][i][e] & (int)0x88888888u);
                                        fb[1][i][e] = fb[0][i][e] ^ (fb[1][i][e] & (int)0x88888888u);
                                    }
                            }
                            if (cc == 3) {
#pragma unroll
                                for (int i = 0; i < 2; ++i)
#pragma unroll
                                    for (int e = 0; e < 4; ++e) { fa[0][i][e] &= 0x77777777; fb[0][i][e] &= 0x77777777; }
                            }
                            const int set = (cc == 0 || cc == 3) ? 0 : 1;
#pragma unroll
                            for (int i = 0; i < 2; ++i)
#pragma unroll
                                for (int j = 0; j < 2; ++j) {
                                    if ((i == 1 && !qi1) || (j == 1 && !mj1)) continue;  // (wave-uniform)
                                    if (cc < 3) s1[i][j] = mfma_f4(fa[set][i], fb[set][j], s1[i][j]);
                                    else s2[i][j] = mfma_f4(fa[set][i], fb[set][j], s2[i][j]);
                                }
                        }
                    }
                    __builtin_amdgcn_wave_barrier();
                }
                // (the next group's plane words, once this group's are expanded.  A second set requested a group ahead was measured:
                // 0.95 against 0.88 ms per launch -- its 24 registers spill at two wavefronts per SIMD; one wavefront per SIMD without
                // spills 1.17, three 2.59: profiles/r05_cluster_dist_exp.txt)
                if (g + 1 < G) fetch(g + 1, qm, q0, q1, pm, p0, p1);
            }
            // C layout of the 32 x 32 blocks: column (member) = lane & 31, row (query) = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5).
            // A block's 16 table look-ups per lane leave together, then its 16 stores (a look-up behind a store would wait for it:
            // the compiler cannot tell the table from the rows)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if ((i == 1 && !qi1) || (j == 1 && !mj1)) continue;
                    const int m = j * 32 + fr;
                    const bool m_in = m < msz;
                    const int my_slot = (a.item_bad && a.self_slot) ? a.mem_slot[mb + mc0 + (m_in ? m : 0)] : -1;
                    double d[16];
#pragma unroll
                    for (int x = 0; x < 16; ++x) {
                        const int nv = (int)s2[i][j][x], nmis = (3 * nv - (int)s1[i][j][x]) >> 2;
#ifdef CL_EXP_NO_EPILOGUE  // (timing experiment: the main loop alone)
                        d[x] = (double)(nv + nmis);
#else
                        d[x] = a.seg_lut[(int64_t)nv * (nv + 1) / 2 + nmis];
#endif
                    }
#pragma unroll
                    for (int x = 0; x < 16; ++x) {
                        const int jq = i * 32 + (x & 3) + 8 * (x >> 2) + 4 * fh;
#ifdef CL_EXP_NO_EPILOGUE
                        if (jq < nqt && m_in && d[x] == -12345.0) {
#else
                        if (jq < nqt && m_in) {
#endif
                            a.tmp_d[row_start(a.row_off, sh_q[jq], a.stride) + sh_o[jq] + mc0 + m] = d[x];
                            // clade blocks: a member the reference drops, an exact match or the query's own row -- the item goes without
                            if (a.item_bad && (!(d[x] > 0) || my_slot == sh_self[jq])) a.item_bad[tile.y + jq] = 1;
                        }
                    }
                }
        }
    }
}

// The same tiles for scoredist contexts (-p with clusters): the exact distance of every (member, query) pair of a tile with the
// arithmetic of k_scoredist (dist.hip: fp64, sites left to right, the 21 x 21 table in LDS -- the same bits).  A wavefront takes
// 64 members of the cluster x TQ of the tile's queries at a time: the queries are wave-uniform (their residues select table
// rows by scalar arithmetic, all lanes of a look-up index one 21-entry row: conflict-free), a member's 16 residues of a site
// block are unpacked once for the TQ queries.
template <int TQ>
__global__ __launch_bounds__(APPLES_TPB) void k_cluster_dist_sd(SelectArgs a) {
    __shared__ double T[21 * 21];
    __shared__ int sh_q[64], sh_o[64];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    for (int i = tid; i < 21 * 21; i += APPLES_TPB) T[i] = a.table[i];
    const char *Tb = reinterpret_cast<const char *>(T);
    const int n16 = a.Lpad / 16;
    const int n_tiles = *a.cl_ntiles;
    for (int t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const int4 tile = a.cl_tiles[t];
        const int c = tile.x, nqt = tile.z;
        const int mb = a.rep_moff[c], sz = a.rep_moff[c + 1] - mb;
        __syncthreads();  // the previous tile's last readers of sh_q / sh_o (first tile: T is written)
        if (tid < nqt) {
            const int2 it = a.cl_items[tile.y + tid];
            sh_q[tid] = it.x; sh_o[tid] = it.y;
        }
        __syncthreads();
        const int MC = (sz + 63) >> 6, QG = (nqt + TQ - 1) / TQ;
        for (int item = wv; item < MC * QG; item += APPLES_TPB / 64) {  // (wave-uniform)
            const int mc = item % MC, qg = item / MC;
            const int m = mc * 64 + lane;
            const bool act = m < sz;
            const int64_t slot = a.mem_slot[mb + (act ? m : 0)];
            int64_t qi[TQ];
#pragma unroll
            for (int k = 0; k < TQ; ++k) qi[k] = __builtin_amdgcn_readfirstlane(sh_q[qg * TQ + k < nqt ? qg * TQ + k : qg * TQ]);
            double tot[TQ];
            uint32_t nv[TQ];
#pragma unroll
            for (int k = 0; k < TQ; ++k) { tot[k] = 0.0; nv[k] = 0; }
            // the member's rows: from the cluster-major copy (consecutive lanes = consecutive 16-byte words) where the context has one
            const uint8_t *ridx = a.aa_cm_idx ? a.aa_cm_idx : a.aa_idx;
            const uint16_t *rmsk = a.aa_cm_idx ? a.aa_cm_mask : a.aa_mask;
            const int64_t rstride = a.aa_cm_idx ? a.cm_pad : a.stride, rpos = a.aa_cm_idx ? (int64_t)mb + (act ? m : 0) : slot;
            for (int s16 = 0; s16 < n16; ++s16) {
                const uint4 rw = *reinterpret_cast<const uint4 *>(ridx + ((int64_t)s16 * rstride + rpos) * 16);
                const uint32_t rmask = rmsk[(int64_t)s16 * rstride + rpos];
                uint32_t r8[16];
                const uint32_t rr[4] = {rw.x, rw.y, rw.z, rw.w};
#pragma unroll
                for (int k = 0; k < 16; ++k) r8[k] = (rr[k >> 2] >> (8 * (k & 3))) & 0xffu;
                uint32_t qv[TQ][4];
#pragma unroll
                for (int k = 0; k < TQ; ++k) {
                    const uint4 qw = *reinterpret_cast<const uint4 *>(a.q_aa + qi[k] * (int64_t)a.Lpad + s16 * 16);
                    qv[k][0] = qw.x; qv[k][1] = qw.y; qv[k][2] = qw.z; qv[k][3] = qw.w;
                    nv[k] += __popc(rmask & (uint32_t)a.q_aam[qi[k] * (int64_t)n16 + s16]);
                }
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    double v[TQ];
#pragma unroll
                    for (int x = 0; x < TQ; ++x) {
                        const uint32_t qrow = ((qv[x][k >> 2] >> (8 * (k & 3))) & 0xffu) * 168u;
                        v[x] = *reinterpret_cast<const double *>(Tb + qrow + r8[k]);
                    }
#pragma unroll
                    for (int x = 0; x < TQ; ++x) tot[x] += v[x];
                }
            }
#pragma unroll
            for (int k = 0; k < TQ; ++k) {
                const int j = qg * TQ + k;
                if (j < nqt && act) {
                    const uint32_t valid = nv[k];
                    double d;
                    if (valid == 0 || (double)valid / (double)a.L < a.overlap) d = -1.0;
                    else {
                        const double r1 = 1 - tot[k] / (double)valid;
                        if (0 >= r1) d = -1.0;
                        else d = -log_libm(r1) * 1.3;
                    }
                    a.tmp_d[row_start(a.row_off, qi[k], a.stride) + sh_o[j] + m] = d;
                    if (a.item_bad && (!(d > 0) || (a.self_slot && slot == a.self_slot[qi[k]]))) a.item_bad[tile.y + j] = 1;  // (as k_cluster_dist)
                }
            }
        }
    }
}

int launch_select_clusters(apples_ctx *ctx, const SelectArgs &a, int64_t nq) {
    if (nq == 0) return 0;
    // the slot bitmap and its 16-bit in-run prefixes (runs of 16 words: 262 144 slots); with clade blocks over emission indices
    size_t dyn = (size_t)(((a.e_of_slot ? std::max(a.n_e, a.n_members) : a.n_members) + 63) >> 6) * 10;
    if (a.e_of_slot) dyn = std::max<size_t>(dyn, 1024 * 16 + (SELECT_CLUSTERS_ACC_CAP + 1) * 4);  // (the short form's list lives in the bitmap's memory)
    dyn += (size_t)knob(ctx, "APPLES_SELECT_LDS_PAD_KB", 0) * 1024;  // experiment knob: fewer workgroups per CU in the last phase
    const bool sd = a.aa_idx != nullptr;  // scoredist context: k_cluster_dist_sd computes the member distances (no by-query form)
    const bool by_query = (ctx->dbg & APPLES_DBG_CLUSTER_BY_QUERY) != 0 && !sd;  // diagnostic switch: phase 0 alone
    if (dyn > 48 * 1024) {  // (beyond the default allowance of dynamic LDS -- references of more than ~300 000 slots; per device, so asked for at every launch)
        HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&k_select_clusters<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn));
        HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&k_select_clusters<3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn));
        HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&k_select_clusters<3, SELECT_CLUSTERS_ACC_CAP, true, APPLES_TPB>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn));
        HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&k_select_clusters<3, SELECT_CLUSTERS_BIG_CAP, true, 1024>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn));
        HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&k_select_clusters<3, SELECT_CLUSTERS_HUGE_CAP, true, 1024>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn));
    }
    if (sd && !a.cl_count) { ctx->err = "scoredist cluster route without its tile scratch"; return 1; }
    if (by_query || !a.cl_count) {
        hipLaunchKernelGGL(k_select_clusters<0>, dim3((unsigned)nq), dim3(APPLES_TPB), dyn, ctx->stream, a);
        HIP_TRY(ctx, hipGetLastError());
        return 0;
    }
    HIP_TRY(ctx, hipMemsetAsync(a.cl_count, 0, (size_t)a.n_reps * sizeof(int32_t), ctx->stream));
    constexpr int BIG = SELECT_CLUSTERS_BIG_CAP, HUGE = SELECT_CLUSTERS_HUGE_CAP;
    const bool big = a.big_list != nullptr, huge = big && a.n_reps > BIG;  // (huge: the lists of the form in a.big_scr)
    SelectArgs b = a;  // the second form: a workgroup per entry of the list phase 1 writes
    b.qlist = a.big_list; b.qcount = a.big_count;
    const dim3 gbig((unsigned)std::min<int64_t>(nq, SELECT_CLUSTERS_BIG_LIST));
    if (big) HIP_TRY(ctx, hipMemsetAsync(a.big_count, 0, sizeof(int32_t), ctx->stream));
    hipLaunchKernelGGL(k_select_clusters<1>, dim3((unsigned)nq), dim3(APPLES_TPB), 0, ctx->stream, a);
    if (huge) hipLaunchKernelGGL((k_select_clusters<1, HUGE, true>), gbig, dim3(APPLES_TPB), 0, ctx->stream, b);
    else if (big) hipLaunchKernelGGL((k_select_clusters<1, BIG, true>), gbig, dim3(APPLES_TPB), 0, ctx->stream, b);
    hipLaunchKernelGGL(k_cluster_tiles, dim3(1), dim3(CL_TILES_TPB), 0, ctx->stream, a);
    hipLaunchKernelGGL(k_select_clusters<2>, dim3((unsigned)nq), dim3(APPLES_TPB), 0, ctx->stream, a);
    if (huge) hipLaunchKernelGGL((k_select_clusters<2, HUGE, true>), gbig, dim3(APPLES_TPB), 0, ctx->stream, b);
    else if (big) hipLaunchKernelGGL((k_select_clusters<2, BIG, true>), gbig, dim3(APPLES_TPB), 0, ctx->stream, b);
    if (ctx->n_cu == 0) {
        hipDeviceProp_t prop;
        HIP_TRY(ctx, hipGetDeviceProperties(&prop, ctx->device));
        ctx->n_cu = prop.multiProcessorCount;
    }
    const int per_cu = (int)knob(ctx, "APPLES_CLUSTER_WGS", 8);  // tuning knob
    if (sd) hipLaunchKernelGGL(k_cluster_dist_sd<4>, dim3((unsigned)(ctx->n_cu * std::max(per_cu, 1))), dim3(APPLES_TPB), 0, ctx->stream, a);
    else if (a.cl_mfma) hipLaunchKernelGGL(k_cluster_dist_mfma, dim3((unsigned)(ctx->n_cu * 4 * CLM_WAVES)), dim3(WAVE), 0, ctx->stream, a);  // (18 KB of LDS per wavefront: eight per CU)
    else hipLaunchKernelGGL(k_cluster_dist, dim3((unsigned)(ctx->n_cu * std::max(per_cu, 1))), dim3(APPLES_TPB), 0, ctx->stream, a);
    // the second form's last phase beside the first's, on the spare stream: a hundred-odd workgroups of 1 024 threads (their
    // rounds of member lookups are what such a workgroup takes: a quarter of the rounds of 256 threads) leave the chip idle
    if (a.blk_tiles) HIP_TRY(ctx, hipEventRecord(ctx->ev_blk[0], ctx->stream));
    if (big) {
        HIP_TRY(ctx, hipEventRecord(ctx->ev_cl[0], ctx->stream));
        HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream2, ctx->ev_cl[0], 0));
        if (huge) hipLaunchKernelGGL((k_select_clusters<3, HUGE, true, 1024>), gbig, dim3(1024), dyn, ctx->stream2, b);
        else hipLaunchKernelGGL((k_select_clusters<3, BIG, true, 1024>), gbig, dim3(1024), dyn, ctx->stream2, b);
        HIP_TRY(ctx, hipEventRecord(ctx->ev_cl[1], ctx->stream2));
    }
    // clade blocks: the S tuples inside them (k_blocks_up), on the sweep's side stream beside the last phase, which names their
    // roots in the observation lists.  Which items go without blocks is k_cluster_dist's finding (item_bad), where their tuples will
    // be is phase 2's arithmetic: the last phase needs nothing of this kernel.  Launched AFTER the last phase: that one's short
    // workgroups get in first and the persistent ones of k_blocks_up find their places as those retire; the other way round, or
    // both on one stream, the pass is 0.3 - 0.5 ms longer (profiles/r05_blk_order_exp.txt)
    auto launch_blk = [&]() -> int {
        if (a.blk_tiles) {
            BlockArgs bb{};
            bb.tiles = a.blk_tiles; bb.n_tiles = a.blk_ntiles; bb.items = a.cl_items; bb.rec_i = a.blk_rec_i; bb.rec_e = a.blk_rec_e; bb.rec_c = a.blk_rec_c; bb.stat = a.blk_stat;
            bb.rep_soff = a.rep_soff; bb.rep_moff = a.rep_moff; bb.slot_rep = a.slot_rep; bb.slot_mpos = a.slot_mpos; bb.self_slot = a.self_slot; bb.tmp_d = a.tmp_d; bb.row_off = a.row_off;
            bb.stride = a.stride; bb.pool = a.blk_pool; bb.item_sbase = a.item_sbase; bb.item_bad = a.item_bad; bb.cursor = a.q_item_cursor + 1; bb.method = a.method;
            hipStream_t bs = ctx->stream_big;
            HIP_TRY(ctx, hipStreamWaitEvent(bs, ctx->ev_blk[0], 0));
            const bool timed = ctx->ev_blk_time[0] && ctx->ev_blk_time[1];
            if (timed) HIP_TRY(ctx, hipEventRecord(ctx->ev_blk_time[0], bs));
            if (launch_blocks_up(ctx, bb, bs)) return 1;
            if (timed) { HIP_TRY(ctx, hipEventRecord(ctx->ev_blk_time[1], bs)); ctx->ev_blk_time[0] = nullptr; }  // (recorded: run_block reads them)
            HIP_TRY(ctx, hipEventRecord(ctx->ev_blk[1], bs));
        }
        return 0;
    };
    const bool blk_first = knob_on(ctx, "APPLES_BLK_FIRST");  // experiment knob: k_blocks_up ahead of the last phase's launches
    if (blk_first && launch_blk()) return 1;
    if (a.e_of_slot && a.gen_list && !knob_on(ctx, "APPLES_NO_SHORT_SPLIT")) {  // (SHORT_ONLY, above; the knob: one launch as before, diagnostic)
        const size_t dyn_short = SELECT_SHORT_ONLY_CAP * 16 + (SELECT_CLUSTERS_ACC_CAP + 1) * 4;
        SelectArgs g = a;
        g.qlist = a.gen_list; g.qcount = a.gen_count;
        HIP_TRY(ctx, hipMemsetAsync(a.gen_count, 0, sizeof(int32_t), ctx->stream));
        hipLaunchKernelGGL((k_select_clusters<3, SELECT_CLUSTERS_ACC_CAP, false, APPLES_TPB, true>), dim3((unsigned)nq), dim3(APPLES_TPB), dyn_short, ctx->stream, a);
        hipLaunchKernelGGL((k_select_clusters<3, SELECT_CLUSTERS_ACC_CAP, true, APPLES_TPB>), dim3((unsigned)nq), dim3(APPLES_TPB), dyn, ctx->stream, g);
    } else
    hipLaunchKernelGGL(k_select_clusters<3>, dim3((unsigned)nq), dim3(APPLES_TPB), dyn, ctx->stream, a);
    if (!blk_first && launch_blk()) return 1;
    if (big) HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_cl[1], 0));
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}

// The slow list of the clustered fast path (fewer than `-b` valid member distances inside the threshold): phase 4 computes the
// query's distance to every representative, walks the top-up rule over them and expands the clusters it accepts -- a few
// thousand pair counts per query where the general route computes a full row of the reference first.  What it cannot
// hold (more than ACC_CAP clusters, representatives beyond its LDS row) it forwards to that route.
int launch_select_clusters_listed(apples_ctx *ctx, const SelectArgs &a, int64_t nq_max) {
    if (nq_max == 0) return 0;
    SelectArgs b = a;
    const size_t n_words = (size_t)((a.n_members + 63) >> 6);
    b.rep_cache = a.n_reps <= 8192 ? 1 : 0;
    size_t dyn = (n_words + (n_words + 3) / 4 + (b.rep_cache ? (size_t)a.n_reps : 0)) * 8;
    if (dyn > 150 * 1024) {  // (a compute unit has 160 KB: without the row of representatives the phase forwards its queries to the general route)
        b.rep_cache = 0;
        dyn = (n_words + (n_words + 3) / 4) * 8;
    }
    if (dyn > 48 * 1024)  // (beyond the default allowance of dynamic LDS; per device, so asked for at every launch)
        HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&k_select_clusters<4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn));
    hipLaunchKernelGGL(k_select_clusters<4>, dim3((unsigned)std::min<int64_t>(nq_max, 1024)), dim3(APPLES_TPB), dyn, ctx->stream, b);
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}

// Distance-table rows arrive in the caller's column order; the selection and the sweep want them
// in slot order (columns sorted by tree level).  One gather pass per uploaded block, after which
// every later pass over the rows is a coalesced stream.
__global__ __launch_bounds__(APPLES_TPB) void k_permute_cols(const double *__restrict__ in, double *__restrict__ out,
                                                             const int32_t *__restrict__ perm, int64_t n_cols) {
    const int64_t q = blockIdx.y;
    const int64_t s = (int64_t)blockIdx.x * APPLES_TPB + threadIdx.x;
    if (s < n_cols) out[q * n_cols + s] = in[q * n_cols + perm[s]];
}

int launch_permute_cols(apples_ctx *ctx, const double *in, double *out, const int32_t *perm, int64_t nq, int64_t n_cols) {
    if (nq == 0 || n_cols == 0) return 0;
    hipLaunchKernelGGL(k_permute_cols, dim3((unsigned)((n_cols + APPLES_TPB - 1) / APPLES_TPB), (unsigned)nq), dim3(APPLES_TPB),
                       0, ctx->stream, in, out, perm, n_cols);
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}

int launch_select_fast(apples_ctx *ctx, const SelectArgs &a, int64_t nq) {
    if (nq == 0) return 0;
    SelectArgs b = a;
    const int64_t n_seg = a.stride >> 6;
    b.flat_pref = n_seg <= 16384 ? 1 : 0;
    // (APPLES_SELECT_FAST_TPB=128: two wavefronts per query and 16 workgroups per CU -- measured slower, 4.7 against 4.1 ms per C3
    // pass: the rounds' barriers and scans cost more than the extra workgroups in flight bring)
    const int tpb = (int)knob(ctx, "APPLES_SELECT_FAST_TPB", 256);
    const size_t dyn = b.flat_pref ? (size_t)(n_seg + 1) * sizeof(int) : 0;
    if (tpb == 128) hipLaunchKernelGGL(k_select_fast<128>, dim3((unsigned)nq), dim3(128), dyn, ctx->stream, b);
    else hipLaunchKernelGGL(k_select_fast<256>, dim3((unsigned)nq), dim3(256), dyn, ctx->stream, b);
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}


// Selection for singleton clusters (alignment input whose top-up rule fired, or a distance table):
// the row is streamed once, wavefront-parallel and without workgroup barriers -- each wavefront owns
// a contiguous quarter of the slots, compacts its keepers with ballots into the start of its own
// quarter of the output, and the four sparse pieces are then closed up.  If fewer than `baseobs`
// entries pass the threshold the top-up rule computes a (d, i) cut and the row is streamed again.
#ifndef STREAM_WAVES
#define STREAM_WAVES 4
#endif
#define STREAM_MERGE_CAP 1280  // `-b` up to which the top-up rule's entries are merged in LDS (half of the four queues of the smaller form)
// SU2 = 16-byte loads in flight per lane.  4 with 640-entry queues (30 KB: four workgroups per CU) for blocks of a few thousand
// rows; 8 with 1 152-entry queues (54 KB: two workgroups per CU -- fewer, deeper streams) from 8 192 rows on: config 5's 12 500-row
// shard 4.37 -> 4.00 ms (5.0 TB/s), its 4 096-row block 1.76 -> 1.97 the other way (scripts/r04_stream_su_exp.sh).  A queue holds a
// whole round of loads (at most SU x 128 candidates, or 8 x 64 with 8-byte loads) + 128; a smaller queue looked at mid-round was
// measured too: the extra branch costs more than the occupancy brings (scripts/r04_stream_su_exp2.sh).
template <int SU2>
__global__ __launch_bounds__(APPLES_TPB, SU2 > 4 ? 2 : STREAM_WAVES) void k_select_stream(SelectArgs a) {  // (the deep form's LDS lets a CU hold two)
    constexpr int STREAM_QUEUE = (SU2 > 4 ? SU2 : 4) * 128 + 128;  // candidates a wavefront queues before it looks at them
    static_assert(4 * STREAM_QUEUE >= 2 * STREAM_MERGE_CAP, "the merge's two halves live in the four queues");
    __shared__ int sh_i[8];
    __shared__ int sh_j[8];
    __shared__ double sh_d[8];
    __shared__ int sh_wcnt[4];
    __shared__ int sh_znode;
    __shared__ int sh_qs[APPLES_TPB / WAVE][STREAM_QUEUE];     // per-wavefront queue of candidates: slot (relative to the quarter) ...
    __shared__ double sh_qd[APPLES_TPB / WAVE][STREAM_QUEUE];  // ... and distance
    const int64_t n_list = a.qcount ? (int64_t)*a.qcount : a.n_rows_plain;
    const int tid = threadIdx.x, lane = tid & (WAVE - 1), wv = tid / WAVE;
    const double thr = a.thr;
    const bool table = a.table_mode != 0;
    // A row's node is needed while streaming only to skip table columns that are not tree leaves; if
    // there are none (or this is not a table) it is looked up for the entries that are kept
    const bool early_node = table && !a.cols_all_in_tree;
    const bool late_node = a.cols_all_in_tree != 0;  // (table or alignment: no slot without a tree leaf)
    constexpr int SU1 = 8;  // independent load instructions per lane in flight with 8-byte loads (SU2: with 16-byte loads)
    // (a.row_cursor: rows handed out one at a time instead of dealt out in advance -- a tuning knob, see launch_select)
    __shared__ int sh_row;
    for (int64_t r = blockIdx.x;; r += gridDim.x) {
        if (a.row_cursor) {
            if (tid == 0) sh_row = atomicAdd(a.row_cursor, 1);
            __syncthreads();
            r = sh_row;
            // (the row's closing barrier keeps thread 0's next write behind this read)
        }
        if (r >= n_list) break;
        const int64_t q = a.qlist ? a.qlist[r] : r;
        const double *row = a.dist + (a.rows_by_query ? q : r) * a.stride;
        // a compact row: nm of the reference's values, entry s belongs to slot rs[s] (slots ascending); else entry s = slot s
        const int32_t *rs = a.row_len ? reinterpret_cast<const int32_t *>(row + a.row_cap) : nullptr;
        if (a.row_len && a.row_len[r] < 0) continue;  // (block-uniform: this query's list did not fit; another launch has its row)
        const int64_t nm = a.row_len ? a.row_len[r] : a.n_members;
        // quarters aligned to 64 entries
        const int64_t per = ((nm + 4 * WAVE - 1) / (4 * WAVE)) * WAVE;
        const int64_t w_lo = std::min<int64_t>(nm, per * wv), w_hi = std::min<int64_t>(nm, per * (wv + 1));
        const int self = a.self_slot ? a.self_slot[q] : -1;
        int32_t *o_node = a.obs_node + q * a.obs_cap;
        double *o_dist = a.obs_dist + q * a.obs_cap;
        int32_t *cg = a.cnt_gt ? a.cnt_gt + q * (int64_t)(a.height + 2) : nullptr;
        double cut_d = -INF_D;
        int cut_i = -1;
        int n_total = 0, thr_cnt = 0, n_emit = 0;
        int z_i = 0x7fffffff, z_node = -2;
        double z_d = INF_D;
        bool merged = false;  // the top-up rule's entries were merged into the list without another pass over the row
        for (int round = 0; round < 2; ++round) {
            n_total = 0; thr_cnt = 0; z_i = 0x7fffffff; z_node = -2; z_d = INF_D;
            int wbase = 0;  // keepers this wavefront has written (wave-uniform)
            // One pass over this wavefront's quarter of the row, V values per lane and load (V = 2: 16-byte loads, lane l of a
            // load instruction holds slots 2l and 2l + 1 of its 128; rows whose quarter starts on an odd slot or at an
            // unaligned address take V = 1).  SU load instructions are in flight before the first is consumed; survivors
            // leave in slot order (lane-major, then the lane's own two).
            // Candidates (0 <= d <= bound) go through a small per-wavefront queue in LDS, in slot order, and are looked at
            // 64 at a time (node lookup, cut test, own entry, zero, ordered emission): the streaming loop itself then holds
            // nothing but loads, two comparisons per value and a ballot, and its loads are waited for one by one.
            int q_n = 0;  // queued candidates (wave-uniform)
            auto drain = [&]() {
                for (int i0 = 0; i0 < q_n; i0 += WAVE) {
                    const int i = i0 + lane;
                    bool emit = false;
                    int node = 0;
                    double d = 0;
                    if (i < q_n) {
                        const int64_t e_ = w_lo + sh_qs[wv][i];
                        const int64_t s = rs ? rs[e_] : e_;  // the entry's slot
                        d = sh_qd[wv][i];
                        bool ok = true;
                        if (early_node) { node = a.slot_node[s]; ok = node >= 0; }
                        if (ok) {
                            const bool in_thr = d <= thr;
                            thr_cnt += in_thr;
                            bool in_dict = in_thr;
                            if (!in_thr) in_dict = key_le(d, a.slot_rep[s], cut_d, cut_i);  // (bound > thr only with a cut)
                            if (in_dict && (int)s != self) {
                                // (every slot a tree leaf: the slot itself is written and turned into its node in the row's
                                // tail, so this loop waits for no load)
                                if (late_node) node = (int)s;
                                else if (!early_node) node = a.slot_node[s];
                                ++n_total;
                                if (d == 0) {
                                    const int ri = a.slot_rep[s];
                                    if (ri < z_i) { z_i = ri; z_node = late_node ? a.slot_node[s] : node; z_d = 0; }
                                }
                                emit = node >= 0;
                            }
                        }
                    }
                    const unsigned long long m = __ballot(emit);
                    if (emit) {
                        const int64_t o = w_lo + wbase + __popcll(m & ((1ull << lane) - 1ull));
                        o_node[o] = node;
                        o_dist[o] = d;
                    }
                    wbase += __popcll(m);
                }
                q_n = 0;
                __builtin_amdgcn_wave_barrier();
            };
            auto pass = [&](auto vc) {
                constexpr int V = decltype(vc)::value;
                constexpr int SU = V == 2 ? SU2 : SU1;
                const unsigned long long below = (1ull << lane) - 1ull;
                const double *wrow = row + w_lo;           // this wavefront's quarter: slots [0, wn) of it
                const int wn_all = (int)(w_hi - w_lo);
                const int wn = V == 2 ? (wn_all & ~1) : wn_all;  // 16-byte loads: whole pairs; an odd last slot follows below
                if (wn < V) return;                        // (V == 2 is only chosen for quarters of at least two slots)
                const int step = SU * WAVE * V;
                // nothing above this bound can be kept: the threshold, or (second pass) the larger of it and the cut
                const double bound = (cut_i >= 0 && cut_d > thr) ? cut_d : thr;
                // loads are unconditional (the address is clamped into the quarter, the value masked by its slot
                // afterwards): the round below is straight-line code, so every load is waited for by count, not all at once
                const int last = wn - V;
                auto fetch = [&](int s0, double *v) {
                    const int c = s0 < last ? s0 : last;
                    if (V == 2) {
                        const double2 t2 = *reinterpret_cast<const double2 *>(wrow + c);
                        v[0] = t2.x; v[V - 1] = t2.y;
                    } else {
                        v[0] = wrow[c];
                    }
                };
                // a ring of SU loads per lane: slot u of the next round is requested before slot u of this round is looked
                // at, so SU load instructions per lane stay in flight for the whole pass
                double ring[SU][V];
#pragma unroll
                for (int u = 0; u < SU; ++u) fetch((u * WAVE + lane) * V, ring[u]);
                int sb = 0;
                while (sb < wn) {
                    do {  // rounds, until the queue might not hold another one's candidates
#pragma unroll
                        for (int u = 0; u < SU; ++u) {
                            const int s0 = sb + (u * WAVE + lane) * V;
                            double cur[V];
#pragma unroll
                            for (int e = 0; e < V; ++e) cur[e] = ring[u][e];
                            fetch(s0 + step, ring[u]);
                            bool cand[V];
#pragma unroll
                            for (int e = 0; e < V; ++e) cand[e] = s0 + e < wn && cur[e] >= 0 && cur[e] <= bound;
                            const unsigned long long m0 = __ballot(cand[0]);
                            const unsigned long long m1 = V == 2 ? __ballot(cand[V - 1]) : 0ull;
                            if (m0 | m1) {
                                int at = q_n + __popcll(m0 & below) + __popcll(m1 & below);
                                if (cand[0]) { sh_qs[wv][at] = s0; sh_qd[wv][at] = cur[0]; ++at; }
                                if (V == 2 && cand[V - 1]) { sh_qs[wv][at] = s0 + 1; sh_qd[wv][at] = cur[V - 1]; }
                                q_n += __popcll(m0) + __popcll(m1);
                            }
                        }
                        sb += step;
                    } while (sb < wn && q_n <= STREAM_QUEUE - SU * V * WAVE);
                    __builtin_amdgcn_wave_barrier();
                    drain();
                }
                if (V == 2 && (wn_all & 1)) {  // the quarter's odd last slot
                    const double d = wrow[wn_all - 1];
                    if (d >= 0 && d <= bound) {
                        if (lane == 0) { sh_qs[wv][0] = wn_all - 1; sh_qd[wv][0] = d; }
                        q_n = 1;
                        __builtin_amdgcn_wave_barrier();
                        drain();
                    }
                }
            };
            const bool vec2 = (((size_t)(row + w_lo)) & 15u) == 0 && w_hi - w_lo >= 2;
            if (vec2) pass(std::integral_constant<int, 2>());
            else pass(std::integral_constant<int, 1>());
            if (lane == 0) sh_wcnt[wv] = wbase;
            const int obs = block_sum(thr_cnt, sh_i);  // (barriers inside also publish sh_wcnt)
            if (round == 0 && obs < a.baseobs) {
                // ---- top-up (Reference.py:144 / PoolQueryWorker.py:55): smallest (d, i) beyond the threshold
                int have = obs;
                constexpr int KL = 4;
                double cd[KL];
                int ci[KL];
                int cs[KL];  // the slots of the cached keys
                // what the rule takes is noted as it is taken (the queues of the streaming pass are idle: their upper half), so
                // that the row need not be streamed a third time to collect it
                int *const mx_s = &sh_qs[0][0] + STREAM_MERGE_CAP;
                double *const mx_d = &sh_qd[0][0] + STREAM_MERGE_CAP;
                const bool direct = late_node && a.baseobs <= STREAM_MERGE_CAP && !a.third_pass;
                int n_x = 0;
                int head = 0, filled = 0;
                bool more = true;
                auto refill = [&](double lo_d, int lo_i) {
                    filled = 0; head = 0;
                    int seen = 0;
                    for (int64_t s0 = tid; s0 < nm; s0 += 4 * APPLES_TPB) {  // four independent loads in flight
                        double dd[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const int64_t s = s0 + u * APPLES_TPB;
                            dd[u] = s < nm ? row[s] : -1.0;
                        }
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const int64_t e_ = s0 + u * APPLES_TPB;
                            const double d = dd[u];
                            if (!(d >= 0 && d > thr) || d < lo_d) continue;
                            const int64_t s = rs ? rs[e_] : e_;  // the entry's slot
                            if (early_node && a.slot_node[s] < 0) continue;
                            // the index decides ties only: beyond the cut and beyond the cache's last key it is not needed
                            const bool fits = filled < KL || d <= cd[KL - 1];
                            if (d > lo_d && !fits) { ++seen; continue; }
                            const int i = a.slot_rep[s];
                            if (!key_lt(lo_d, lo_i, d, i)) continue;
                            ++seen;
                            int pos = filled < KL ? filled : KL;
                            while (pos > 0 && key_lt(d, i, cd[pos - 1], ci[pos - 1])) --pos;
                            if (pos < KL) {
                                for (int k = (filled < KL ? filled : KL - 1); k > pos; --k) { cd[k] = cd[k - 1]; ci[k] = ci[k - 1]; cs[k] = cs[k - 1]; }
                                cd[pos] = d; ci[pos] = i; cs[pos] = (int)s;
                                if (filled < KL) ++filled;
                            }
                        }
                    }
                    more = seen > filled;
                };
                refill(cut_d, cut_i);
                while (have < a.baseobs) {
                    if (head == filled && more) refill(cut_d, cut_i);
                    double bd = head < filled ? cd[head] : INF_D;
                    int bi = head < filled ? ci[head] : 0x7fffffff, bj = 0;
                    const int mine = bi;
                    block_argmin3(bd, bi, bj, sh_d, sh_i, sh_j);
                    if (bi == 0x7fffffff) break;
                    if (mine == bi && head < filled) {
                        if (direct) { mx_s[n_x] = cs[head]; mx_d[n_x] = cd[head]; }  // (n_x < baseobs <= STREAM_MERGE_CAP)
                        ++head;
                    }
                    ++n_x;
                    cut_d = bd; cut_i = bi;
                    ++have;
                }
                if (cut_i >= 0 && !direct) continue;  // stream again with the cut
                if (cut_i >= 0) {
                    // The second pass would keep what the first kept plus every key in (threshold, cut]: exactly the n_x entries
                    // taken above (keys are distinct: one per slot), the query's own slot aside.  Fewer than 2 x baseobs entries
                    // in all: the first pass's four pieces and the new entries go through LDS, every entry finds its place
                    // by counting the smaller slots, and the list is written closed up.
                    int *const m_s = &sh_qs[0][0];
                    double *const m_d = &sh_qd[0][0];
                    __syncthreads();  // (mx_* written; sh_wcnt published by block_sum above)
                    int tot0 = 0;
                    for (int k = 0; k < 4; ++k) {
                        const int cnt = sh_wcnt[k];
                        const int64_t src = std::min<int64_t>(nm, per * k);
                        for (int c = tid; c < cnt; c += APPLES_TPB) { m_s[tot0 + c] = o_node[src + c]; m_d[tot0 + c] = o_dist[src + c]; }
                        tot0 += cnt;
                    }
                    __syncthreads();
                    const int total = tot0 + n_x;
                    int dropped = 0;
                    for (int j = 0; j < n_x; ++j) dropped += mx_s[j] == self;
                    for (int e = tid; e < total; e += APPLES_TPB) {
                        const int s_e = e < tot0 ? m_s[e] : mx_s[e - tot0];
                        if (e >= tot0 && s_e == self) continue;
                        const double d_e = e < tot0 ? m_d[e] : mx_d[e - tot0];
                        int rank = 0;
                        for (int f = 0; f < tot0; ++f) rank += m_s[f] < s_e;
                        for (int f = 0; f < n_x; ++f) rank += mx_s[f] < s_e && mx_s[f] != self;
                        o_node[rank] = s_e;  // (slots for now: late_node; they become nodes in the row's tail below)
                        o_dist[rank] = d_e;
                    }
                    n_emit = total - dropped;
                    if (tid == 0) n_total += n_x - dropped;
                    merged = true;
                }
            }
            break;
        }
        // ---- close up the four pieces: piece w starts at w_lo of wavefront w ----------------------------
        __syncthreads();
        int cnts[4], pre[4];
        int acc = 0;
        for (int k = 0; k < 4; ++k) { cnts[k] = sh_wcnt[k]; pre[k] = acc; acc += cnts[k]; }
        if (!merged) n_emit = acc;
        for (int k = 1; k < 4 && !merged; ++k) {
            const int64_t src = std::min<int64_t>(nm, per * k);
            if (src == pre[k]) continue;
            for (int c0 = 0; c0 < cnts[k]; c0 += APPLES_TPB) {  // moving down: read, barrier, write
                const int c = c0 + tid;
                int nd = 0;
                double dd = 0;
                if (c < cnts[k]) { nd = o_node[src + c]; dd = o_dist[src + c]; }
                __syncthreads();
                if (c < cnts[k]) { o_node[pre[k] + c] = nd; o_dist[pre[k] + c] = dd; }
                __syncthreads();
            }
        }
        n_total = block_sum(n_total, sh_i);
        double zd = z_d; int zi = z_i, zp = 0;
        block_argmin3(zd, zi, zp, sh_d, sh_i, sh_j);
        if (tid == 0) sh_znode = -2;
        __syncthreads();
        if (z_i == zi && zi != 0x7fffffff) sh_znode = z_node;
        __syncthreads();
        // per-level offsets into the level-sorted list (the sweep's cnt_gt); with the slots written in place of the nodes,
        // level and node of an entry are two independent lookups by slot (one round trip per 256 entries, not two)
        if (late_node) {
            for (int i = tid; cg && i <= n_emit; i += APPLES_TPB) {
                const int lv = (i < n_emit) ? a.slot_level[o_node[i]] : -1;
                const int lprev = (i == 0) ? a.height + 1 : a.slot_level[o_node[i - 1]];
                for (int l = lv; l < lprev; ++l) cg[l + 1] = i;
            }
            __syncthreads();  // (every neighbour's slot has been read before it becomes a node)
            for (int i = tid; i < n_emit; i += APPLES_TPB) o_node[i] = a.slot_node[o_node[i]];
        } else {
        for (int i = tid; cg && i <= n_emit; i += APPLES_TPB) {
            const int lv = (i < n_emit) ? a.node_level[o_node[i]] : -1;
            const int lprev = (i == 0) ? a.height + 1 : a.node_level[o_node[i - 1]];
            for (int l = lv; l < lprev; ++l) cg[l + 1] = i;
        }
        }
        if (tid == 0) {
            apples_placement p;
            p.edge = 0; p.flags = 0; p.error = 0.0; p.distal = 0.0; p.pendant = 0.0; p.n_obs = n_total; p.n_valid = 0;
            int ne = n_emit;
            if (zi != 0x7fffffff) {
                p.flags = APPLES_F_EXACT | APPLES_F_PENDANT_INT;
                p.edge = sh_znode;
                if (sh_znode < 0) { p.flags |= APPLES_F_ZERO_NOT_IN_TREE; p.edge = -1; }
                ne = 0;
            } else if (n_total <= 2) {
                p.flags = APPLES_F_INSUFFICIENT | APPLES_F_PENDANT_INT;
                p.edge = -1;
                ne = 0;
            } else if (ne < 2) {
                p.flags = APPLES_F_DEGENERATE | APPLES_F_PENDANT_INT;
                p.edge = -1;
                ne = 0;
            }
            a.out[q] = p;
            a.n_obs[q] = ne;
            enlist(a, q, ne);
        }
        __syncthreads();
    }
}

// Top-up path of the fused pipeline (singleton clusters): a query with fewer than `-b` references
// inside the threshold keeps exactly its `-b` nearest by (distance, index) -- everything inside the
// threshold is among them.  The listed distance kernel left the smallest key of every 64-slot
// segment next to the full row, and the `-b` nearest can only live in the `-b` segments with the
// smallest minima: pick those, rank their <= 64 * b entries, write the survivors in slot order.
// The row itself (n_members values) is never streamed.
#define TOPUP_MAX_B 256
#define TOPUP_KEYS 16  // keys a thread can hold in registers for the fast selection

// wavefront-wide lexicographic arg-min over (d, i) with payload j; result in every lane (shuffles only)
__device__ __forceinline__ void wave_argmin3(double &d, int &i, int &j) {
    for (int o = WAVE / 2; o > 0; o >>= 1) {
        const double d2 = shfl_down_f64(d, o);
        const int i2 = __shfl_down(i, o, WAVE);
        const int j2 = __shfl_down(j, o, WAVE);
        if (d2 < d || (d2 == d && i2 < i)) { d = d2; i = i2; j = j2; }
    }
    d = __hiloint2double(__shfl(__double2hiint(d), 0, WAVE), __shfl(__double2loint(d), 0, WAVE));
    i = __shfl(i, 0, WAVE);
    j = __shfl(j, 0, WAVE);
}

// The B smallest (d, i) keys of a workgroup, each thread holding up to TOPUP_KEYS keys in registers
// (i == INT_MAX marks an empty slot).  Every wavefront first extracts its own B smallest with
// shuffle-only rounds (no workgroup barrier per round), then the at most 4 B candidates are ranked by
// counting.  Results in key order: out_d/out_i/out_j[0..n), n returned.  All in LDS.
__device__ int block_smallest(const double (&kd)[TOPUP_KEYS], const int (&ki)[TOPUP_KEYS], const int (&kj)[TOPUP_KEYS],
                              int B, double *cand_d, int *cand_i, int *cand_j, int *cand_n, double *out_d, int *out_i,
                              int *out_j) {
    const int lane = threadIdx.x & (WAVE - 1), wv = threadIdx.x / WAVE;
    double last_d = -INF_D;
    int last_i = -1, mine = 0;
    for (int k = 0; k < B; ++k) {
        double bd = INF_D;
        int bi = 0x7fffffff, bj = 0;
#pragma unroll
        for (int t = 0; t < TOPUP_KEYS; ++t)
            if (ki[t] != 0x7fffffff && key_lt(last_d, last_i, kd[t], ki[t]) && key_lt(kd[t], ki[t], bd, bi)) { bd = kd[t]; bi = ki[t]; bj = kj[t]; }
        wave_argmin3(bd, bi, bj);
        if (bi == 0x7fffffff) break;
        if (lane == 0) { cand_d[wv * TOPUP_MAX_B + k] = bd; cand_i[wv * TOPUP_MAX_B + k] = bi; cand_j[wv * TOPUP_MAX_B + k] = bj; }
        last_d = bd; last_i = bi;
        ++mine;
    }
    if (lane == 0) cand_n[wv] = mine;
    __syncthreads();
    int total = 0, n[APPLES_TPB / WAVE];
    for (int w = 0; w < APPLES_TPB / WAVE; ++w) { n[w] = cand_n[w]; total += n[w]; }
    for (int c = threadIdx.x; c < total; c += APPLES_TPB) {  // candidate c = the k-th of wavefront w
        int w = 0, k = c;
        while (k >= n[w]) { k -= n[w]; ++w; }
        const double d = cand_d[w * TOPUP_MAX_B + k];
        const int i = cand_i[w * TOPUP_MAX_B + k];
        int rank = 0;
        for (int w2 = 0; w2 < APPLES_TPB / WAVE; ++w2)
            for (int k2 = 0; k2 < n[w2]; ++k2) rank += key_lt(cand_d[w2 * TOPUP_MAX_B + k2], cand_i[w2 * TOPUP_MAX_B + k2], d, i);
        if (rank < B) { out_d[rank] = d; out_i[rank] = i; out_j[rank] = cand_j[w * TOPUP_MAX_B + k]; }
    }
    __syncthreads();
    return total < B ? total : B;
}

__global__ __launch_bounds__(APPLES_TPB) void k_select_topup(SelectArgs a) {
    __shared__ int sh_i[8];
    __shared__ int sh_j[8];
    __shared__ double sh_d[8];
    __shared__ int sh_cand[TOPUP_MAX_B];
    __shared__ int sh_sorted[TOPUP_MAX_B];
    __shared__ int sh_znode;
    __shared__ double cand_d[(APPLES_TPB / WAVE) * TOPUP_MAX_B], out_d[TOPUP_MAX_B];
    __shared__ int cand_i[(APPLES_TPB / WAVE) * TOPUP_MAX_B], cand_j[(APPLES_TPB / WAVE) * TOPUP_MAX_B], cand_n[APPLES_TPB / WAVE];
    __shared__ int out_i[TOPUP_MAX_B];
    const int64_t n_list = *a.qcount;
    const int tid = threadIdx.x;
    const int64_t nm = a.n_members;
    const int n_seg = (int)((nm + 63) >> 6);
    const int B = a.baseobs;
    for (int64_t r = blockIdx.x; r < n_list; r += gridDim.x) {
        const int64_t q = a.qlist[r];
        const double *row = a.dist + r * a.stride;
        const double *smd = a.segmin_d + r * a.stride;
        const int32_t *smi = a.segmin_i + r * a.stride;
        const int self = a.self_slot ? a.self_slot[q] : -1;
        int32_t *o_node = a.obs_node + q * a.obs_cap;
        double *o_dist = a.obs_dist + q * a.obs_cap;
        int32_t *cg = a.cnt_gt ? a.cnt_gt + q * (int64_t)(a.height + 2) : nullptr;
        // ---- the B segments with the smallest minima, in key order
        int n_cand = 0;
        if (n_seg <= TOPUP_KEYS * APPLES_TPB) {  // keys fit in registers: wavefront-local rounds + one ranking
            double kd[TOPUP_KEYS];
            int ki[TOPUP_KEYS], kj[TOPUP_KEYS];
#pragma unroll
            for (int t = 0; t < TOPUP_KEYS; ++t) {
                const int sgm = tid + t * APPLES_TPB;
                const bool ok = sgm < n_seg;
                kd[t] = ok ? smd[sgm] : INF_D;
                ki[t] = ok ? smi[sgm] : 0x7fffffff;
                kj[t] = sgm;
            }
            n_cand = block_smallest(kd, ki, kj, B, cand_d, cand_i, cand_j, cand_n, out_d, out_i, sh_cand);
        } else {
            double last_d = -INF_D;
            int last_i = -1;
            for (int k = 0; k < B; ++k) {
                double bd = INF_D;
                int bi = 0x7fffffff, bs = 0;
                for (int sgm = tid; sgm < n_seg; sgm += APPLES_TPB) {
                    const double d = smd[sgm];
                    const int i = smi[sgm];
                    if (i != 0x7fffffff && key_lt(last_d, last_i, d, i) && key_lt(d, i, bd, bi)) { bd = d; bi = i; bs = sgm; }
                }
                block_argmin3(bd, bi, bs, sh_d, sh_i, sh_j);
                if (bi == 0x7fffffff) break;
                if (tid == 0) sh_cand[k] = bs;
                last_d = bd; last_i = bi;
                ++n_cand;
            }
            __syncthreads();
        }
        // ---- the B smallest keys among the candidates' entries -> the cut (Reference.py:144-152)
        const int n_ent = n_cand * 64;
        double cut_d = -INF_D;
        int cut_i = -1;
        if (n_ent <= TOPUP_KEYS * APPLES_TPB) {
            double kd[TOPUP_KEYS];
            int ki[TOPUP_KEYS], kj[TOPUP_KEYS];
#pragma unroll
            for (int t = 0; t < TOPUP_KEYS; ++t) {
                const int e = tid + t * APPLES_TPB;
                kd[t] = INF_D; ki[t] = 0x7fffffff; kj[t] = 0;
                if (e < n_ent) {
                    const int64_t s = (int64_t)sh_cand[e >> 6] * 64 + (e & 63);
                    if (s < nm) {
                        const double d = row[s];
                        if (d >= 0) { kd[t] = d; ki[t] = a.slot_rep[s]; }
                    }
                }
            }
            const int n_top = block_smallest(kd, ki, kj, B, cand_d, cand_i, cand_j, cand_n, out_d, out_i, sh_sorted);
            if (n_top > 0) { cut_d = out_d[n_top - 1]; cut_i = out_i[n_top - 1]; }
            __syncthreads();
        } else {
            for (int k = 0; k < B; ++k) {
                double bd = INF_D;
                int bi = 0x7fffffff, bj = 0;
                for (int e = tid; e < n_ent; e += APPLES_TPB) {
                    const int64_t s = (int64_t)sh_cand[e >> 6] * 64 + (e & 63);
                    if (s >= nm) continue;
                    const double d = row[s];
                    if (!(d >= 0)) continue;
                    const int i = a.slot_rep[s];
                    if (key_lt(cut_d, cut_i, d, i) && key_lt(d, i, bd, bi)) { bd = d; bi = i; }
                }
                block_argmin3(bd, bi, bj, sh_d, sh_i, sh_j);
                if (bi == 0x7fffffff) break;
                cut_d = bd; cut_i = bi;
            }
        }
        // ---- candidates in slot order
        if (tid < n_cand) {
            const int mine = sh_cand[tid];
            int rank = 0;
            for (int k = 0; k < n_cand; ++k) rank += sh_cand[k] < mine;
            sh_sorted[rank] = mine;
        }
        __syncthreads();
        int base = 0, n_total = 0;
        int z_i = 0x7fffffff, z_node = -2;
        for (int e0 = 0; e0 < n_ent; e0 += APPLES_TPB) {
            const int e = e0 + tid;
            int emit = 0, node = -1;
            double d = 0;
            if (e < n_ent) {
                const int64_t s = (int64_t)sh_sorted[e >> 6] * 64 + (e & 63);
                if (s < nm) {
                    d = row[s];
                    const int i = a.slot_rep[s];
                    if (d >= 0 && key_le(d, i, cut_d, cut_i) && (int)s != self) {
                        ++n_total;
                        node = a.slot_node[s];
                        if (d == 0 && i < z_i) { z_i = i; z_node = node; }
                        emit = node >= 0;
                    }
                }
            }
            int tot;
            const int pos = base + block_excl_scan(emit, sh_j, &tot);
            if (emit) { o_node[pos] = node; o_dist[pos] = d; }
            base += tot;
        }
        __syncthreads();
        const int n_emit = base;
        n_total = block_sum(n_total, sh_i);
        double zd = 0; int zi = z_i, zp = 0;
        block_argmin3(zd, zi, zp, sh_d, sh_i, sh_j);
        if (tid == 0) sh_znode = -2;
        __syncthreads();
        if (z_i == zi && zi != 0x7fffffff) sh_znode = z_node;
        __syncthreads();
        // per-level offsets into the level-sorted list (the sweep's cnt_gt)
        for (int i = tid; cg && i <= n_emit; i += APPLES_TPB) {
            const int lv = (i < n_emit) ? a.node_level[o_node[i]] : -1;
            const int lprev = (i == 0) ? a.height + 1 : a.node_level[o_node[i - 1]];
            for (int l = lv; l < lprev; ++l) cg[l + 1] = i;
        }
        if (tid == 0) {
            apples_placement p;
            p.edge = 0; p.flags = 0; p.error = 0.0; p.distal = 0.0; p.pendant = 0.0; p.n_obs = n_total; p.n_valid = 0;
            int ne = n_emit;
            if (zi != 0x7fffffff) {
                p.flags = APPLES_F_EXACT | APPLES_F_PENDANT_INT;
                p.edge = sh_znode;
                if (sh_znode < 0) { p.flags |= APPLES_F_ZERO_NOT_IN_TREE; p.edge = -1; }
                ne = 0;
            } else if (n_total <= 2) {
                p.flags = APPLES_F_INSUFFICIENT | APPLES_F_PENDANT_INT;
                p.edge = -1;
                ne = 0;
            } else if (ne < 2) {
                p.flags = APPLES_F_DEGENERATE | APPLES_F_PENDANT_INT;
                p.edge = -1;
                ne = 0;
            }
            a.out[q] = p;
            a.n_obs[q] = ne;
            enlist(a, q, ne);
        }
        __syncthreads();
    }
}

int launch_select_topup(apples_ctx *ctx, const SelectArgs &a, int64_t nq) {
    if (nq == 0) return 0;
    hipLaunchKernelGGL(k_select_topup, dim3((unsigned)std::min<int64_t>(nq, 1024)), dim3(APPLES_TPB), 0, ctx->stream, a);
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}

int launch_select(apples_ctx *ctx, const SelectArgs &a, int64_t nq) {
    if (nq == 0) return 0;
    // listed mode: the list length lives on the device, so a bounded grid loops over it
    unsigned grid = a.qcount ? (unsigned)std::min<int64_t>(nq, 512) : (unsigned)nq;
    SelectArgs b = a;
    b.n_rows_plain = nq;
    const bool no_stream = (ctx->dbg & APPLES_DBG_NO_STREAM_SELECT) != 0;  // diagnostic switch
    if (a.all_singleton && !a.gather && !no_stream) {  // singleton clusters, rows in slot order: barrier-free streaming form
        // one workgroup per row up to 4 per CU, then rows in turn (C5: 1.70 ms per 4 096 rows against 1.85-1.9 with one
        // workgroup per row); APPLES_STREAM_GRID: tuning knob
        const int cap = (int)knob(ctx, "APPLES_STREAM_GRID", 1024);
        if (cap > 0) grid = std::min<unsigned>(grid, (unsigned)cap);
        b.third_pass = (ctx->dbg & APPLES_DBG_STREAM_THIRD_PASS) ? 1 : 0;  // diagnostic switch
        // rows dealt out in advance (row r to workgroup r mod grid) unless APPLES_STREAM_DYNAMIC_ROWS hands them out one at a time:
        // measured on config 5, selection 1.80 -> 1.90 ms for 4 096 rows and 4.55 -> 4.47 for 12 500 (the atomic and its barrier
        // per row against a better balance of the rows that are streamed twice); off
        const bool dynamic_rows = knob_on(ctx, "APPLES_STREAM_DYNAMIC_ROWS");  // tuning knob
        if (!dynamic_rows) b.row_cursor = nullptr;
        if (b.row_cursor) HIP_TRY(ctx, hipMemsetAsync(b.row_cursor, 0, sizeof(int32_t), ctx->stream));
        const int su_env = (int)knob(ctx, "APPLES_STREAM_SU", 0);  // tuning knob: 4 or 8
        const bool deep = su_env ? su_env == 8 : (!a.qcount && nq >= 8192);
        if (deep) hipLaunchKernelGGL(k_select_stream<8>, dim3(grid), dim3(APPLES_TPB), 0, ctx->stream, b);
        else hipLaunchKernelGGL(k_select_stream<4>, dim3(grid), dim3(APPLES_TPB), 0, ctx->stream, b);
    }
    else {
        // clustered rows: the representatives' distances in LDS where they fit (48 KB: 6 144 of them)
        b.rep_cache = (!a.all_singleton && a.n_reps > 0 && a.n_reps <= 6144) ? 1 : 0;
        // a listed row is one workgroup's job and its time is that workgroup's latency (a few hundred rows per batch at most, far
        // fewer than the chip holds): 1 024 threads per row there, a quarter of the rounds
        const size_t dyn = b.rep_cache ? (size_t)a.n_reps * sizeof(double) : 0;
        if (a.qcount) hipLaunchKernelGGL(k_select<1024>, dim3(grid), dim3(1024), dyn, ctx->stream, b);
        else hipLaunchKernelGGL(k_select<APPLES_TPB>, dim3(grid), dim3(APPLES_TPB), dyn, ctx->stream, b);
    }
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}
