// Least-squares placement sweep: one persistent workgroup walks queries, one query at a time.
//
// Per query (apples/PoolQueryWorker.py:101-133):
//   Subtree            apples/Subtree.py:23-43   -> bottom-up level loop below (marks + LCA)
//   all_S_values       apples/OLS.py:12-44 ...   -> fused into the same bottom-up loop
//   all_R_values       apples/OLS.py:46-80 ...   -> top-down level loop
//   placement_per_edge apples/OLS.py:83-97 + apples/util.py:6-54   -> fused into the top-down loop
//   error_per_edge     apples/OLS.py:100-128 ... -> fused into the top-down loop
//   placement          apples/Algorithm.py:62-101 -> wavefront/LDS arg-min over the candidates
//   unroll_changes     apples/Subtree.py:72-76   -> the team's node bits are cleared / its map tag is incremented
//
// The reference pops the deepest frontier node, marks it valid and pushes its parent until one
// node (the LCA) is left.  Level-synchronous form: the nodes at level l are the observed leaves
// of that level (contiguous in the level-sorted obs list) plus the parents reported from level
// l+1; the loop stops when a level holds a single node and no shallower leaves remain.
//
// Data layout (the kernel is bound by HBM/L2 misses on scattered 64-byte lines and by the chain of
// dependent loads in every level step, so lines are few and whole): NodeRec (64 B, tree constant:
// the node and what a parent needs of its first two children); per team and per INTERNAL subtree
// node, in compact order, one Rec (64 B: a tuple, descriptors of the first two valid children,
// node id).  The tuple is S after the bottom-up pass and is overwritten with R by the node's
// parent on the way down.  Observed leaves get no record: a leaf is named by its position j in
// the query's level-sorted observation list (child descriptor -(j+2)) and its tuple is rebuilt
// from the distance whenever a parent needs it.  The top-down pass is parent-centric: a node forms
// each valid child's R (siblings in file order, then its own lifted R), solves that child's 2x2
// system and residual, and stores R only for children that are internal.
// Which nodes are in the subtree, and where their records are: NodeBits (LDS, trees up to ~50 k
// nodes) or NodeMap (global scratch, bigger trees), see below.
//
// Bit parity: fp64, compiled with -ffp-contract=off; every sum is taken in the order of the
// cited source line, children/siblings in file order and the parent term last.
// `x ** 2` is libm pow in the reference; pow2_libm below reproduces its bits.
// HBM-bound by design: ~60 fp64 flops against ~330 B per node; no MFMA.
#include <algorithm>
#include <cstdlib>

#include "common.h"

#include "sweep_math.h"

// One 64-byte record per INTERNAL subtree node, in compact order of discovery.  T holds the node's
// S tuple after the bottom-up pass; on the way down the parent overwrites it with the node's R
// tuple (S is dead once the parent has formed the siblings' R values and solved this edge), so a
// node costs one line of scratch, written once per pass.
struct __attribute__((aligned(64))) Rec {
    double T[6];
    int32_t k0, k1;  // first two valid children: > 0 internal (compact index + 1), <= -2 leaf (-(j+2))
    int32_t node;
    uint32_t meta;   // number of valid children | META_POLY | META_K0C1
};
#define META_POLY 0x40000000u  // more than two children in the tree: walk the CSR list
#define META_K0C1 0x80000000u  // the only valid child is the node's second child
#define META_NK 0x00ffffffu


// Which nodes are in the query's subtree: one bit per node in the tree's level-ordered bit space
// (DevTree: per level a block for its internal nodes and a block for its leaves, each in node-id
// order).  Because the level's records are created in that same order, a node's compact index is
// the start of its level plus its rank inside the block, and an observed leaf's position in the
// level-sorted observation list is the start of its level plus its rank: the rank (per-word
// prefix count + popcount) replaces a node -> index table.  The space is a few KB for a 10^4-leaf
// tree and lives in LDS; larger trees keep it in per-team global scratch.
struct NodeBits {
    unsigned long long *bm;
    uint32_t *pre;  // per word: set bits of the same block in earlier words
    // child descriptor of the node with tree record nc: > 0 internal (compact index + 1),
    // <= -2 observed leaf (-(j+2)), 0 not in the subtree
    __device__ __forceinline__ int desc_at(int lpos, bool leaf, int base_int, int lo_leaf) const {
        const int w = lpos >> 6, b = lpos & 63;
        const unsigned long long word = bm[w];
        if (!((word >> b) & 1ull)) return 0;
        const int r = (int)pre[w] + __popcll(word & ((1ull << b) - 1ull));
        return leaf ? -(lo_leaf + r) - 2 : base_int + r + 1;
    }
    __device__ __forceinline__ int desc(const NodeRec &nc, int base_int, int lo_leaf) const {
        return desc_at(nc.lpos, nc.nchild == 0, base_int, lo_leaf);
    }
    __device__ __forceinline__ void set(int lpos) const { atomicOr(&bm[lpos >> 6], 1ull << (lpos & 63)); }
};

// Big trees (the bit space would not fit in LDS): a per-team node -> descriptor table in global
// scratch.  Entries carry the query's tag in their high bits, so the table is never cleared
// between queries (unroll_changes becomes a counter increment; wiped once when the tags run out).
struct NodeMap {
    uint32_t *m;
    uint32_t ver;
    int vb;
    __device__ __forceinline__ int get(int v) const {
        const uint32_t w = m[v];
        if ((w >> vb) != ver) return 0;
        const uint32_t p = w & ((1u << vb) - 1u);
        if (p == 0) return 0;  // claimed, index not stored yet: never read in that state
        return (p & 1u) ? -(int)(p >> 1) - 1 : (int)(p >> 1);
    }
    __device__ __forceinline__ void set_internal(int v, int idx) const { m[v] = (ver << vb) | ((uint32_t)(idx + 1) << 1); }
    __device__ __forceinline__ void set_leaf(int v, int j) const { m[v] = (ver << vb) | ((uint32_t)(j + 1) << 1) | 1u; }
    // first child to report a parent in this query (tags only grow, so one atomic max decides)
    __device__ __forceinline__ bool claim(int v) const { return atomicMax(&m[v], ver << vb) < (ver << vb); }
};

// Position of a registering lane in the next level's list: lanes of a wavefront in lane order, one
// LDS add per wavefront for the block of positions.  Must be reached by the whole wavefront.
__device__ __forceinline__ int ordered_slot(bool claim, int lane, int *counter) {
    const unsigned long long m = __ballot(claim);
    int wb = 0;
    if (lane == 0 && m) wb = atomicAdd(counter, __popcll(m));
    wb = __shfl(wb, 0, WAVE);
    return wb + __popcll(m & ((1ull << lane) - 1ull));
}

// position of the k-th (0-based) set bit of x
__device__ __forceinline__ int select64(unsigned long long x, int k) {
    int pos = 0;
    int c = __popc((uint32_t)x);
    if (k >= c) { k -= c; x >>= 32; pos = 32; }
    uint32_t v = (uint32_t)x;
    c = __popc(v & 0xffffu); if (k >= c) { k -= c; v >>= 16; pos += 16; }
    c = __popc(v & 0xffu);   if (k >= c) { k -= c; v >>= 8;  pos += 8; }
    c = __popc(v & 0xfu);    if (k >= c) { k -= c; v >>= 4;  pos += 4; }
    c = __popc(v & 0x3u);    if (k >= c) { k -= c; v >>= 2;  pos += 2; }
    if (k >= (int)(v & 1u)) pos += 1;
    return pos;
}

// bit position of the k-th node of the block occupying words [w0, w1): the last word whose prefix
// count is <= k holds it
__device__ __forceinline__ int kth_in_block(const NodeBits &nb, int w0, int w1, int k) {
    int lo = w0, hi = w1;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if ((int)nb.pre[mid] <= k) lo = mid; else hi = mid;
    }
    return lo * 64 + select64(nb.bm[lo], k - (int)nb.pre[lo]);
}

// per-word ranks of the block [w0, w1); returns the number of set bits.  One wavefront scans.
template <int TEAM>
__device__ int block_ranks(const NodeBits &nb, int w0, int w1, int tid, int *sh_slot) {
    int run = 0;
    if (TEAM == WAVE || tid < WAVE) {
        const int lane = tid & (WAVE - 1);
        for (int wb = w0; wb < w1; wb += WAVE) {
            const int w = wb + lane;
            const int c = (w < w1) ? __popcll(nb.bm[w]) : 0;
            int incl = c;
#pragma unroll
            for (int o = 1; o < WAVE; o <<= 1) {
                const int t = __shfl_up(incl, o, WAVE);
                if (lane >= o) incl += t;
            }
            if (w < w1) nb.pre[w] = (uint32_t)(run + incl - c);
            run += __shfl(incl, WAVE - 1, WAVE);
        }
        if (TEAM != WAVE && tid == 0) *sh_slot = run;
    }
    if (TEAM != WAVE) {
        __syncthreads();
        run = *sh_slot;
        __syncthreads();
    }
    return run;
}

struct Kid {
    double S[6];
    double e;
    int32_t node;
};

// a child's S tuple from its descriptor; node id and edge length come from the tree record
template <int M>
__device__ __forceinline__ void load_kid(int kd, int node, double e, const Rec *__restrict__ rec,
                                         const double *__restrict__ o_dist, Kid &k) {
    k.node = node;
    k.e = e;
    if (kd > 0) {
        const Rec &r = rec[kd - 1];
#pragma unroll
        for (int x = 0; x < 6; ++x) k.S[x] = r.T[x];
    } else {
        leaf_tuple<M>(o_dist[-kd - 2], k.S);
    }
}

// the same, when the child level's records are still in the team's LDS staging area
template <int M>
__device__ __forceinline__ void load_kid_staged(int kd, int node, double e, const Rec *__restrict__ rec, const uint4 *tstage,
                                                int kid_base, bool staged, const double *__restrict__ o_dist, Kid &k) {
    if (staged && kd > 0) {
        k.node = node;
        k.e = e;
        const double2 *p = reinterpret_cast<const double2 *>(tstage + (size_t)(kd - 1 - kid_base) * 4);
        const double2 a = p[0], b = p[1], c = p[2];
        k.S[0] = a.x; k.S[1] = a.y; k.S[2] = b.x; k.S[3] = b.y; k.S[4] = c.x; k.S[5] = c.y;
    } else {
        load_kid<M>(kd, node, e, rec, o_dist, k);
    }
}

// One team = TEAM threads working on one query: a wavefront (TEAM == 64, four independent teams
// per workgroup; the level loops need no s_barrier) or the whole workgroup (TEAM == 256).  Queries
// whose subtree does not fit a team's scratch (`cap` internal nodes) are appended to an overflow
// list that a second launch with full-size scratch takes.
// LDS of one workgroup (shared by the team shapes a kernel instantiates)
struct SweepShared {
    // per-wavefront staging area for 64 records: they are built one per lane but stored to HBM as
    // whole 1-KiB rows (4 store instructions per 64 records instead of 256 16-byte partial writes)
    uint4 stage[APPLES_TPB / WAVE][WAVE * 4];
    // libm pow tables (5 KiB): two lookups per candidate edge would otherwise be ten scattered
    // global loads
    double pow[384 + 256];
    double d[4];
    int cnt[APPLES_TPB / WAVE][4];
    int i[4];
    int w[APPLES_TPB / WAVE];
    int mk[APPLES_TPB / WAVE][2][WAVE];  // merge layout: the two 64-key windows of a wavefront's merge step (a workgroup-sized team: two 256-key windows)
    int mx[2][APPLES_TPB];               // merge layout, workgroup-sized team: parents and descriptors of a step's keys
    int mcnt[2][APPLES_TPB / WAVE];      // ... per-wavefront counts of a step
};

__device__ __forceinline__ void sweep_shared_init(SweepShared &sh) {
    for (int i = threadIdx.x; i < 384; i += APPLES_TPB) sh.pow[i] = (&kPowLogTab[0][0])[i];
    for (int i = threadIdx.x; i < 256; i += APPLES_TPB) sh.pow[384 + i] = __longlong_as_double((long long)kExpTab[i]);
    __syncthreads();
}

// Merge layout (big binary trees, wavefront-sized teams): the next level's list of internal nodes without a node
// map.  The nodes of one level sorted by node id (= post-order) have their siblings next to each other and their
// parents in sorted order, so the parents of this level's internal nodes (ent[base .. base + nA), sorted) and of
// its observed leaves (o_node[lo .. lo + nB), sorted) merged by node id are the next list in sorted order, and a
// parent's valid children are the run of (at most two) neighbours that name it.  One wavefront merges 64 keys
// per step through two windows in LDS (merge path: lane p finds the p-th smallest by a binary search on the
// diagonal); the first two keys of a run are never separated: a step whose last key opens a run leaves it to the
// next step (longer runs -- polytomies -- may continue into the next step; their parents look their children up
// by binary search in the sorted lists instead of reading them from the entry).
// Entry = {node, first valid child, second valid child or 0, node id of the first valid child}; child descriptors as
// everywhere (> 0: compact index + 1 of an internal node, <= -2: -(j + 2) for observed leaf j).  Returns the number
// of entries written at ent[next_base ...].
__device__ __forceinline__ int merge_parents(int4 *__restrict__ ent, int base, int nA, const int32_t *__restrict__ o_node,
                                             int lo, int nB, int next_base, const int32_t *__restrict__ parent_of,
                                             int *mk_a, int *mk_b, int lane) {
    int out = 0, ia = 0, ib = 0;
    int carry = -2;  // parent of the last key of the previous step (a polytomy's run may continue across steps)
    const unsigned long long below = (1ull << lane) - 1ull;
    while (ia < nA || ib < nB) {
        const int rem = (nA - ia) + (nB - ib);
        const int wa = min(nA - ia, WAVE), wb = min(nB - ib, WAVE);
        mk_a[lane] = lane < wa ? ent[base + ia + lane].x : 0x7fffffff;
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
        const int par = active ? parent_of[key] : -3;
        const int prev = __shfl_up(par, 1, WAVE);
        const bool first = active && par != (lane == 0 ? carry : prev);
        const int last_first = __shfl(first ? 1 : 0, tot - 1, WAVE);
        const int use = (rem > tot && last_first && tot > 1) ? tot - 1 : tot;
        const int next_desc = __shfl_down(desc, 1, WAVE), next_first = __shfl_down(first ? 1 : 0, 1, WAVE);
        const bool mine = first && lane < use;
        const unsigned long long fm = __ballot(mine);
        if (mine) ent[next_base + out + __popcll(fm & below)] = make_int4(par, desc, (lane + 1 < use && !next_first) ? next_desc : 0, key);
        const int ca = __popcll(__ballot(lane < use && from_a));
        carry = __shfl(par, use - 1, WAVE);
        ia += ca;
        ib += use - ca;
        out += __popcll(fm);
        __builtin_amdgcn_wave_barrier();
    }
    return out;
}

// The same for a workgroup-sized team: 256 keys per step, neighbours and counts through LDS instead of shuffles.
__device__ __forceinline__ int merge_parents_wg(int4 *__restrict__ ent, int base, int nA, const int32_t *__restrict__ o_node,
                                                int lo, int nB, int next_base, const int32_t *__restrict__ parent_of,
                                                SweepShared &sh) {
    int *mk_a = &sh.mk[0][0][0], *mk_b = mk_a + APPLES_TPB;
    int *m_par = sh.mx[0], *m_desc = sh.mx[1];
    const int tid = threadIdx.x, lane = tid & (WAVE - 1), wave = tid / WAVE;
    int out = 0, ia = 0, ib = 0;
    int carry = -2;
    const unsigned long long below = (1ull << lane) - 1ull;
    while (ia < nA || ib < nB) {
        const int rem = (nA - ia) + (nB - ib);
        const int wa = min(nA - ia, APPLES_TPB), wb = min(nB - ib, APPLES_TPB);
        mk_a[tid] = tid < wa ? ent[base + ia + tid].x : 0x7fffffff;
        mk_b[tid] = tid < wb ? o_node[lo + ib + tid] : 0x7fffffff;
        __syncthreads();
        const int tot = min(wa + wb, APPLES_TPB);
        const bool active = tid < tot;
        int i_lo = max(0, tid - wb), i_hi = min(tid, wa);
        while (i_lo < i_hi) {
            const int i = (i_lo + i_hi) >> 1;
            if (mk_a[i] < mk_b[tid - 1 - i]) i_lo = i + 1; else i_hi = i;
        }
        const int i = i_lo, j = tid - i_lo;
        const int ka = i < wa ? mk_a[i] : 0x7fffffff, kb = j < wb ? mk_b[j] : 0x7fffffff;
        const bool from_a = ka < kb;
        const int key = from_a ? ka : kb;
        const int desc = from_a ? base + ia + i + 1 : -(lo + ib + j) - 2;
        const int par = active ? parent_of[key] : -3;
        m_par[tid] = par;
        m_desc[tid] = desc;
        __syncthreads();
        const bool first = active && par != (tid == 0 ? carry : m_par[tid - 1]);
        const bool last_first = m_par[tot - 1] != (tot == 1 ? carry : m_par[tot - 2]);
        const int use = (rem > tot && last_first && tot > 1) ? tot - 1 : tot;
        const bool next_first = tid + 1 >= tot || m_par[tid + 1] != par;
        const int next_desc = tid + 1 < APPLES_TPB ? m_desc[tid + 1] : 0;
        const bool mine = first && tid < use;
        const unsigned long long fm = __ballot(mine), am = __ballot(tid < use && from_a);
        if (lane == 0) { sh.mcnt[0][wave] = __popcll(fm); sh.mcnt[1][wave] = __popcll(am); }
        __syncthreads();
        int before = 0, total = 0, ca = 0;
#pragma unroll
        for (int w = 0; w < APPLES_TPB / WAVE; ++w) {
            if (w < wave) before += sh.mcnt[0][w];
            total += sh.mcnt[0][w];
            ca += sh.mcnt[1][w];
        }
        if (mine) ent[next_base + out + before + __popcll(fm & below)] = make_int4(par, desc, (tid + 1 < use && !next_first) ? next_desc : 0, key);
        carry = m_par[use - 1];
        ia += ca;
        ib += use - ca;
        out += total;
        __syncthreads();
    }
    return out;
}

template <int M, int TEAM>
__device__ void sweep_team(const SweepArgs &a, int64_t nq, SweepShared &sh) {
    constexpr int TEAMS_PER_WG = APPLES_TPB / TEAM;
    constexpr bool BME = (M == APPLES_BME);
    double *sh_d = sh.d;
    int *sh_i = sh.i;
    uint4 *stage = sh.stage[threadIdx.x / WAVE];
    uint4 *tstage = TEAM == WAVE ? stage : &sh.stage[0][0];  // the team's staging area, one 64-byte slot per thread
    const int lane = threadIdx.x & (WAVE - 1);
    const double *lds_pow = sh.pow;
    const int team_in_wg = threadIdx.x / TEAM;
    const int tid = threadIdx.x % TEAM;
    int *sh_cnt = sh.cnt[team_in_wg];
    const DevTree &T = a.tree;
    const NodeRec *__restrict__ NR = T.rec;
    const int64_t nn = T.n_nodes;
    const int64_t cap = a.cap;  // scratch capacity of this launch's teams, in internal nodes
    const int64_t team = (int64_t)blockIdx.x * TEAMS_PER_WG + team_in_wg;
    // Two ways to know which nodes are in the subtree.  Small trees: the level-ordered bit space,
    // a few KB of LDS per team.  Big trees: the tagged node map in global scratch.
    const bool umap = a.map != nullptr;
    const bool umerge = a.ent != nullptr;  // level lists by merging (merge_parents), no map, no bits
    int4 *ent = umerge ? a.ent + team * (cap + 1) : nullptr;
    int *mk_a = sh.mk[threadIdx.x / WAVE][0], *mk_b = sh.mk[threadIdx.x / WAVE][1];
    NodeBits nb;
    NodeMap map;
    const int bm_words = T.bm_words;
    int32_t *order = nullptr;  // big trees: node ids in compact order
    if (umap) {
        map.m = a.map + team * nn;
        map.vb = a.map_bits;
        map.ver = a.map_ver[team];
        order = a.order + team * (cap + 1);
        nb.bm = nullptr; nb.pre = nullptr;
    } else if (umerge) {
        nb.bm = nullptr; nb.pre = nullptr;
        map.m = nullptr; map.vb = 0; map.ver = 0;
    } else {  // dynamic LDS: [teams of this workgroup][bm_words] words, then the ranks
        extern __shared__ unsigned long long dyn_lds[];
        nb.bm = dyn_lds + (size_t)team_in_wg * bm_words;
        nb.pre = reinterpret_cast<uint32_t *>(dyn_lds + (size_t)TEAMS_PER_WG * bm_words) + (size_t)team_in_wg * bm_words;
        map.m = nullptr; map.vb = 0; map.ver = 0;
    }
    const uint32_t ver_max = umap ? (1u << (32 - a.map_bits)) - 1u : 0u;
    const int32_t *__restrict__ lvlw = T.lvlw;
    const int2 *__restrict__ npos = reinterpret_cast<const int2 *>(T.npos);
    // descriptor of node v (tree record position lpos / leaf flag), whichever layout is in use
    const int32_t *cur_obs = nullptr;  // the current query's observed leaves (set per query)
    // (merge layout: polytomies and the HYBRID re-ranking -- by binary search in the node's level: its observed leaves
    // o_node[lo_leaf, hi_leaf) and its entries ent[base_int, end_int), both sorted by node id)
    auto desc_of = [&](int v, int lpos, bool leaf, int base_int, int lo_leaf, int end_int, int hi_leaf) -> int {
        if (umerge) {
            int l = leaf ? lo_leaf : base_int, h = leaf ? hi_leaf : end_int;
            while (l < h) {
                const int mid = (l + h) >> 1;
                const int key = leaf ? cur_obs[mid] : ent[mid].x;
                if (key < v) l = mid + 1; else h = mid;
            }
            const int end = leaf ? hi_leaf : end_int;
            if (l >= end) return 0;
            const int key = leaf ? cur_obs[l] : ent[l].x;
            if (key != v) return 0;
            return leaf ? -l - 2 : l + 1;
        }
        return umap ? map.get(v) : nb.desc_at(lpos, leaf, base_int, lo_leaf);
    };
    Rec *rec = reinterpret_cast<Rec *>(a.A) + team * (cap + 1);
    // R values of a polytomy's children wait here until all of them are formed (in-place update
    // would destroy sibling S values that are still needed); unused for binary trees
    double *rtmp = a.B ? reinterpret_cast<double *>(a.B) + team * (cap + 1) * 6 : nullptr;
    int32_t *grp_off = a.grp_off + team * (T.height + 4);
    double *xe = a.xe ? a.xe + team * (cap + a.leaf_cap) * XE_STRIDE : nullptr;
    // work queue: size-class lists written by the selection kernel (small teams), a device-side
    // list (routed / overflow queries), or simply 0..nq-1
    int c0 = 0, c1 = 0, c2 = 0, c3 = 0;
    int64_t n_work;
    if (a.cls_list) {
        c0 = a.cls_count[0]; c1 = a.cls_count[1]; c2 = a.cls_count[2]; c3 = a.cls_count[3];
        n_work = (int64_t)c0 + c1 + c2 + c3;
    } else {
        n_work = a.work_count ? *a.work_count : nq;
    }
    int *sh_w = sh.w;
    while (true) {
        // dynamic scheduling: one atomic add per query, broadcast to the team
        if (tid == 0) sh_w[team_in_wg] = atomicAdd(a.cursor, 1);
        team_sync<TEAM>();
        const int64_t w = sh_w[team_in_wg];
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
        cur_obs = o_node;
        const double *o_dist = a.obs_dist + row_start(a.row_off, q, a.obs_cap);
        const int32_t *cg = a.cnt_gt + q * (int64_t)(T.height + 2);

        // ------------------------------------------------------------ bottom-up: mark + S values
        const int lvl_first = T.level[o_node[0]];
        int lvl = lvl_first;
        int base = 0, kid_base = 0, n_par = 0, G = 0, lca = -1;
        bool overflow = false;
        bool prev_staged = false;  // the level below fitted one pass: its records are still in LDS
        if (umap) {
            // a fresh tag for this query's map entries; when the tags run out, wipe the table once
            if (map.ver == ver_max) {
                team_sync<TEAM>();
                for (int64_t i = tid; i < nn; i += TEAM) map.m[i] = 0;
                map.ver = 0;
                team_sync<TEAM>();
            }
            ++map.ver;
            if (tid < 3) sh_cnt[tid] = 0;
            for (int j = tid; j < n; j += TEAM) map.set_leaf(o_node[j], j);
        } else if (umerge) {
            if (tid < 3) sh_cnt[tid] = 0;
        } else {
            for (int i = tid; i < bm_words; i += TEAM) nb.bm[i] = 0;  // unroll_changes of the previous query
            team_sync<TEAM>();
            // every observed leaf marks itself and its parent (a parent is in the subtree as soon as
            // one child is; setting a bit twice is harmless, so nobody has to be "the" registering child)
            for (int j = tid; j < n; j += TEAM) {
                const int2 np = npos[o_node[j]];
                nb.set(np.x);
                if (np.y >= 0 && n > 1) nb.set(np.y);
            }
        }
        team_sync<TEAM>();
        while (true) {
            const int lo = cg[lvl + 1], hi = cg[lvl];  // observed leaves of this level: obs[lo, hi)
            const int n_leaf = hi - lo;
            int w0 = 0, w1 = 0;
            if (!umap && !umerge) {
                // all children of this level's nodes have reported: rank the level's two blocks
                w0 = lvlw[2 * lvl]; w1 = lvlw[2 * lvl + 1];
                const int wl1 = lvlw[2 * lvl + 2];
                n_par = block_ranks<TEAM>(nb, w0, w1, tid, &sh_cnt[0]);
                block_ranks<TEAM>(nb, w1, wl1, tid, &sh_cnt[0]);
                team_sync<TEAM>();
            }
            if (n_par + n_leaf == 1 && hi == n) {  // one node left in the frontier: the LCA (Subtree.py:36-43)
                lca = (n_par == 1) ? (umap ? order[base] : umerge ? ent[base].x : T.lnode[kth_in_block(nb, w0, w1, 0)]) : o_node[lo];
                break;
            }
            if (cap < nn && (int64_t)base + (umap || umerge ? 2 * (int64_t)n_par + n_leaf : (int64_t)n_par) > cap) { overflow = true; break; }
            if (tid == 0) { grp_off[G] = base; if (umap) sh_cnt[(G + 1) % 3] = 0; }
            int *next_cnt = &sh_cnt[G % 3];
            const int next_base = base + n_par;
            if (umap) {  // observed leaves of this level: the first child to report registers the parent
                for (int k0 = 0; k0 < n_leaf; k0 += TEAM) {  // team-uniform trip count
                    const int k = k0 + tid;
                    int parent = -1;
                    bool claimer = false;
                    if (k < n_leaf) {
                        parent = NR[o_node[lo + k]].parent;
                        claimer = parent >= 0 && map.claim(parent);
                    }
                    const int nidx = next_base + ordered_slot(claimer, lane, next_cnt);
                    if (claimer) { order[nidx] = parent; map.set_internal(parent, nidx); }
                }
            }
            const int lo_kids = lvl + 2 <= T.height + 1 ? cg[lvl + 2] : 0;  // first observed leaf of the level below
            // internal nodes of this level, in node-id order: S tuple from the valid children in
            // file order, then tell the parent
            // children's S tuples come from LDS when both levels are single-pass (the staging area
            // still holds the level below, and nothing overwrites it before everybody has read)
            const bool staged = prev_staged && n_par <= TEAM;
            for (int k0 = 0; k0 < n_par; k0 += TEAM) {  // team-uniform trip count
                const int k = k0 + tid;
                const bool active = k < n_par;
                int parent = -1;
                Rec r;
                if (active) {
                    // (bit space: the k-th set bit's position leads straight to the record)
                    int4 e = make_int4(0, 0, 0, 0);
                    if (umerge) e = ent[base + k];
                    const NodeRec nr = umap ? NR[order[base + k]] : umerge ? NR[e.x] : T.rec_l[kth_in_block(nb, w0, w1, k)];
                    r.node = nr.node;
                    parent = nr.parent;
#pragma unroll
                    for (int c = 0; c < 6; ++c) r.T[c] = 0;
                    if (nr.nchild <= 2) {
                        int m0, m1;
                        if (umerge) {  // the entry names the valid children; a single one is the first or the second child
                            m0 = (e.z != 0 || e.w == nr.c0) ? e.y : 0;
                            m1 = e.z != 0 ? e.z : (e.w == nr.c0 ? 0 : e.y);
                        } else {
                            m0 = nr.nchild >= 1 ? desc_of(nr.c0, nr.c0pos, nr.kleaf & 1u, kid_base, lo_kids, base, lo) : 0;
                            m1 = nr.nchild >= 2 ? desc_of(nr.c1, nr.c1pos, nr.kleaf & 2u, kid_base, lo_kids, base, lo) : 0;
                        }
                        const int nk = (m0 != 0) + (m1 != 0);
                        const double coef = BME ? 1.0 / (double)nk : 1.0;  // apples/BME.py:20
                        r.k0 = m0 ? m0 : m1;
                        r.k1 = (m0 && m1) ? m1 : 0;
                        r.meta = (uint32_t)nk | ((!m0) ? META_K0C1 : 0u);
                        {
                            Kid kd;
                            load_kid_staged<M>(r.k0, m0 ? nr.c0 : nr.c1, m0 ? nr.e0 : nr.e1, rec, tstage, kid_base, staged, o_dist, kd);
                            double t[6];
                            lift<M>(kd.S, kd.e, t);
#pragma unroll
                            for (int x = 0; x < 6; ++x) r.T[x] += BME ? coef * t[x] : t[x];
                        }
                        if (nk > 1) {
                            Kid kd;
                            load_kid_staged<M>(r.k1, nr.c1, nr.e1, rec, tstage, kid_base, staged, o_dist, kd);
                            double t[6];
                            lift<M>(kd.S, kd.e, t);
#pragma unroll
                            for (int x = 0; x < 6; ++x) r.T[x] += BME ? coef * t[x] : t[x];
                        }
                    } else {
                        int nk = 0;
                        r.k0 = r.k1 = 0;
                        const int cb = T.child_off[r.node], ce = cb + nr.nchild;
                        for (int ci = cb; ci < ce; ++ci) {
                            const int mc = desc_of(T.child_idx[ci], npos[T.child_idx[ci]].x, NR[T.child_idx[ci]].nchild == 0, kid_base, lo_kids, base, lo);
                            if (mc != 0) { if (nk == 0) r.k0 = mc; else if (nk == 1) r.k1 = mc; ++nk; }
                        }
                        const double coef = BME ? 1.0 / (double)nk : 1.0;
                        for (int ci = cb; ci < ce; ++ci) {
                            const int cn = T.child_idx[ci];
                            const NodeRec cr = NR[cn];
                            const int mc = desc_of(cn, cr.lpos, cr.nchild == 0, kid_base, lo_kids, base, lo);
                            if (mc != 0) {
                                Kid kd;
                                load_kid<M>(mc, cn, cr.e, rec, o_dist, kd);
                                double t[6];
                                lift<M>(kd.S, kd.e, t);
#pragma unroll
                                for (int x = 0; x < 6; ++x) r.T[x] += BME ? coef * t[x] : t[x];
                            }
                        }
                        r.meta = (uint32_t)nk | META_POLY;
                    }
                    if (!umap && !umerge && nr.ppos >= 0) nb.set(nr.ppos);  // tell the parent
                }
                if (staged) team_sync<TEAM>();  // everybody has read the level below
                __builtin_amdgcn_wave_barrier();
                if (active) {
                    // stage the record; its 64 bytes leave as part of a 1-KiB row below
                    const uint4 *src = reinterpret_cast<const uint4 *>(&r);
#pragma unroll
                    for (int x = 0; x < 4; ++x) stage[lane * 4 + x] = src[x];
                }
                __builtin_amdgcn_wave_barrier();
                {
                    const int wave_k0 = k0 + (tid - lane);         // first record of this wavefront's 64
                    const int n_here = n_par - wave_k0;            // records this wavefront holds (may be <= 0)
                    uint4 *dst = reinterpret_cast<uint4 *>(&rec[base + wave_k0]);
#pragma unroll
                    for (int x = 0; x < 4; ++x) {
                        const int c = x * WAVE + lane;             // 16-byte chunk of the 4-KiB block
                        if ((c >> 2) < n_here) dst[c] = stage[c];
                    }
                }
                __builtin_amdgcn_wave_barrier();
                if (umap) {
                    const bool claimer = active && parent >= 0 && map.claim(parent);
                    const int nidx = next_base + ordered_slot(claimer, lane, next_cnt);
                    if (claimer) { order[nidx] = parent; map.set_internal(parent, nidx); }
                }
            }
            team_sync<TEAM>();
            prev_staged = n_par > 0 && n_par <= TEAM;
            int merged = 0;
            if (umerge) merged = TEAM == WAVE ? merge_parents(ent, base, n_par, o_node, lo, n_leaf, next_base, T.parent, mk_a, mk_b, lane)
                                              : merge_parents_wg(ent, base, n_par, o_node, lo, n_leaf, next_base, T.parent, sh);
            kid_base = base;
            base = next_base;
            if (umap) n_par = *next_cnt;
            if (umerge) n_par = merged;
            ++G;
            --lvl;
        }
        if (overflow) {  // hand the query to the big-team launch (this team's marks die with the tag / the next clear)
            if (tid == 0) a.overflow_list[atomicAdd(a.overflow_count, 1)] = (int32_t)q;
            team_sync<TEAM>();
            continue;
        }
        const int VI = base;      // internal valid nodes; the LCA's record sits at index VI
        const int V = base + n;   // Subtree.num_nodes
        if (tid == 0) {
            grp_off[G] = VI;
            grp_off[G + 1] = VI + 1;
            const NodeRec nr = NR[lca];
            Rec &r = rec[VI];
            r.node = lca;
            int nk = 0, k0 = 0, k1 = 0, first = -1;
            const int klo = lvl + 2 <= T.height + 1 ? cg[lvl + 2] : 0;
            if (nr.nchild <= 2) {
                int m0, m1;
                if (umerge) {
                    const int4 e = nr.nchild > 0 ? ent[VI] : make_int4(0, 0, 0, 0);
                    m0 = (e.z != 0 || e.w == nr.c0) ? e.y : 0;
                    m1 = e.z != 0 ? e.z : (e.w == nr.c0 ? 0 : e.y);
                } else {
                    m0 = nr.nchild >= 1 ? desc_of(nr.c0, nr.c0pos, nr.kleaf & 1u, kid_base, klo, VI, cg[lvl + 1]) : 0;
                    m1 = nr.nchild >= 2 ? desc_of(nr.c1, nr.c1pos, nr.kleaf & 2u, kid_base, klo, VI, cg[lvl + 1]) : 0;
                }
                nk = (m0 != 0) + (m1 != 0);
                k0 = m0 ? m0 : m1;
                k1 = (m0 && m1) ? m1 : 0;
                first = m0 ? 0 : 1;
            } else {
                const int cb = T.child_off[lca];
                for (int ci = cb; ci < cb + nr.nchild; ++ci) {
                    const int mc = desc_of(T.child_idx[ci], npos[T.child_idx[ci]].x, NR[T.child_idx[ci]].nchild == 0, kid_base, klo, VI, cg[lvl + 1]);
                    if (mc != 0) { if (nk == 0) k0 = mc; else if (nk == 1) k1 = mc; ++nk; }
                }
            }
            r.k0 = k0; r.k1 = k1;
            r.meta = (uint32_t)nk | (nr.nchild > 2 ? META_POLY : 0u) | (first == 1 ? META_K0C1 : 0u);
        }
        team_sync<TEAM>();

        // ------------------------------------------------------------ top-down, parent-centric:
        // a node forms R for each valid child (all_R_values), solves it (placement_per_edge) and
        // evaluates its residual (error_per_edge)
        double best_key = INF_D;
        int best_v = 0x7fffffff;
        Sol best_sol;
        double best_e = 0;
        best_sol.x1 = best_sol.x2 = best_sol.err = 0; best_sol.x1_int = 0; best_sol.x1n = best_sol.x2n = 0;
#ifdef APPLES_BU_ONLY
        for (int g = 0; g >= 1; --g) {
#else
        for (int g = (a.debug_phase == 1 ? 0 : G); g >= 1; --g) {
#endif
            const int g0 = grp_off[g], g1 = grp_off[g + 1];
            for (int idx = g0 + tid; idx < g1; idx += TEAM) {
                const bool is_lca = (idx == VI);
                const Rec self = rec[idx];
                const NodeRec nr = NR[self.node];
                const int nk = (int)(self.meta & META_NK);
                // apples/BME.py:36-37: 1 / (nonroot + #valid siblings)
                const double coef = BME ? 1.0 / (double)((is_lca ? 0 : 1) + nk - 1) : 1.0;
                double plift[6];
                if (!is_lca) lift<M>(self.T, nr.e, plift);  // self.T is this node's R by now
                auto finish_kid = [&](int kd, const Kid &kid, const double *acc, bool defer) {
                    const Sol r = solve_edge<M>(kid.S, acc, kid.e, a.negative, lds_pow);
                    if (kd > 0) {
                        double *dst = defer ? rtmp + (int64_t)(kd - 1) * 6 : rec[kd - 1].T;
#pragma unroll
                        for (int x = 0; x < 6; ++x) dst[x] = acc[x];
                    }
                    if (a.keep_edges) {
                        double *xp = xe + (int64_t)(kd > 0 ? kd - 1 : cap + (-kd - 2)) * XE_STRIDE;
                        xp[0] = r.x1; xp[1] = r.x2; xp[2] = r.x1n; xp[3] = r.x2n; xp[4] = r.err;
#pragma unroll
                        for (int x = 0; x < 6; ++x) { xp[5 + x] = acc[x]; xp[11 + x] = kid.S[x]; }
                        xp[17] = r.x1_int ? 1.0 : 0.0;
                    }
                    const double key = (a.criterion == APPLES_ME) ? r.x1 : r.err;
                    if (key < best_key || (key == best_key && kid.node < best_v)) {
                        best_key = key; best_v = kid.node; best_sol = r; best_e = kid.e;
                    }
                };
                if (!(self.meta & META_POLY)) {
                    Kid kid[2];
                    if (nk > 0) {
                        const bool second = (self.meta & META_K0C1) != 0;
                        load_kid<M>(self.k0, second ? nr.c1 : nr.c0, second ? nr.e1 : nr.e0, rec, o_dist, kid[0]);
                    }
                    if (nk > 1) load_kid<M>(self.k1, nr.c1, nr.e1, rec, o_dist, kid[1]);
#pragma unroll
                    for (int z = 0; z < 2; ++z) {
                        if (z < nk) {
                            double acc[6];
#pragma unroll
                            for (int x = 0; x < 6; ++x) acc[x] = 0;
                            if (nk > 1) {  // the one valid sibling (apples/OLS.py:59-69)
                                double t[6];
                                lift<M>(kid[1 - z].S, kid[1 - z].e, t);
#pragma unroll
                                for (int x = 0; x < 6; ++x) acc[x] += BME ? coef * t[x] : t[x];
                            }
                            if (!is_lca) {  // parent term last (apples/OLS.py:70-80)
#pragma unroll
                                for (int x = 0; x < 6; ++x) acc[x] += BME ? coef * plift[x] : plift[x];
                            }
                            finish_kid(z == 0 ? self.k0 : self.k1, kid[z], acc, false);
                        }
                    }
                } else {  // polytomy: children through the CSR list
                    const int cb = T.child_off[self.node], ce = cb + nr.nchild;
                    const int kb = grp_off[g - 1], klo = cg[lvl_first - g + 2];  // the children's level
                    for (int ci = cb; ci < ce; ++ci) {
                        const int cn = T.child_idx[ci];
                        const NodeRec cr = NR[cn];
                        const int mc = desc_of(cn, cr.lpos, cr.nchild == 0, kb, klo, g0, cg[lvl_first - g + 1]);
                        if (mc == 0) continue;
                        double acc[6];
#pragma unroll
                        for (int x = 0; x < 6; ++x) acc[x] = 0;
                        for (int cj = cb; cj < ce; ++cj) {
                            if (cj == ci) continue;
                            const int sn = T.child_idx[cj];
                            const NodeRec sr = NR[sn];
                            const int ms = desc_of(sn, sr.lpos, sr.nchild == 0, kb, klo, g0, cg[lvl_first - g + 1]);
                            if (ms != 0) {
                                Kid sk;
                                load_kid<M>(ms, sn, sr.e, rec, o_dist, sk);
                                double t[6];
                                lift<M>(sk.S, sk.e, t);
#pragma unroll
                                for (int x = 0; x < 6; ++x) acc[x] += BME ? coef * t[x] : t[x];
                            }
                        }
                        if (!is_lca) {
#pragma unroll
                            for (int x = 0; x < 6; ++x) acc[x] += BME ? coef * plift[x] : plift[x];
                        }
                        Kid kr;
                        load_kid<M>(mc, cn, cr.e, rec, o_dist, kr);
                        finish_kid(mc, kr, acc, true);
                    }
                    for (int ci = cb; ci < ce; ++ci) {  // all siblings done: S -> R in place
                        const int mc = desc_of(T.child_idx[ci], npos[T.child_idx[ci]].x, NR[T.child_idx[ci]].nchild == 0, kb, klo, g0, cg[lvl_first - g + 1]);
                        if (mc > 0) {
#pragma unroll
                            for (int x = 0; x < 6; ++x) rec[mc - 1].T[x] = rtmp[(int64_t)(mc - 1) * 6 + x];
                        }
                    }
                }
            }
            team_sync<TEAM>();
        }

        // ------------------------------------------------------------ selection (apples/Algorithm.py:74-91)
        int win;
        const int my_best = best_v;
        if (a.criterion == APPLES_HYBRID) {
            // nsmallest(floor(log2(num_nodes))) by error (stable = ties to the smaller edge_index),
            // then the first minimum of x_1 among them in that order.  Candidates: internal nodes
            // (xe slot = compact index) and observed leaves (xe slot = cap + j).
            const int kk = 31 - __clz(V);
            double last_e = -INF_D;
            int last_v = -1;
            double bx = INF_D;
            int win_slot = -1;
            win = -1;
            for (int r = 0; r < kk; ++r) {
                double ke = INF_D;
                int kv = 0x7fffffff;
                for (int i = tid; i < V; i += TEAM) {
                    const int64_t slot = i < VI ? i : cap + (i - VI);
                    const int v = i < VI ? rec[i].node : o_node[i - VI];
                    const double e = xe[slot * XE_STRIDE + 4];
                    const bool after = (e > last_e) || (e == last_e && v > last_v);
                    if (after && (e < ke || (e == ke && v < kv))) { ke = e; kv = v; }
                }
                team_argmin<TEAM>(ke, kv, sh_d, sh_i);
                if (kv == 0x7fffffff) break;
                last_e = ke; last_v = kv;
                const int kl = T.level[kv];
                const int mk = desc_of(kv, npos[kv].x, NR[kv].nchild == 0, grp_off[lvl_first - kl], cg[kl + 1], grp_off[lvl_first - kl + 1], cg[kl]);
                const int64_t slot = mk > 0 ? mk - 1 : cap + (-mk - 2);
                const double x1 = xe[slot * XE_STRIDE + 0];
                if (win < 0 || x1 < bx) { bx = x1; win = kv; win_slot = (int)slot; }
            }
            if (win >= 0 && tid == 0) {  // the winner's stored solution
                const double *xp = xe + (int64_t)win_slot * XE_STRIDE;
                best_sol.x1 = xp[0]; best_sol.x2 = xp[1]; best_sol.err = xp[4];
                best_sol.x1_int = xp[17] != 0.0;
                best_e = NR[win].e;
            }
        } else {
            team_argmin<TEAM>(best_key, best_v, sh_d, sh_i);
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
        if (tid == 0) { grp_off[T.height + 3] = lca; grp_off[T.height + 2] = VI; }
        team_sync<TEAM>();
    }
    if (umap && tid == 0) a.map_ver[team] = map.ver;
}


#ifndef APPLES_SWEEP_WAVES
#define APPLES_SWEEP_WAVES 2
#endif

template <int M, int TEAM>
__global__ __launch_bounds__(APPLES_TPB, APPLES_SWEEP_WAVES) void k_sweep(SweepArgs a, int64_t nq) {
    __shared__ SweepShared sh;
    sweep_shared_init(sh);
    sweep_team<M, TEAM>(a, nq, sh);
}

// One launch for a whole batch: the first `n_big` workgroups first serve, as workgroup-sized teams
// with full-size scratch, the queries the selection kernel routed to them (many observed leaves:
// the longest jobs start first), then every workgroup splits into four wavefront-sized teams that
// drain the size-class queues.
template <int M>
__global__ __launch_bounds__(APPLES_TPB, APPLES_SWEEP_WAVES) void k_sweep_mixed(SweepArgs small, SweepArgs big, int64_t nq, int n_big) {
    __shared__ SweepShared sh;
    sweep_shared_init(sh);
    if ((int)blockIdx.x < n_big) {
        sweep_team<M, APPLES_TPB>(big, nq, sh);
        __syncthreads();
    }
    sweep_team<M, WAVE>(small, nq, sh);
}

bool sweep_merge_lists(const DevTree &t) {
    // switches (apples_params.debug / environment): NO_SWEEP_MERGE = the tagged node map for big trees as before;
    // SWEEP_MERGE = the merge layout also where the node bits would fit in LDS (tests run it on small trees)
    const bool off = (t.dbg & APPLES_DBG_NO_SWEEP_MERGE) != 0, force = (t.dbg & APPLES_DBG_SWEEP_MERGE) != 0;
    if (off || t.scan || !t.merge_ok || (t.dbg & APPLES_DBG_NODE_MAP)) return false;
    if (force || (size_t)4 * t.bm_words * 12 > 40 * 1024) return true;
    // ... and from 2 048 nodes up wherever sweep_lean.hip can serve the tree: its sweep beats the level loop over node bits in
    // LDS by 4-13 % (BASELINE config 2's shape at 1 000 / 2 000 / 5 000 / 10 000 / 20 000 leaves: sweep 0.78 -> 0.75, 1.00 -> 0.96,
    // 1.22 -> 1.15, 1.40 -> 1.28, 1.52 -> 1.32 ms; at 500 leaves the node bits win, 0.72 against 0.79:
    // profiles/r04_small_tree_sweep_exp.txt).  HYBRID keeps the node bits on such trees (its per-edge records need the level loop).
    return t.n_nodes >= 2048 && t.lean_small && t.pe != nullptr && !(t.dbg & APPLES_DBG_NO_SWEEP_LEAN);
}

// sweep_lean.hip serves the teams of a big tree in the merge layout (any number of children per node since round 6 -- child
// records, lean_poly_S / lean_poly_td -- and any height: the per-level offsets are a window in LDS), without per-edge
// records -- inspection keeps the level loop above.  APPLES_NO_SWEEP_LEAN: the level loop everywhere.
bool sweep_lean_layout(const DevTree &t, bool per_edge_records) {
    return sweep_merge_lists(t) && !per_edge_records && t.pe != nullptr && !(t.dbg & APPLES_DBG_NO_SWEEP_LEAN);
}

bool sweep_bits_in_lds(const DevTree &t) {
    if (t.dbg & APPLES_DBG_NODE_MAP) return false;  // test knob: exercise the big-tree layout on a small tree
    if (sweep_merge_lists(t)) return false;  // (merged lists: forced, or chosen for a tree the lean sweep serves)
    // (a big tree whose numbering the merged lists cannot take keeps the tagged node map)
    return (size_t)4 * t.bm_words * 12 <= 40 * 1024;
}
static size_t dyn_lds_bytes(const DevTree &t, int teams_per_wg) {
    return sweep_bits_in_lds(t) ? (size_t)teams_per_wg * t.bm_words * 12 : 0;
}

int launch_sweep_mixed(apples_ctx *ctx, const SweepArgs &small, const SweepArgs &big, int64_t nq, int wgs, int n_big,
                       hipStream_t st) {
    if (nq == 0) return 0;
    int64_t need = (nq + 3) / 4;
    // every workgroup indexes the small teams' scratch by blockIdx.x, so the grid never exceeds `wgs`;
    // big-team duty falls to the first min(n_big, grid) workgroups
    dim3 grid((unsigned)std::min<int64_t>(std::max<int64_t>(need, std::min<int64_t>(n_big, nq)), wgs)), block(APPLES_TPB);
    if (n_big > (int)grid.x) n_big = (int)grid.x;
    // (APPLES_SWEEP_LDS_PAD: occupancy experiments -- extra dynamic LDS per workgroup limits how many are resident per CU)
    const size_t lds_pad = (size_t)knob(ctx, "APPLES_SWEEP_LDS_PAD", 0);
    const size_t dyn = dyn_lds_bytes(small.tree, 4) + lds_pad;
    switch (small.method) {
        case APPLES_FM: hipLaunchKernelGGL((k_sweep_mixed<APPLES_FM>), grid, block, dyn, st, small, big, nq, n_big); break;
        case APPLES_BME: hipLaunchKernelGGL((k_sweep_mixed<APPLES_BME>), grid, block, dyn, st, small, big, nq, n_big); break;
        case APPLES_BE: hipLaunchKernelGGL((k_sweep_mixed<APPLES_BE>), grid, block, dyn, st, small, big, nq, n_big); break;
        default: hipLaunchKernelGGL((k_sweep_mixed<APPLES_OLS>), grid, block, dyn, st, small, big, nq, n_big); break;
    }
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}

template <int TEAM>
static void launch_sweep_team(apples_ctx *ctx, const SweepArgs &a, int64_t nq, int wgs, hipStream_t st) {
    dim3 grid((unsigned)wgs), block(APPLES_TPB);
    const size_t dyn = dyn_lds_bytes(a.tree, APPLES_TPB / TEAM);
    switch (a.method) {
        case APPLES_FM: hipLaunchKernelGGL((k_sweep<APPLES_FM, TEAM>), grid, block, dyn, st, a, nq); break;
        case APPLES_BME: hipLaunchKernelGGL((k_sweep<APPLES_BME, TEAM>), grid, block, dyn, st, a, nq); break;
        case APPLES_BE: hipLaunchKernelGGL((k_sweep<APPLES_BE, TEAM>), grid, block, dyn, st, a, nq); break;
        default: hipLaunchKernelGGL((k_sweep<APPLES_OLS, TEAM>), grid, block, dyn, st, a, nq); break;
    }
}

int launch_sweep(apples_ctx *ctx, const SweepArgs &a, int64_t nq, int wgs, int team, hipStream_t stream) {
    if (nq == 0) return 0;
    hipStream_t st = stream ? stream : ctx->stream;
    if (team == 64) {
        int64_t need = (nq + 3) / 4;
        launch_sweep_team<64>(ctx, a, nq, (int)(need < wgs ? need : wgs), st);
    } else {
        launch_sweep_team<256>(ctx, a, nq, (int)(nq < wgs ? nq : wgs), st);
    }
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}
