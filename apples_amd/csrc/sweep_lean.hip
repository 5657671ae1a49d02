// Least-squares placement sweep, lean form for big binary trees (the C3 route): the level loop of sweep.hip with
// the merged level lists, without tree records and with the small levels kept on chip.
//
//   apples/Subtree.py:23-43 (validate_edges), OLS.py:12-44 ... (all_S_values) -> bottom-up pass: level lists by merging,
//                                                                                S tuples as the lists form
//   apples/OLS.py:46-128 ..., util.py:6-54                                    -> top-down pass: R tuples, 2x2 solve,
//                                                                                residual, running arg-min
//   apples/Algorithm.py:62-101 (placement)                                    -> wavefront arg-min, the placement struct
//
// A query's sweep is a chain of level steps (some 45 at 200 k leaves) and a level step is a chain of dependent memory
// round trips; what the kernel takes per launch is (queries / teams in flight) x (steps x trips x latency), far from
// any bandwidth.  So this form removes trips:
//   * no tree records.  What sweep.hip gathers per internal node and pass -- a 64-byte tree record, 25.6 MB of them at
//     200 k leaves -- is one 16-byte gather per list key, {parent, edge length} (`pe`, 6.4 MB).  A list entry carries
//     everything later steps need of its (at most two) valid children: descriptors, node ids, edge lengths and, for
//     observed leaves, distances; a node's R tuple is stored already lifted over its own edge.  Entries are field
//     arrays (every load of a level is a contiguous run per wavefront) and a child's tuple sits in the level below in
//     its parent's order: near-sequential too;
//   * a list of at most 64 entries stays in the lanes' registers, the gather for its keys is issued as soon as they
//     exist and lands while the S tuples of the step are computed; the observed leaves of the next level (node,
//     parent, edge length, distance: sequential arrays, the first two filled up front by independent gathers) are
//     requested one step ahead; the per-level offsets sit in LDS.  A level that fits one merge step is then merged
//     entirely out of LDS windows and hands its entries and S tuples to the next level through LDS: no dependent
//     memory round trip in the step.
//
// Team = one wavefront per query (four per workgroup, no s_barrier).  Queries with many observed leaves are routed to
// workgroup-sized teams (k_sweep_lean_big below); per-edge inspection keeps sweep.hip's level loop (the launcher decides).
// Trees with polytomies run the PL instances of the kernels (child records, lean_poly_S / lean_poly_td); trees of more than
// LEAN_MAX_LEVELS - 2 levels keep a window of the per-level offsets in LDS.  The HYBRID criterion keeps every edge's solution in the entries' tuple slots and
// ranks them after the top-down pass (lean_hybrid_pick).  Arithmetic: sweep_math.h, shared with sweep.hip --
// same expressions in the same order, so placements are bit-identical to the level loop's.
#include <algorithm>
#include <type_traits>
#include <cstdlib>

#include "common.h"

#include "sweep_math.h"

// wavefronts per SIMD the two kernels of the wavefront-sized teams are built for (registers: 171 and 157 unconstrained;
// the bottom-up kernel's LDS, 40 KB per workgroup, would allow four workgroups per CU, its registers do not: 45 spilled)
#ifndef LEAN_UP_WAVES
#define LEAN_UP_WAVES 3
#endif
#ifndef LEAN_UP_WAVES_PL
#define LEAN_UP_WAVES_PL 2  // ... of the bottom-up kernel's instantiation for trees with polytomies: its merge keeps more alive -- 368 bytes of
                            // scratch per lane at three (48 without polytomies); at two, config 3's tree with 1 % polytomies sweeps in 18.95 ms
                            // against 20.29, unrooted 17.6 against 18.1 (profiles/r06_pl_waves_exp.txt)
#endif
#ifndef LEAN_DOWN_WAVES
#define LEAN_DOWN_WAVES 3
#endif

namespace {

// LDS of one wavefront-sized team
struct LeanWave {
    int cg[LEAN_MAX_LEVELS];   // the query's per-level offsets into its level-sorted observation list (cnt_gt)
    double2 stage[3][WAVE];    // S tuples of a level of at most 64 nodes, for the level above
    // the two windows of a merge step: keys, and for a level that fits one step also parent, edge length (and, for an
    // observed leaf, its distance) of every key
    int ka[WAVE], kb[WAVE], pa[WAVE], pb[WAVE];
    double ea[WAVE], eb[WAVE], db[WAVE];
    // the entries a one-step merge produced, in list order: the next level takes them from here
    int oK[WAVE];
    int2 oD[WAVE], oN[WAVE];
    double2 oE[WAVE], oDD[WAVE];
};

struct LeanUpShared {
    LeanWave w[APPLES_TPB / WAVE];
};

struct LeanDownShared {
    double pow[384 + 256];  // libm pow tables (sweep_math.h)
    double2 stage[APPLES_TPB / WAVE][3][WAVE];  // per wavefront: lifted R tuples on their way to a level of at most 64 nodes
};
struct LeanDownSharedPL : LeanDownShared {
    double pu[APPLES_TPB / WAVE][6][WAVE];  // per wavefront: a polytomy's children's lifted tuples (lean_poly_td)
};

// per-team scratch: one array per field, `cap1` entries each (cap1 a multiple of 4), then two per observed leaf
struct LeanTeam {
    double2 *T0, *T1, *T2;  // the node's tuple: S after the bottom-up pass, lift(R) once the top-down pass reached its parent
    double2 *E;             // edge lengths of the first and second valid child
    double2 *DD;            // their distances where they are observed leaves
    int2 *D;                // their descriptors (> 0: entry index + 1, <= -2: observed leaf -(j+2), 0: none)
    int2 *N;                // their node ids
    int32_t *K;             // node id
    double *LE;             // per observed leaf: edge length ...
    int32_t *LP;            // ... and parent (pe gathered once, up front)
    double *BP;             // clade blocks: the pool of their tuples (SweepArgs::blk_pool), or nullptr
};

// An observed "leaf" that is the root of a clade block (DevAlign::blk_*, select.hip phase 3): its distance is a boxed index -- a
// negative quiet NaN with a tag (common.h: APPLES_BLOCK_BOX) whose low 48 bits say where the block's tuple is in the pool (component x at index + 64 x): S after k_blocks_up;
// the top-down pass leaves lift(R) over the root's edge there, which k_blocks_down takes on into the block.
__device__ __forceinline__ bool lean_is_block(double dist) { return ((unsigned)__double2hiint(dist) & 0xffff0000u) == (unsigned)(APPLES_BLOCK_BOX >> 32); }
__device__ __forceinline__ long long lean_block_at(double dist) { return __double_as_longlong(dist) & 0x0000ffffffffffffLL; }

// ---- polytomies (template parameter PL; a binary tree's kernels carry none of this) ---------------------------------------
// A node with more than two valid children in a query's subtree (apples/OLS.py:36,59 loop over any number of children; a
// root trifurcation is what every unrooted Newick has).  Its entry keeps the first two as ever; the merge step that meets a
// third child appends CHILD RECORDS for all of them -- one per valid child, in file order -- to the top end of the query's
// entry range, growing downward: record x of the query sits in entry slot `xtop - x` and reuses the slot's fields,
//   D = (child's descriptor, position: 0, 1, 2, then 3 for every later one)   N = (child's node id, the polytomy's entry)
//   E = (child's edge length, its observed distance; after the top-down pass under HYBRID: x_2)
//   T0..T2 = a copy of the child's S tuple (lean_poly_S), so that the top-down pass may overwrite the child's own slot
//   DD = HYBRID's per-edge record (error, x_1)
// The S tuple of such a node is summed over the records in order after the level's binary step (lean_poly_S; BME's
// 1 / #valid children, apples/BME.py:19-20), which also marks the entry: D = (first record, LEAN_POLY_SELF | children).
// Top-down, the binary steps skip marked entries and lean_poly_td forms every child's R from ALL its siblings in file order,
// then the parent term (apples/OLS.py:59-80; BME: 1 / (nonroot + #valid siblings), apples/BME.py:36-38) -- O(children^2)
// like the reference.  A polytomy's own lift(R) always travels through its tuple slot, never through the LDS hand-over: its
// descriptor in its parent's entry and its key in the level list carry LEAN_POLY_KID so that the parent knows.
#define LEAN_POLY_KID 0x40000000
#define LEAN_POLY_SELF 0x20000000
#define LEAN_DESC_MASK 0x1fffffff
template <bool PL> __device__ __forceinline__ int lean_desc_idx(int kd) { return PL ? (kd & LEAN_DESC_MASK) : kd; }
template <bool PL> __device__ __forceinline__ int lean_key_node(int k) { return PL ? (k & ~LEAN_POLY_KID) : k; }
__device__ __forceinline__ bool lean_is_poly_entry(const int2 d) { return d.y > 0 && (d.y & LEAN_POLY_SELF) != 0 && (d.y & LEAN_POLY_KID) == 0; }

__device__ __forceinline__ LeanTeam lean_team(void *base, int64_t team, int64_t cap1, int64_t leaf1) {
    char *p = reinterpret_cast<char *>(base) + team * (cap1 * LEAN_BYTES_PER_NODE + leaf1 * LEAN_BYTES_PER_LEAF);
    LeanTeam t;
    t.T0 = reinterpret_cast<double2 *>(p); p += cap1 * 16;
    t.T1 = reinterpret_cast<double2 *>(p); p += cap1 * 16;
    t.T2 = reinterpret_cast<double2 *>(p); p += cap1 * 16;
    t.E = reinterpret_cast<double2 *>(p); p += cap1 * 16;
    t.DD = reinterpret_cast<double2 *>(p); p += cap1 * 16;
    t.D = reinterpret_cast<int2 *>(p); p += cap1 * 8;
    t.N = reinterpret_cast<int2 *>(p); p += cap1 * 8;
    t.K = reinterpret_cast<int32_t *>(p); p += cap1 * 4;
    t.LE = reinterpret_cast<double *>(p); p += leaf1 * 8;
    t.LP = reinterpret_cast<int32_t *>(p);
    t.BP = nullptr;
    return t;
}

// A query's entries inside the batch's pool (wavefront-sized teams: the bottom-up and the top-down pass are two kernels,
// so a query's arrays outlive the team that built them): `n` entries per field array, the query's at [off, off + its cap);
// the two per-leaf arrays are scratch of the bottom-up team
__device__ __forceinline__ LeanTeam lean_pool_view(void *pool, int64_t n, int64_t off, void *leaf, int64_t team, int64_t leaf1) {
    char *p = reinterpret_cast<char *>(pool);
    LeanTeam t;
    t.T0 = reinterpret_cast<double2 *>(p) + off; p += n * 16;
    t.T1 = reinterpret_cast<double2 *>(p) + off; p += n * 16;
    t.T2 = reinterpret_cast<double2 *>(p) + off; p += n * 16;
    t.E = reinterpret_cast<double2 *>(p) + off; p += n * 16;
    t.DD = reinterpret_cast<double2 *>(p) + off; p += n * 16;
    t.D = reinterpret_cast<int2 *>(p) + off; p += n * 8;
    t.N = reinterpret_cast<int2 *>(p) + off; p += n * 8;
    t.K = reinterpret_cast<int32_t *>(p) + off;
    char *l = reinterpret_cast<char *>(leaf) + team * leaf1 * LEAN_BYTES_PER_LEAF;
    t.LE = reinterpret_cast<double *>(l);
    t.LP = reinterpret_cast<int32_t *>(l + leaf1 * 8);
    t.BP = nullptr;
    return t;
}

__device__ __forceinline__ double shfl_down_f64(double v, int delta) {
    return __hiloint2double(__shfl_down(__double2hiint(v), delta, WAVE), __shfl_down(__double2loint(v), delta, WAVE));
}

__device__ __forceinline__ double pe_len(const int4 &r) { return __hiloint2double(r.w, r.z); }

// What a merge step knows about its runs of siblings (PL), as lane masks from the mask of the runs' first keys alone (scalar
// bit operations: nothing per lane): `extra` = the third and later keys of their runs, `xfirst` = exactly the third keys,
// `third` = the first keys of runs that have a third key inside this step.  A run that began in the step before: `ccnt` = how
// many of its keys that step saw (capped at 3); its keys here are the non-first lanes from lane 0 on.
struct LeanRunMasks {
    unsigned long long extra, xfirst, third;
};
__device__ __forceinline__ LeanRunMasks lean_run_masks(unsigned long long fm, int tot, int ccnt) {
    const unsigned long long act = tot >= 64 ? ~0ull : ((1ull << tot) - 1ull);
    const unsigned long long nf = ~fm & act;  // keys that continue a run
    LeanRunMasks r;
    // a key behind another continuing key is at least its run's third; lane 0 continues the carried run: third or later from two carried keys on
    r.extra = (nf & (nf << 1)) | ((nf & 1ull) && ccnt >= 2 ? 1ull : 0ull);
    // exactly the third: two lanes behind a first key -- or, in the carried run, where carried keys + lanes before make two
    r.xfirst = (r.extra & (fm << 2)) | ((nf & 1ull) && ccnt == 2 ? 1ull : 0ull) | ((nf & 3ull) == 3ull && ccnt == 1 ? 2ull : 0ull);
    r.third = fm & (nf >> 1) & (nf >> 2);
    return r;
}
struct LeanRun {
    bool extra, xfirst;
};

// The two polytomy passes are real calls, and their call sites hand them COPIES of what they take by reference (the team's
// pointers, the running best) and park the lane state that must survive in LDS: inlined, their registers (a node's children
// in flight together) came on top of the level loops' own and the hot paths of a tree without a single polytomy spilled --
// forced onto config 3's binary tree the PL kernels took 20.9 ms where the plain ones take 15.3 (profiles/r06_poly_exp.txt).
#ifndef LEAN_POLY_INLINE
#define LEAN_POLY_INLINE __noinline__
#endif
template <int M, bool PL>
__device__ __forceinline__ void kid_tuple(int kd, double dist, const LeanTeam &t, const double2 (*stage)[WAVE], bool staged,
                                          int stage_base, double *S);

// Polytomies whose whole run of sibling keys lies inside ONE merge step (nearly all of them: a run is cut only where a level of
// more than 64 keys happens to break inside it) are finished right there (PL).  The m lanes of the run hold the node's children:
// each loads its child's S tuple (final: the level below is complete), writes the child's record -- position -1 - j, which the
// later pass over the records takes for "done" --, lifts the tuple; the node's tuple is their sum in file order, handed from lane
// to lane (m - 1 rounds of shuffles; BME: every share times 1 / m, apples/BME.py:19-20), and the run's last lane stores it in the
// node's tuple slot, where the next level step picks it up instead of forming a tuple from the entry's first two children.  The
// run's first lane marks the entry (D = (first record, LEAN_POLY_SELF | m)).  fm: the step's first keys, third: the first keys of
// runs with a third key in the step (lean_run_masks), ebase: the entry of the step's first run head.
// Returns the records written; pm = the lanes served; head = this lane begins such a run (x0 / m: its first record, its children).
#ifndef LEAN_POLY_FAST_INLINE
#define LEAN_POLY_FAST_INLINE __forceinline__
#endif
template <int M>
__device__ LEAN_POLY_FAST_INLINE int lean_poly_fast(const LeanTeam &t, int xtop, int xc, unsigned long long fm, unsigned long long third, int tot,
                                              bool last_step, int ebase, int desc, int key, double e, double dist,
                                              const double2 (*stage)[WAVE], bool staged, int stage_base, int lane,
                                              unsigned long long extra, unsigned long long xfirst,
                                              unsigned long long &pm, bool &head, int &x0, int &m) {
    constexpr bool BME = (M == APPLES_BME);
    const unsigned long long below = (1ull << lane) - 1ull;
    const unsigned long long hb = fm & (below | (1ull << lane));  // run heads at or below this lane
    bool in = false;
    int hl = 0, end = 0;
    if (lane < tot && hb != 0ull) {
        hl = 63 - __clzll((long long)hb);
        const unsigned long long above = hl == 63 ? 0ull : (fm & ~((2ull << hl) - 1ull));
        end = above != 0ull ? __ffsll((long long)above) - 1 : tot;
        // (a run that reaches the end of a step that is not the level's last may go on in the next one: the records' pass serves it)
        in = ((third >> hl) & 1ull) != 0ull && (end < tot || last_step);
    }
#ifdef LEAN_EXP_NO_FAST
    in = false;  // (experiment: every polytomy through the records' pass)
#endif
    pm = __ballot(in);
    head = false; x0 = 0; m = 0;
    if (pm == 0ull) return 0;
    const int pos = lane - hl;
    m = end - hl;
    // (records go out in lane order, these and the ones of lean_poly_records together: a run that a step's end cut keeps its
    // records in one piece -- its lanes of the next step come first there)
    const unsigned long long s3 = xfirst & ~pm, s1 = extra & ~xfirst & ~pm;
    const int x = xc + __popcll(pm & below) + 3 * __popcll(s3 & below) + __popcll(s1 & below);
    x0 = x - pos;
    head = in && pos == 0;
    const int en = ebase + __popcll(hb) - 1;
    double S[6], u[6], acc[6];
    kid_tuple<M, true>(in ? desc : 0, dist, t, stage, staged, stage_base, S);
    if (in) {
        const int s = xtop - x;
        t.D[s] = make_int2(desc, -1 - pos);
        t.N[s] = make_int2(key, en);
        t.E[s] = make_double2(e, dist);
        t.T0[s] = make_double2(S[0], S[1]); t.T1[s] = make_double2(S[2], S[3]); t.T2[s] = make_double2(S[4], S[5]);
    }
    const double coef = BME ? 1.0 / (double)(m > 0 ? m : 1) : 1.0;
    lift<M>(S, e, u);
#pragma unroll
    for (int c = 0; c < 6; ++c) { u[c] = BME ? coef * u[c] : u[c]; acc[c] = 0.0 + u[c]; }  // (the reference starts every sum at 0)
    for (int k = 1; __ballot(in && pos >= k) != 0ull; ++k) {  // lane of position k: the sum so far (from the lane before) + its own share
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            const double pv = __hiloint2double(__shfl_up(__double2hiint(acc[c]), 1, WAVE), __shfl_up(__double2loint(acc[c]), 1, WAVE));
            if (in && pos == k) acc[c] = pv + u[c];
        }
    }
    if (in && pos == m - 1) {
        t.T0[en] = make_double2(acc[0], acc[1]); t.T1[en] = make_double2(acc[2], acc[3]); t.T2[en] = make_double2(acc[4], acc[5]);
    }
    return __popcll(pm);
}

// the child records of this step's third-and-later siblings (PL): `en` = the entry of the lane's run, `xc` = records so far;
// returns how many records the step added (three for a run's third key: the first two children's records are filled in by lean_poly_S)
__device__ __forceinline__ int lean_poly_records(const LeanTeam &t, int xtop, int xc, const LeanRun &run, int en, int par, int desc,
                                                 int key, double e, double dist, int lane, unsigned long long pm = 0ull) {
    const unsigned long long below = (1ull << lane) - 1ull;
    const unsigned long long m3 = __ballot(run.xfirst), m1 = __ballot(run.extra && !run.xfirst);
    if (run.extra) {
        const int x0 = xc + 3 * __popcll(m3 & below) + __popcll(m1 & below) + __popcll(pm & below);  // (pm: lean_poly_fast's lanes, same order)
        const int x = run.xfirst ? x0 + 2 : x0;
        t.D[xtop - x] = make_int2(desc, run.xfirst ? 2 : 3);
        t.N[xtop - x] = make_int2(key, en);
        t.E[xtop - x] = make_double2(e, dist);
        if (run.xfirst) {
            t.D[xtop - x0] = make_int2(0, 0);
            t.N[xtop - x0] = make_int2(-1, en);
            t.D[xtop - x0 - 1] = make_int2(0, 1);
            t.N[xtop - x0 - 1] = make_int2(-1, en);
            t.K[en] = par | LEAN_POLY_KID;
        }
    }
    return 3 * __popcll(m3) + __popcll(m1);
}

// One level of the lists, general form (any sizes): merge by node id the parents of this level's internal nodes
// (K[base .. base + nA), sorted) and of its observed leaves (o_node[lo .. lo + nB), sorted): the next level's list, sorted
// (sweep.hip:merge_parents -- the same merge-path step of 64 keys through two LDS windows).  The entry of a parent names
// its valid children and carries their node ids, edge lengths and (observed leaves) distances.  A binary tree's runs
// have at most two keys; a run cut by the end of a step is completed by the next step's first key, which patches the
// entry -- so how far a step advances in the two lists is known right after its search, and the next step's windows are
// requested before this step's gather is waited for: one memory round trip per 64 keys, not two.
// PL: runs of any length; the third and later keys of a run become child records (above), `xc` counts them.
// Returns the number of entries written from next_base on.
// (`xs`: the records left to lean_poly_S, i.e. of runs that a step's end cut)
template <int M, bool PL>
__device__ __forceinline__ int lean_merge(const LeanTeam &t, int base, int nA, const int32_t *__restrict__ o_node,
                                          const double *__restrict__ o_dist, int lo, int nB, int next_base,
                                          const int4 *__restrict__ pe, int *mk_a, int *mk_b, int lane, int xtop, int &xc, int &xs) {
    int out = 0, ia = 0, ib = 0, carry = -2, ccnt = 0;
    const unsigned long long below = (1ull << lane) - 1ull;
    int wk_a = lane < nA ? t.K[base + lane] : 0x7fffffff, wk_b = lane < nB ? o_node[lo + lane] : 0x7fffffff;
    while (ia < nA || ib < nB) {
        const int wa = min(nA - ia, WAVE), wb = min(nB - ib, WAVE);
        mk_a[lane] = wk_a;
        mk_b[lane] = wk_b;
        __builtin_amdgcn_wave_barrier();
        const int tot = min(wa + wb, WAVE);
        const bool active = lane < tot;
        int i_lo = max(0, lane - wb), i_hi = min(lane, wa);  // i = keys of the first window among the lane smallest
        while (i_lo < i_hi) {
            const int i = (i_lo + i_hi) >> 1;
            if (lean_key_node<PL>(mk_a[i]) < mk_b[lane - 1 - i]) i_lo = i + 1; else i_hi = i;
        }
        const int i = i_lo, j = lane - i_lo;
        const int kaf = i < wa ? mk_a[i] : 0x7fffffff;  // (PL: a polytomy's key carries LEAN_POLY_KID)
        const int ka = i < wa ? lean_key_node<PL>(kaf) : 0x7fffffff, kb = j < wb ? mk_b[j] : 0x7fffffff;
        const bool from_a = ka < kb;
        const int key = from_a ? ka : kb;
        const int desc = from_a ? ((base + ia + i + 1) | (PL ? (kaf & LEAN_POLY_KID) : 0)) : -(lo + ib + j) - 2;
        const int ca = __popcll(__ballot(active && from_a));
        const int nia = ia + ca, nib = ib + tot - ca;
        wk_a = nia + lane < nA ? t.K[base + nia + lane] : 0x7fffffff;  // the next step's windows
        wk_b = nib + lane < nB ? o_node[lo + nib + lane] : 0x7fffffff;
        int par = -3;
        double e = 0, dist = 0;
        if (active) {
            if (from_a) {
                const int4 r = pe[key];
                par = r.x;
                e = pe_len(r);
            } else {
                par = t.LP[lo + ib + j];
                e = t.LE[lo + ib + j];
                dist = o_dist[lo + ib + j];
            }
        }
        const int prev = __shfl_up(par, 1, WAVE);
        const bool first = active && par != (lane == 0 ? carry : prev);
        const int next_desc = __shfl_down(desc, 1, WAVE), next_key = __shfl_down(key, 1, WAVE);
        const int next_first = __shfl_down(first ? 1 : 0, 1, WAVE);
        const double next_e = shfl_down_f64(e, 1), next_dist = shfl_down_f64(dist, 1);
        const bool two = lane + 1 < tot && !next_first;
        const unsigned long long fm = __ballot(first);
        LeanRun run = {false, false};
        bool third = false;  // (PL) the run this lane begins has a third key inside this step
        LeanRunMasks rm = {0, 0, 0};
        bool fhead = false;  // (PL) ... and lies inside the step altogether: finished here (lean_poly_fast)
        int fx0 = 0, fm_ = 0, nfast = 0;
        unsigned long long pmf = 0ull;
        if (PL) {
            rm = lean_run_masks(fm, tot, ccnt);
            third = (rm.third >> lane) & 1ull;
            if (rm.third != 0ull) {  // (never on a binary tree)
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");  // (the level below's tuples: this wavefront's own stores)
                {   // (copies: the call takes references, and what the hot path reads stays in registers)
                    const LeanTeam tc = t;
                    unsigned long long pm_ = 0ull;
                    bool h_ = false;
                    int a_ = 0, b_ = 0;
                    nfast = lean_poly_fast<M>(tc, xtop, xc, fm, rm.third, tot, nia >= nA && nib >= nB, next_base + out, desc, key, e, dist,
                                              nullptr, false, 0, lane, rm.extra, rm.xfirst, pm_, h_, a_, b_);
                    pmf = pm_; fhead = h_; fx0 = a_; fm_ = b_;
                }
                rm.extra &= ~pmf; rm.xfirst &= ~pmf;
            }
            run.extra = (rm.extra >> lane) & 1ull;
            run.xfirst = (rm.xfirst >> lane) & 1ull;
        }
        if (first) {
            const int at = next_base + out + __popcll(fm & below);
            t.K[at] = (PL && third) ? (par | LEAN_POLY_KID) : par;
            t.D[at] = (PL && fhead) ? make_int2(fx0, LEAN_POLY_SELF | fm_) : make_int2(desc, two ? next_desc : 0);
            t.N[at] = make_int2(key, two ? next_key : -1);
            t.E[at] = make_double2(e, two ? next_e : 0.0);
            t.DD[at] = make_double2(dist, two ? next_dist : 0.0);
        } else if (lane == 0 && active && !(PL && run.extra)) {  // the second child of the previous step's last entry
            const int at = next_base + out - 1;
            reinterpret_cast<int *>(t.D + at)[1] = desc;
            reinterpret_cast<int *>(t.N + at)[1] = key;
            reinterpret_cast<double *>(t.E + at)[1] = e;
            reinterpret_cast<double *>(t.DD + at)[1] = dist;
        }
        if (PL) {
            if (rm.extra != 0ull) {  // (never on a binary tree)
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");  // (the mark on K follows the entry's own store)
                const int en = next_base + out + __popcll(fm & (below | (1ull << lane))) - 1;
                const int added = lean_poly_records(t, xtop, xc, run, en, par, desc, key, e, dist, lane, pmf);
                xc += added; xs += added;
            }
            xc += nfast;
            // how many keys of the last run this step saw (capped at 3), for the lanes 0..2 of the next step
            const int lastfirst = fm ? 63 - __clzll(fm) : -1;
            ccnt = lastfirst >= 0 ? min(tot - lastfirst, 3) : min(ccnt + tot, 3);
        }
        carry = __shfl(par, tot - 1, WAVE);
        ia = nia;
        ib = nib;
        out += __popcll(fm);
        __builtin_amdgcn_wave_barrier();
    }
    return out;
}

// a child's S tuple: an internal child's from the arrays (or from the LDS stage when its level is there), a leaf's
// rebuilt from its distance.  The loads from the arrays are unconditional (a leaf child or no child reads entry 0 and
// drops it): no branch stands between the loads of a node's two children and of its own tuple, so they all leave together
// and a step waits for one round trip, not for one per child.
template <int M, bool PL = false>
__device__ __forceinline__ void kid_tuple(int kd, double dist, const LeanTeam &t, const double2 (*stage)[WAVE], bool staged,
                                          int stage_base, double *S) {
    double2 a, b, c;
    if (staged) {  // (wave-uniform)
        const int p = kd > 0 ? lean_desc_idx<PL>(kd) - 1 - stage_base : 0;
        a = stage[0][p]; b = stage[1][p]; c = stage[2][p];
    } else {
        const int ki = kd > 0 ? lean_desc_idx<PL>(kd) - 1 : 0;
        a = t.T0[ki]; b = t.T1[ki]; c = t.T2[ki];
    }
    S[0] = a.x; S[1] = a.y; S[2] = b.x; S[3] = b.y; S[4] = c.x; S[5] = c.y;
    if (kd <= 0) {
        if (t.BP && lean_is_block(dist)) {  // the root of a clade block: its tuple waits in the pool
            const double *bp = t.BP + lean_block_at(dist);
#pragma unroll
            for (int x = 0; x < 6; ++x) S[x] = bp[x * 64];
        } else {
            leaf_tuple<M>(dist, S);
        }
    }
}

// S tuple of a node from its entry (apples/OLS.py:25-44: children in file order)
template <int M, bool PL = false>
__device__ __forceinline__ void node_S(const int2 d, const double2 e, const double2 dd, const LeanTeam &t,
                                       const double2 (*stage)[WAVE], bool staged, int kid_base, double *r) {
    constexpr bool BME = (M == APPLES_BME);
    const double coef = BME ? 1.0 / (double)(d.y != 0 ? 2 : 1) : 1.0;  // apples/BME.py:20
    double S0[6], S1[6], u[6];
    kid_tuple<M, PL>(d.x, dd.x, t, stage, staged, kid_base, S0);
    kid_tuple<M, PL>(d.y, dd.y, t, stage, staged, kid_base, S1);  // (no second child: a dummy that is not used)
    lift<M>(S0, e.x, u);
#pragma unroll
    for (int x = 0; x < 6; ++x) r[x] = 0;
#pragma unroll
    for (int x = 0; x < 6; ++x) r[x] += BME ? coef * u[x] : u[x];
    if (d.y != 0) {
        lift<M>(S1, e.y, u);
#pragma unroll
        for (int x = 0; x < 6; ++x) r[x] += BME ? coef * u[x] : u[x];
    }
}

// S tuples of the entries [lo, hi), `stride` lanes apart (a level of more than one chunk): the entries of the next chunk
// are requested before this chunk's children are waited for, so a chunk costs one memory round trip, not two
template <int M, bool PL = false>
__device__ __forceinline__ void lean_S_chunks(const LeanTeam &t, int lo, int hi, int first, int stride,
                                              const double2 (*stage)[WAVE], bool staged, int kid_base) {
    int idx = lo + first;
    int2 d = make_int2(0, 0);
    double2 e = make_double2(0, 0), dd = make_double2(0, 0);
    if (idx < hi) { d = t.D[idx]; e = t.E[idx]; dd = t.DD[idx]; }
    while (idx < hi) {
        const int nidx = idx + stride;
        int2 d2 = make_int2(0, 0);
        double2 e2 = make_double2(0, 0), dd2 = make_double2(0, 0);
        if (nidx < hi) { d2 = t.D[nidx]; e2 = t.E[nidx]; dd2 = t.DD[nidx]; }
        if (!(PL && lean_is_poly_entry(d))) {  // (a marked polytomy: lean_poly_fast left its tuple in the slot)
            double r[6];
            node_S<M, PL>(d, e, dd, t, stage, staged, kid_base, r);
            t.T0[idx] = make_double2(r[0], r[1]);
            t.T1[idx] = make_double2(r[2], r[3]);
            t.T2[idx] = make_double2(r[4], r[5]);
        }
        idx = nidx; d = d2; e = e2; dd = dd2;
    }
}

// running arg-min of a lane over the edges it solved (apples/Algorithm.py:74-91: smallest key, ties to the smaller edge_index)
struct LeanBest {
    double key, x1, x2, err, e;
    int v, x1_int;
};

// HYBRID's per-edge records: a pendant length that is Python's int 0 (apples/util.py:36-50, Sol::x1_int) is stored as this NaN (no
// arithmetic produces its payload)
#define LEAN_BOXED_INT0 0x7ff8ab5e00000001LL

__device__ __forceinline__ void lean_best_init(LeanBest &b) {
    b.key = INF_D; b.v = 0x7fffffff; b.x1 = b.x2 = b.err = b.e = 0; b.x1_int = 0;
}

// Top-down step for ONE valid child of an internal node (parent-centric): the node forms the child's R (all_R_values,
// OLS.py:57-80: the valid sibling, then the node's own lifted R), solves it (placement_per_edge, util.py:6-54) and
// evaluates its residual (error_per_edge); an internal child receives lift(R) over its own edge -- what it will add for
// each of its own children -- in its tuple slot, or through LDS (`hand`) when its level has at most 64 nodes.
// Sk / ek / kd / kn: the child's S tuple, edge length, descriptor, node id; Ss / es: the sibling's (nk > 1 only).
// HY (the HYBRID criterion, apples/Algorithm.py:76-82): nothing is compared here; the edge's error, x_1 and x_2 go into half `rz` of
// the tuple slots of its parent's entry `ridx` (dead by now: the entry's S went into its own parent's step, its lifted R was read
// when this step began), for lean_hybrid_pick.
template <int M, bool HY = false, bool PL = false>
__device__ __forceinline__ void lean_td_kid(const LeanTeam &t, const double *Sk, const double *Ss, const double *plift, double ek,
                                            double es, int kd, int kn, int nk, bool is_lca, double coef, int negative,
                                            int criterion, const double *lds_pow, double2 (*hand)[WAVE], int hand_base,
                                            LeanBest &best, double ddk = 0.0, int ridx = 0, int rz = 0) {
    constexpr bool BME = (M == APPLES_BME);
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
    const Sol r = solve_edge<M>(Sk, acc, ek, negative, lds_pow);
    if (kd > 0) {
        double u[6];
        lift<M>(acc, ek, u);
        const int ki = lean_desc_idx<PL>(kd) - 1;
        if (hand && !(PL && (kd & LEAN_POLY_KID))) {  // (a polytomy takes its lift(R) from its tuple slot: lean_poly_td)
            const int p = ki - hand_base;
            hand[0][p] = make_double2(u[0], u[1]);
            hand[1][p] = make_double2(u[2], u[3]);
            hand[2][p] = make_double2(u[4], u[5]);
        } else {
            t.T0[ki] = make_double2(u[0], u[1]);
            t.T1[ki] = make_double2(u[2], u[3]);
            t.T2[ki] = make_double2(u[4], u[5]);
        }
    } else if (t.BP && lean_is_block(ddk)) {  // the root of a clade block: lift(R) over its edge replaces its S in the pool (k_blocks_down)
        double u[6];
        lift<M>(acc, ek, u);
        double *bp = t.BP + lean_block_at(ddk);
#pragma unroll
        for (int x = 0; x < 6; ++x) bp[x * 64] = u[x];
    }
    if (HY) {
        reinterpret_cast<double *>(t.T0 + ridx)[rz] = r.err;
        reinterpret_cast<double *>(t.T1 + ridx)[rz] = r.x1_int ? __longlong_as_double(LEAN_BOXED_INT0) : r.x1;
        reinterpret_cast<double *>(t.T2 + ridx)[rz] = r.x2;
        return;
    }
    const double key = (criterion == APPLES_ME) ? r.x1 : r.err;
    if (key < best.key || (key == best.key && kn < best.v)) {
        best.key = key; best.v = kn; best.x1 = r.x1; best.x2 = r.x2; best.err = r.err; best.x1_int = r.x1_int; best.e = ek;
    }
}

// the node's own lifted R: from the LDS hand-over of its parent's step (position p) or from its tuple slot
__device__ __forceinline__ void lean_own_plift(const LeanTeam &t, int idx, const double2 (*hand)[WAVE], int p, double *plift) {
    double2 p0, p1, p2;
    if (hand) { p0 = hand[0][p]; p1 = hand[1][p]; p2 = hand[2][p]; }
    else { p0 = t.T0[idx]; p1 = t.T1[idx]; p2 = t.T2[idx]; }
    plift[0] = p0.x; plift[1] = p0.y; plift[2] = p1.x; plift[3] = p1.y; plift[4] = p2.x; plift[5] = p2.y;
}

template <int M, bool HY>
__device__ __forceinline__ void lean_poly_td(const LeanTeam &t, int xtop, unsigned long long pmask, int my_en, int my_x, int my_m, int VI,
                                             int negative, int criterion, const double *lds_pow, double2 (*hand)[WAVE], int hand_base,
                                             double (*pu)[WAVE], LeanBest &best, int lane);

// Top-down step of one internal node, both children in turn (the two swap roles in between): a rolled loop keeps one
// 2x2 solve's worth of temporaries live, which is what decides how many wavefronts a SIMD holds.
template <int M, bool HY = false, bool PL = false>
__device__ __forceinline__ void lean_td_node(const LeanTeam &t, int idx, const int2 d, const int2 nd, const double2 e, const double2 dd,
                                             bool is_lca, int negative, int criterion,
                                             const double *lds_pow, const double2 (*hand_in)[WAVE], int in_pos,
                                             double2 (*hand_out)[WAVE], int out_base, LeanBest &best, int &p_x, int &p_m) {
    constexpr bool BME = (M == APPLES_BME);
    if (PL && lean_is_poly_entry(d)) { p_x = d.x; p_m = d.y & LEAN_DESC_MASK; return; }  // (lean_poly_td serves it: its first record, its children)
    const int nk = d.y != 0 ? 2 : 1;
    // apples/BME.py:36-37: 1 / (nonroot + #valid siblings)
    const double coef = BME ? 1.0 / (double)((is_lca ? 0 : 1) + nk - 1) : 1.0;
    double plift[6];
    lean_own_plift(t, idx, hand_in, in_pos, plift);  // (the LCA has none: what it reads there is not used)
    double Sk[6], Ss[6];  // the child in hand and its sibling
    kid_tuple<M, PL>(d.x, dd.x, t, nullptr, false, 0, Sk);
    kid_tuple<M, PL>(d.y, dd.y, t, nullptr, false, 0, Ss);
    double ek = e.x, es = e.y, ddk = dd.x, dds = dd.y;
    int kd = d.x, ks = d.y, kn = nd.x, ksn = nd.y;
#pragma unroll 1
    for (int z = 0; z < nk; ++z) {
        lean_td_kid<M, HY, PL>(t, Sk, Ss, plift, ek, es, kd, kn, nk, is_lca, coef, negative, criterion, lds_pow, hand_out, out_base, best, ddk, idx, z);
#pragma unroll
        for (int x = 0; x < 6; ++x) { const double w = Sk[x]; Sk[x] = Ss[x]; Ss[x] = w; }
        { const double w = ek; ek = es; es = w; }
        { const double w = ddk; ddk = dds; dds = w; }
        { const int w = kd; kd = ks; ks = w; }
        { const int w = kn; kn = ksn; ksn = w; }
    }
}

// Top-down steps of the entries [lo, hi), `stride` lanes apart.  AHEAD: the next chunk's entries requested ahead (as
// lean_S_chunks) -- for the workgroup-sized teams, which wait on memory; the wavefront-sized teams' top-down kernel is
// bound by instruction issue at three wavefronts per SIMD, and the registers of the look-ahead (spilled there) cost it
// more than the round trip (measured: 6.9 against 5.0 ms per C3 pass)
template <int M, bool AHEAD, bool HY = false, bool PL = false>
__device__ __forceinline__ void lean_td_chunks(const LeanTeam &t, int lo, int hi, int first, int stride, int VI, int negative,
                                               int criterion, const double *lds_pow, const double2 (*hand_in)[WAVE],
                                               double2 (*hand_out)[WAVE], int out_base, LeanBest &best, int xtop = 0,
                                               double (*pu)[WAVE] = nullptr) {
    int idx = lo + first;
    int2 d = make_int2(0, 0), nd = make_int2(0, 0);
    double2 e = make_double2(0, 0), dd = make_double2(0, 0);
    if (!AHEAD) {
        // (the loop runs to the wavefront's common end: a lane beyond the level idles through the polytomies' call, which is the wavefront's)
        const int rounds = (hi - lo + stride - 1) / stride;
        for (int k = 0; k < rounds; ++k, idx += stride) {
            int p_x = 0, p_m = 0;
            if (idx < hi)
                lean_td_node<M, HY, PL>(t, idx, t.D[idx], t.N[idx], t.E[idx], t.DD[idx], idx == VI, negative, criterion, lds_pow, hand_in, idx - lo,
                                        hand_out, out_base, best, p_x, p_m);
            __builtin_amdgcn_wave_barrier();
            if (PL) {
                const unsigned long long pmask = __ballot(p_m > 0);
                if (pmask != 0ull) lean_poly_td<M, HY>(t, xtop, pmask, idx, p_x, p_m, VI, negative, criterion, lds_pow, hand_out, out_base, pu, best, first & (WAVE - 1));
            }
        }
        return;
    }
    if (idx < hi) { d = t.D[idx]; nd = t.N[idx]; e = t.E[idx]; dd = t.DD[idx]; }
    // (PL: the wavefront stays together to its last lane's last entry -- a polytomy's step is the whole wavefront's)
    while (PL ? __ballot(idx < hi) != 0ull : idx < hi) {
        const bool on = idx < hi;
        const int nidx = idx + stride;
        int2 d2 = make_int2(0, 0), nd2 = make_int2(0, 0);
        double2 e2 = make_double2(0, 0), dd2 = make_double2(0, 0);
        if (nidx < hi) { d2 = t.D[nidx]; nd2 = t.N[nidx]; e2 = t.E[nidx]; dd2 = t.DD[nidx]; }
        int p_x = 0, p_m = 0;
        if (!PL || on)
            lean_td_node<M, HY, PL>(t, idx, d, nd, e, dd, idx == VI, negative, criterion, lds_pow, hand_in, idx - lo, hand_out, out_base, best, p_x, p_m);
        __builtin_amdgcn_wave_barrier();
        if (PL) {  // (a workgroup-sized team: every wavefront serves its own lanes' polytomies, no barrier inside)
            const unsigned long long pmask = __ballot(p_m > 0);
            if (pmask != 0ull) lean_poly_td<M, HY>(t, xtop, pmask, idx, p_x, p_m, VI, negative, criterion, lds_pow, hand_out, out_base, pu, best, first & (WAVE - 1));
        }
        idx = nidx; d = d2; nd = nd2; e = e2; dd = dd2;
    }
}

// The same for a level of at most 32 nodes, one lane per (node, child): lanes 0-31 take the first valid children of
// nodes g0 .. g0 + 31, lanes 32-63 the second: the 2x2 solve and the residual run once per level step, not twice.
template <int M, bool HY = false, bool PL = false>
__device__ __forceinline__ void lean_td_pairs(const LeanTeam &t, int g0, int ng, int VI, int lane, int negative, int criterion,
                                              const double *lds_pow, const double2 (*hand_in)[WAVE], double2 (*hand_out)[WAVE],
                                              int out_base, LeanBest &best, int xtop = 0, double (*pu)[WAVE] = nullptr) {
    constexpr bool BME = (M == APPLES_BME);
    const int i = lane & 31, z = lane >> 5;
    const int idx = g0 + i;
    const bool act = i < ng;
    const bool is_lca = idx == VI;
    int2 d = make_int2(0, 0), nd = make_int2(0, 0);
    double2 e = make_double2(0, 0), dd = make_double2(0, 0);
    if (act) { d = t.D[idx]; nd = t.N[idx]; e = t.E[idx]; dd = t.DD[idx]; }
    const int nk = d.y != 0 ? 2 : 1;
    const double coef = BME ? 1.0 / (double)((is_lca ? 0 : 1) + nk - 1) : 1.0;
    double plift[6], Sk[6], Ss[6];
    const bool mine = act && z < nk && !(PL && lean_is_poly_entry(d));  // (a polytomy: lean_poly_td)
    if (mine) {
        lean_own_plift(t, idx, hand_in, i, plift);
        kid_tuple<M, PL>(z ? d.y : d.x, z ? dd.y : dd.x, t, nullptr, false, 0, Sk);
        kid_tuple<M, PL>(z ? d.x : d.y, z ? dd.x : dd.y, t, nullptr, false, 0, Ss);
    }
    __builtin_amdgcn_wave_barrier();  // (every lane's reads of the hand-over precede the stores of this step)
    if (mine)
        lean_td_kid<M, HY, PL>(t, Sk, Ss, plift, z ? e.y : e.x, z ? e.x : e.y, z ? d.y : d.x, z ? nd.y : nd.x, nk, is_lca, coef, negative,
                               criterion, lds_pow, hand_out, out_base, best, z ? dd.y : dd.x, idx, z);
    if (PL) {  // this level's polytomies (lanes 0-31 own the entries)
        const bool pol = act && z == 0 && lean_is_poly_entry(d);
        const unsigned long long pmask = __ballot(pol);
        if (pmask != 0ull) lean_poly_td<M, HY>(t, xtop, pmask, idx, d.x, d.y & LEAN_DESC_MASK, VI, negative, criterion, lds_pow, hand_out, out_base, pu, best, lane);
    }
}

// Polytomies, bottom-up (PL): the S tuples of the nodes whose child records lie in [xlo, xhi) -- the records a level's merge
// added -- after the level's binary step (which formed something from their first two children alone).  A lane per node: the
// first two children's records are filled in from the entry, every child's S tuple is copied into its record, the node's
// tuple is their sum in file order (apples/OLS.py:36-44; BME: each share times 1 / #valid children, apples/BME.py:19-20) and
// replaces what the binary step left (in the LDS stage too when the level is staged); the entry is marked.  Children's
// tuples come from the arrays (every level's are stored there as well as staged).
#ifndef LEAN_EXP_NO_FIX
#define LEAN_EXP_NO_FIX 0
#endif
#define LEAN_POLY_REG 4  // children a polytomy's lane keeps in registers (all their loads in flight together); more: one by one
template <int M>
__device__ LEAN_POLY_INLINE void lean_poly_S(const LeanTeam &t, int xtop, int xlo, int xhi, int first, int stride,
                                         double2 (*stage)[WAVE], int stage_base) {
    constexpr bool BME = (M == APPLES_BME);
#pragma unroll 1
    for (int x = xlo + first; x < xhi; x += stride) {
        // round 1: is this a node's first record, its entry, how many children (the positions of the records behind it)
        const int p0 = t.D[xtop - x].y, en = t.N[xtop - x].y;
        int pj[LEAN_POLY_REG + 1];
#pragma unroll
        for (int j = 2; j <= LEAN_POLY_REG; ++j) pj[j] = t.D[xtop - min(x + j, xhi - 1)].y;
        if (p0 != 0) continue;  // (a node begins at its position-0 record)
        int m = 2;
#pragma unroll
        for (int j = 2; j <= LEAN_POLY_REG; ++j)
            if (m == j && x + j < xhi && pj[j] >= 2) m = j + 1;
        if (m > LEAN_POLY_REG)
            while (x + m < xhi && t.D[xtop - x - m].y >= 2) ++m;
        // round 2: the first two children from the entry, the others' records
        const int2 ed = t.D[en], ekn = t.N[en];
        const double2 ee = t.E[en], edd = t.DD[en];
        int kd[LEAN_POLY_REG];
        double2 ke[LEAN_POLY_REG];
#pragma unroll
        for (int j = 2; j < LEAN_POLY_REG; ++j) { const int s = xtop - x - min(j, m - 1); kd[j] = t.D[s].x; ke[j] = t.E[s]; }
        kd[0] = ed.x; kd[1] = ed.y; ke[0] = make_double2(ee.x, edd.x); ke[1] = make_double2(ee.y, edd.y);
        t.D[xtop - x] = make_int2(ed.x, 0); t.N[xtop - x] = make_int2(ekn.x, en); t.E[xtop - x] = ke[0];
        t.D[xtop - x - 1] = make_int2(ed.y, 1); t.N[xtop - x - 1] = make_int2(ekn.y, en); t.E[xtop - x - 1] = ke[1];
        const double coef = BME ? 1.0 / (double)m : 1.0;
        double r[6] = {0, 0, 0, 0, 0, 0};
        // round 3: the children's tuples, all requested together (kid_tuple's loads are unconditional)
        double S[LEAN_POLY_REG][6];
#pragma unroll
        for (int j = 0; j < LEAN_POLY_REG; ++j) kid_tuple<M, true>(j < m ? kd[j] : 0, ke[j < m ? j : 0].y, t, nullptr, false, 0, S[j]);
#pragma unroll
        for (int j = 0; j < LEAN_POLY_REG; ++j) {
            if (j < m) {
                const int s = xtop - x - j;
                t.T0[s] = make_double2(S[j][0], S[j][1]); t.T1[s] = make_double2(S[j][2], S[j][3]); t.T2[s] = make_double2(S[j][4], S[j][5]);
                double u[6];
                lift<M>(S[j], ke[j].x, u);
#pragma unroll
                for (int c = 0; c < 6; ++c) r[c] += BME ? coef * u[c] : u[c];
            }
        }
#pragma unroll 1
        for (int j = LEAN_POLY_REG; j < m; ++j) {  // (a polytomy of more than LEAN_POLY_REG children: the rest one by one)
            const int s = xtop - x - j;
            const int dj = t.D[s].x;
            const double2 ej = t.E[s];
            double Sj[6], u[6];
            kid_tuple<M, true>(dj, ej.y, t, nullptr, false, 0, Sj);
            t.T0[s] = make_double2(Sj[0], Sj[1]); t.T1[s] = make_double2(Sj[2], Sj[3]); t.T2[s] = make_double2(Sj[4], Sj[5]);
            lift<M>(Sj, ej.x, u);
#pragma unroll
            for (int c = 0; c < 6; ++c) r[c] += BME ? coef * u[c] : u[c];
        }
        t.T0[en] = make_double2(r[0], r[1]); t.T1[en] = make_double2(r[2], r[3]); t.T2[en] = make_double2(r[4], r[5]);
        if (stage) {
            const int p = en - stage_base;
            stage[0][p] = make_double2(r[0], r[1]); stage[1][p] = make_double2(r[2], r[3]); stage[2][p] = make_double2(r[4], r[5]);
        }
        t.D[en] = make_int2(x, LEAN_POLY_SELF | m);
    }
}

// Polytomies, top-down (PL): the nodes whose child records lie in [xlo, xhi), after the level's binary step skipped them.  A
// lane per node; for every child in file order its R = every other valid child's lifted S in file order, then the node's own
// lifted R (apples/OLS.py:59-80; BME: all of it times 1 / (nonroot + #valid siblings), apples/BME.py:36-38), the 2x2 solve and
// the residual as lean_td_kid; an internal child's lift(R) goes where lean_td_kid would put it (the children's S tuples are
// the copies in the records: the child's own slot is free to take it).  Up to LEAN_POLY_REG children: their records and
// their lifted tuples (coefficient applied: what every sibling's sum adds, the same bits each time) stay in registers.
template <int M, bool HY>
__device__ __forceinline__ void lean_poly_edge(const LeanTeam &t, int s, const double *Sk, const double *acc, double ek, double ddk, int kd,
                                               int kn, int negative, int criterion, const double *lds_pow, double2 (*hand)[WAVE],
                                               int hand_base, LeanBest &best) {
    const Sol r = solve_edge<M>(Sk, acc, ek, negative, lds_pow);
    if (kd > 0) {
        double u[6];
        lift<M>(acc, ek, u);
        const int ki = (kd & LEAN_DESC_MASK) - 1;
        if (hand && !(kd & LEAN_POLY_KID)) {
            const int p = ki - hand_base;
            hand[0][p] = make_double2(u[0], u[1]); hand[1][p] = make_double2(u[2], u[3]); hand[2][p] = make_double2(u[4], u[5]);
        } else {
            t.T0[ki] = make_double2(u[0], u[1]); t.T1[ki] = make_double2(u[2], u[3]); t.T2[ki] = make_double2(u[4], u[5]);
        }
    } else if (t.BP && lean_is_block(ddk)) {  // the root of a clade block (as lean_td_kid)
        double u[6];
        lift<M>(acc, ek, u);
        double *bp = t.BP + lean_block_at(ddk);
#pragma unroll
        for (int k = 0; k < 6; ++k) bp[k * 64] = u[k];
    }
    if (HY) {  // the edge's record for lean_hybrid_pick: (error, x_1) in DD, x_2 in place of the distance
        t.DD[s] = make_double2(r.err, r.x1_int ? __longlong_as_double(LEAN_BOXED_INT0) : r.x1);
        t.E[s] = make_double2(ek, r.x2);
    } else {
        const double key = (criterion == APPLES_ME) ? r.x1 : r.err;
        if (key < best.key || (key == best.key && kn < best.v)) {
            best.key = key; best.v = kn; best.x1 = r.x1; best.x2 = r.x2; best.err = r.err; best.x1_int = r.x1_int; best.e = ek;
        }
    }
}

// The polytomies of a top-down step, one after the other, each by the whole wavefront: lane j takes child j (its record: the copy
// of its S tuple, its edge, descriptor, node id), lifts the tuple (coefficient applied) and leaves it in LDS (`pu`, [6][64]); then
// its R = every OTHER child's share in file order + the node's own lifted R, the 2x2 solve and the residual (lean_poly_edge), the
// lane's running best as in the binary steps.  pmask: the lanes whose entry of this step is a marked polytomy (my_en / my_x /
// my_m: the entry, its first record, its children).  More than 64 children: in rounds of 64, the others' shares from the records
// each time.  Inline: its registers are those of a binary step (one child, one solve per lane).
template <int M, bool HY>
__device__ __forceinline__ void lean_poly_td(const LeanTeam &t, int xtop, unsigned long long pmask, int my_en, int my_x, int my_m, int VI,
                                             int negative, int criterion, const double *lds_pow, double2 (*hand)[WAVE], int hand_base,
                                             double (*pu)[WAVE], LeanBest &best, int lane) {
    constexpr bool BME = (M == APPLES_BME);
    while (pmask != 0ull) {  // (wave-uniform)
        const int src = __ffsll((long long)pmask) - 1;
        pmask &= pmask - 1ull;
        const int en = __shfl(my_en, src, WAVE), x = __shfl(my_x, src, WAVE), m = __shfl(my_m, src, WAVE);
        const bool is_lca = en == VI;
        const double coef = BME ? 1.0 / (double)((is_lca ? 0 : 1) + m - 1) : 1.0;
        double plift[6];
        lean_own_plift(t, en, nullptr, 0, plift);  // (every lane the same address; the LCA has none: not used)
#pragma unroll 1
        for (int c0 = 0; c0 < m; c0 += WAVE) {
            const int j = c0 + lane;
            const bool act = j < m;
            const int s = xtop - x - (act ? j : 0);
            const double2 a0 = t.T0[s], a1 = t.T1[s], a2 = t.T2[s];
            const double Sk[6] = {a0.x, a0.y, a1.x, a1.y, a2.x, a2.y};
            const double2 ed = t.E[s];
            const int kd = t.D[s].x, kn = t.N[s].x;
            double acc[6] = {0, 0, 0, 0, 0, 0};
            if (m <= WAVE) {
                double u[6];
                lift<M>(Sk, ed.x, u);
                __builtin_amdgcn_wave_barrier();  // (the shares of the polytomy before: read)
                if (act) {
#pragma unroll
                    for (int k = 0; k < 6; ++k) pu[k][lane] = BME ? coef * u[k] : u[k];
                }
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                for (int jj = 0; jj < m; ++jj) {  // (uniform bound; the lane skips its own child)
                    if (jj != j) {
#pragma unroll
                        for (int k = 0; k < 6; ++k) acc[k] += pu[k][jj];
                    }
                }
            } else {
#pragma unroll 1
                for (int jj = 0; jj < m; ++jj) {
                    const int s2 = xtop - x - jj;
                    const double2 b0 = t.T0[s2], b1 = t.T1[s2], b2 = t.T2[s2];
                    const double Sj[6] = {b0.x, b0.y, b1.x, b1.y, b2.x, b2.y};
                    double u[6];
                    lift<M>(Sj, t.E[s2].x, u);
                    if (jj != j) {
#pragma unroll
                        for (int k = 0; k < 6; ++k) acc[k] += BME ? coef * u[k] : u[k];
                    }
                }
            }
            if (!is_lca) {
#pragma unroll
                for (int k = 0; k < 6; ++k) acc[k] += BME ? coef * plift[k] : plift[k];
            }
            if (act) lean_poly_edge<M, HY>(t, s, Sk, acc, ed.x, ed.y, kd, kn, negative, criterion, lds_pow, hand, hand_base, best);
        }
    }
}

// the query's placement from the team's winner (apples/Algorithm.py:92-101); `mine` = this lane holds the winning edge
__device__ __forceinline__ void lean_write_placement(apples_placement *out, int64_t q, int V, int win, bool mine, bool lane0,
                                                     const LeanBest &best) {
    // (n_valid: the selection kernel left the nodes inside the query's clade blocks there -- 0 without blocks)
    if (win == 0x7fffffff) {
        if (lane0) {
            apples_placement pl = out[q];
            pl.n_valid += V;
            pl.edge = -1;
            pl.flags |= APPLES_F_DEGENERATE | APPLES_F_PENDANT_INT;
            out[q] = pl;
        }
    } else if (mine) {
        apples_placement pl = out[q];
        pl.n_valid += V;
        pl.edge = win;
        pl.error = best.err;
        pl.distal = best.e - best.x2;
        pl.pendant = best.x1;
        pl.flags = 0;
        if (best.x1_int) pl.flags |= APPLES_F_PENDANT_INT;
        if (best.x1 == 0 && best.err > 0 && (best.x2 == 0 || best.x2 == best.e)) pl.flags |= APPLES_F_MISPLACED;
        out[q] = pl;
    }
}


// HYBRID (apples/Algorithm.py:76-82): heapq.nsmallest(floor(log2(num_nodes))) of the valid edges by error -- stable, so ties go to
// the earlier edge in post-order = the smaller edge index -- then the first minimum of x_1 among those in that order.  The records
// are what lean_td_kid<HY> left in the entries [lo, hi) (T0: the two children's errors, T1: x_1, T2: x_2; D, N, E as the bottom-up
// pass wrote them).  One round per rank: every lane offers the smallest (error, edge) of its entries beyond the last one taken, the
// team's arg-min takes it, and the lane that owns it keeps it if its x_1 beats what the lane kept before (a later rank never
// replaces an equal x_1, and a first rank whose x_1 is a NaN keeps the place as it does under Python's min: key -inf here).  The
// team's arg-min over (x_1 kept, rank) then names the winner.  A NaN error is never taken (sweep.hip's HYBRID: the same rule).
// `argmin(key, id)`: the team's lexicographic arg-min, result in every lane.  Returns 0x7fffffff when there is no edge at all;
// `mine` = this lane holds the winner (its record in `best`).
// xn > 0 (polytomies): the edges below a polytomy have their records in the query's child records [0, xn) (lean_poly_td), the
// marked entries themselves hold none.
template <class ArgMin>
__device__ __forceinline__ int lean_hybrid_pick(const LeanTeam &t, int lo, int hi, int V, int first, int stride, ArgMin &&argmin,
                                                LeanBest &best, bool &mine, int xtop = 0, int xn = 0) {
    const int kk = 31 - __clz(V);
    double last_e = -INF_D;
    int last_v = -1;
    double keep_x1 = INF_D;
    int keep_rank = 0x7fffffff;
    lean_best_init(best);
    // a lane's KL smallest (error, edge) beyond the last one taken, sorted, and where they are (2 x entry + child); `more`: the
    // lane has others beyond those.  A lane offers the head of its list; when the list runs dry and there are more, it scans its
    // entries again from the team's last pick on (what lies before that has been taken: a lane's smaller keys were offered first).
    // floor(log2(num_nodes)) ranks over 64 or more lanes: the second scan is rare, so the ranking costs one pass over the entries.
    constexpr int KL = 4;
    double ce[KL];
    int cv[KL], cat[KL];
    int filled = 0, head = 0;
    bool more = true;
    auto refill = [&]() {
        filled = 0; head = 0;
        int seen = 0;
        auto offer = [&](double e, int v, int at) {
            const bool after = (e > last_e) || (e == last_e && v > last_v);
            if (!after) return;  // (a NaN error is never after anything)
            ++seen;
            int pos = filled < KL ? filled : KL;
#pragma unroll
            for (int k = KL - 1; k >= 0; --k)
                if (k < filled && (e < ce[k] || (e == ce[k] && v < cv[k]))) pos = k;
            if (pos >= KL) return;
#pragma unroll
            for (int k = KL - 1; k > 0; --k)
                if (k > pos && k <= filled) { ce[k] = ce[k - 1]; cv[k] = cv[k - 1]; cat[k] = cat[k - 1]; }
#pragma unroll
            for (int k = 0; k < KL; ++k)
                if (k == pos) { ce[k] = e; cv[k] = v; cat[k] = at; }
            if (filled < KL) ++filled;
        };
        for (int idx = lo + first; idx < hi; idx += stride) {
            const int2 d = t.D[idx], nd = t.N[idx];
            if (xn > 0 && lean_is_poly_entry(d)) continue;
            const double2 er = t.T0[idx];
            offer(er.x, nd.x, 2 * idx);
            if (d.y != 0) offer(er.y, nd.y, 2 * idx + 1);
        }
        for (int x = first; x < xn; x += stride) offer(t.DD[xtop - x].x, t.N[xtop - x].x, LEAN_POLY_KID | x);
        more = seen > filled;
    };
    for (int r = 0; r < kk; ++r) {
        if (head == filled && more) refill();
        double ke = INF_D;
        int kv = 0x7fffffff, at = -1;
#pragma unroll
        for (int k = 0; k < KL; ++k)
            if (k == head && k < filled) { ke = ce[k]; kv = cv[k]; at = cat[k]; }
        const int my_v = kv;
        argmin(ke, kv);
        if (kv == 0x7fffffff) break;
        last_e = ke; last_v = kv;
        if (my_v == kv && at >= 0) {  // (edge indices are distinct: one owner)
            ++head;
            const bool rec = (at & LEAN_POLY_KID) != 0;  // (a child record of a polytomy)
            const int idx = rec ? xtop - (at & ~LEAN_POLY_KID) : at >> 1, z = rec ? 1 : at & 1;
            double x1 = rec ? t.DD[idx].y : reinterpret_cast<const double *>(t.T1 + idx)[z];
            const bool boxed = __double_as_longlong(x1) == LEAN_BOXED_INT0;
            if (boxed) x1 = 0;
            const double kx = (r == 0 && !(x1 == x1)) ? -INF_D : x1;
            if (r == 0 || kx < keep_x1) {
                keep_x1 = kx; keep_rank = r;
                best.v = kv; best.x1 = x1; best.x1_int = boxed ? 1 : 0; best.err = ke;
                best.x2 = rec ? t.E[idx].y : reinterpret_cast<const double *>(t.T2 + idx)[z];
                best.e = reinterpret_cast<const double *>(t.E + idx)[rec ? 0 : z];
            }
        }
    }
    const int my_rank = keep_rank;
    double wk = keep_x1;
    int wr = keep_rank;
    argmin(wk, wr);
    mine = wr != 0x7fffffff && my_rank == wr;
    return wr == 0x7fffffff ? 0x7fffffff : 0;
}

// The size-class queues of a batch (written by the selection kernels): entry w -> query, largest first: the four parts
// of the largest class (lists 4..7, counts at cls_count[16..19]), then classes 1..3
struct LeanQueue {
    int c[7];
    int64_t n_work;
};
__device__ __forceinline__ LeanQueue lean_queue(const SweepArgs &a) {
    LeanQueue qu;
    qu.c[0] = a.cls_count[16]; qu.c[1] = a.cls_count[17]; qu.c[2] = a.cls_count[18]; qu.c[3] = a.cls_count[19];
    qu.c[4] = a.cls_count[1]; qu.c[5] = a.cls_count[2]; qu.c[6] = a.cls_count[3];
    qu.n_work = 0;
#pragma unroll
    for (int k = 0; k < 7; ++k) qu.n_work += qu.c[k];
    return qu;
}
__device__ __forceinline__ int64_t lean_queue_at(const SweepArgs &a, const LeanQueue &qu, int64_t w) {
#pragma unroll
    for (int k = 0; k < 7; ++k) {
        if (w < qu.c[k]) return a.cls_list[(k < 4 ? 4 + k : k - 3) * a.cls_stride + w];
        w -= qu.c[k];
    }
    return 0;
}

// entries a query may use in the pool: the internal nodes of its subtree plus the LCA.  Observed sets average 1.44
// internal nodes per observed leaf at C3, but a handful of leaves far apart in the tree have a path of internal nodes
// each (the -b nearest of a query with nothing inside the threshold): room for some 25 levels of those.  A query that
// needs more than this goes to the workgroup-sized teams.
__device__ __forceinline__ int lean_query_cap(int n) { return (3 * n + 128 + min(22 * n, 896) + 3) & ~3; }

// The top-down pass of ONE query of a wavefront-sized team (all_R_values, placement_per_edge, error_per_edge, the arg-min of
// apples/Algorithm.py:74-91) from what its bottom-up pass left: the pool offset, the number of level groups G and of internal
// nodes VI, the groups' offsets.  `stage`: the wavefront's LDS area for tuples on their way to a level of at most 64 nodes.
// PL (polytomies): xn = the query's child records (in its entry range from the top down), the records of group g's nodes at
// [xoff[g], xoff[g + 1]) with xoff = the second half of the query's group offsets.
template <int M, bool HY = false, bool PL = false>
__device__ __forceinline__ void lean_down_one(const SweepArgs &a, const double *lds_pow, double2 (*stage)[WAVE], int64_t q, int n,
                                              int64_t off, int G, int VI, int lane, int xn = 0, double (*pu)[WAVE] = nullptr) {
    const DevTree &T = a.tree;
    const int V = VI + n;  // Subtree.num_nodes
    LeanTeam t = lean_pool_view(a.lean, a.lean_cap1, off, nullptr, 0, 0);
    t.BP = a.blk_pool;
    const int32_t *grp_off = a.grp_off + q * (int64_t)a.grp_stride;
    const int xtop = lean_query_cap(n) - 1;
    LeanBest best;
    lean_best_init(best);
    bool hand_in = false;  // this level's lifted R tuples wait in LDS (handed over by the level above)
    for (int g = G; g >= 1; --g) {
        const int g0 = grp_off[g], g1 = grp_off[g + 1], k0 = grp_off[g - 1];
        const int ng = g1 - g0;
        const bool hand_out = g0 - k0 <= WAVE;  // the children's level has at most 64 nodes: their tuples go through LDS
        if (ng <= 32) {
            lean_td_pairs<M, HY, PL>(t, g0, ng, VI, lane, a.negative, a.criterion, lds_pow, hand_in ? stage : nullptr,
                                     hand_out ? stage : nullptr, k0, best, xtop, pu);
        } else {
            lean_td_chunks<M, false, HY, PL>(t, g0, g1, lane, WAVE, VI, a.negative, a.criterion, lds_pow, hand_in ? stage : nullptr,
                                             hand_out ? stage : nullptr, k0, best, xtop, pu);
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        hand_in = hand_out;
    }
    if (HY) {
        bool mine = false;
        const int none = lean_hybrid_pick(t, grp_off[1], grp_off[G + 1], V, lane, WAVE,
                                          [](double &d, int &i) { team_argmin<WAVE>(d, i, nullptr, nullptr); }, best, mine, xtop, PL ? xn : 0);
        lean_write_placement(a.out, q, V, mine ? best.v : none, mine, lane == 0, best);
        return;
    }
    const int my_best = best.v;
    double wkey = best.key;
    int win = best.v;
    team_argmin<WAVE>(wkey, win, nullptr, nullptr);
    lean_write_placement(a.out, q, V, win, my_best == win, lane == 0, best);
}

// Bottom-up kernel of the wavefront-sized teams: level lists and S tuples of one query after the other, into the pool.
// FUSED (k_lean_both, APPLES_LEAN_FUSED=1): the team runs the query's top-down pass right behind its bottom-up pass -- one launch
// and one tail per device batch instead of two; measured slower (register spills: launch_sweep_lean), not the default.
// PL: the tree has polytomies (child records, lean_poly_S).  Trees of more than LEAN_MAX_LEVELS - 2 levels: the per-level
// offsets sit in LDS as a window of LEAN_MAX_LEVELS that follows the walk upward.
template <int M, bool FUSED = false, bool PL = false>
__device__ void lean_up_loop(const SweepArgs &a, LeanUpShared &sh, const double *lds_pow = nullptr) {
    const int lane = threadIdx.x & (WAVE - 1);
    const int wave = threadIdx.x / WAVE;
    LeanWave &L = sh.w[wave];
    double2 (*stage)[WAVE] = L.stage;
    const DevTree &T = a.tree;
    const int4 *__restrict__ pe = reinterpret_cast<const int4 *>(T.pe);
    const int64_t team = (int64_t)blockIdx.x * (APPLES_TPB / WAVE) + wave;
    const LeanQueue qu = lean_queue(a);
    const int64_t n_work = qu.n_work;
    const unsigned long long below = (1ull << lane) - 1ull;
    // diagnostic (APPLES_LEAN_PROFILE): cycles per phase of this team, added to a.prof[phase] once per query by lane 0
    unsigned long long pc[7] = {0, 0, 0, 0, 0, 0, 0}, pn[4] = {0, 0, 0, 0}, tk = 0;
    const bool prof = a.prof != nullptr;
#define LEAN_TICK(slot) do { if (prof) { const unsigned long long now_ = __builtin_readcyclecounter(); pc[slot] += now_ - tk; tk = now_; } } while (0)
    if (prof) tk = __builtin_readcyclecounter();
    while (true) {
        // dynamic scheduling: one atomic add per query, broadcast to the wavefront
        int wq = 0;
        if (lane == 0) wq = atomicAdd(a.cursor, 1);
        int64_t w = __builtin_amdgcn_readfirstlane(wq);  // (scalar: the query's pointers and counts then live in SGPRs)
        if (a.w_mod > 1) w = w * a.w_mod + a.w_rem;     // (this launch's share of the queue: launch_sweep_lean's halves)
        if (w >= n_work) break;
        const int64_t q = lean_queue_at(a, qu, w);
        const int n = a.n_obs[q];
        if (n == 0) continue;
        const int32_t *o_node = a.obs_node + row_start(a.row_off, q, a.obs_cap);
        const double *o_dist = a.obs_dist + row_start(a.row_off, q, a.obs_cap);
        const int32_t *cg = a.cnt_gt + q * (int64_t)(T.height + 2);
        // the query's share of the pool (one atomic add per query); out of pool = the workgroup-sized teams take it
        const int qcap = lean_query_cap(n);
        // (32-bit cursor: ensure_workspace keeps batch x the largest share below 2^32, so the sum of every query's request cannot
        // wrap it; asking the cursor first -- a load of the one line every team's atomic add goes to -- doubled the sweep's time
        // at 50 000 queries per batch)
        unsigned int off0 = 0;
        if (lane == 0) off0 = atomicAdd(a.pool_cursor, (unsigned int)qcap);
        const int64_t off = (int64_t)(unsigned int)__builtin_amdgcn_readfirstlane((int)off0);
        int32_t *grp_off = a.grp_off + q * (int64_t)a.grp_stride;
        int32_t *xoff = grp_off + (T.height + 4);  // (PL) child records before each group's
        if (off + qcap > a.lean_cap1) {
            if (lane == 0) { a.lean_meta[q] = make_int4(0, -1, 0, 0); a.overflow_list[atomicAdd(a.overflow_count, 1)] = (int32_t)q; }
            continue;
        }
        const int64_t cap = qcap - 1;
        LeanTeam t = lean_pool_view(a.lean, a.lean_cap1, off, a.lean_leaf, team, a.lean_leaf1);
        t.BP = a.blk_pool;
        LEAN_TICK(0);

        // ------------------------------------------------------------ up front: the per-level offsets into LDS; parent and
        // edge length of every observed leaf (independent gathers, all in flight together) into the team's arrays
        // (a deep tree: the window [wlo, wlo + LEAN_MAX_LEVELS) of the offsets, reloaded when the walk leaves it below)
        const bool deep = T.height + 2 > LEAN_MAX_LEVELS;
        int wlo = 0;
        if (deep) wlo = max(0, T.level[o_node[0]] + 2 - LEAN_MAX_LEVELS);
        for (int i = lane; i < min(T.height + 2 - wlo, LEAN_MAX_LEVELS); i += WAVE) L.cg[i] = cg[wlo + i];
        for (int j0 = 0; j0 < n; j0 += 4 * WAVE) {
            int4 r[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j = j0 + u * WAVE + lane;
                r[u] = pe[o_node[j < n ? j : n - 1]];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j = j0 + u * WAVE + lane;
                if (j < n) { t.LP[j] = r[u].x; t.LE[j] = pe_len(r[u]); }
            }
        }
        const int lvl_first = T.level[o_node[0]];
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");

        // ------------------------------------------------------------ bottom-up: the level lists (Subtree.py:23-43) and the S
        // tuples (OLS.py:25-44) level by level.  The list of a level = the parents of the level below's internal nodes and
        // observed leaves, merged by node id.  A list of at most 64 entries lives in the lanes' registers (entry k in
        // lane k) with {parent, edge length} of its keys already on the way; a level that fits one merge step (list +
        // observed leaves <= 64 keys) is merged out of LDS windows filled from registers -- no dependent memory round trip.
        int lvl = lvl_first, base = 0, n_par = 0, G = 0, kid_base = 0;
        bool overflow = false, prev_staged = false;
        int cK = 0, pf_par = 0;
        int2 cD = make_int2(0, 0), cN = make_int2(0, 0);
        double2 cE = make_double2(0, 0), cDD = make_double2(0, 0);
        double pf_e = 0;
        int lo = L.cg[lvl + 1 - wlo], hi = L.cg[lvl - wlo];  // observed leaves of this level: obs[lo, hi)
        int n_leaf = hi - lo;
        const int xtop = qcap - 1;  // (PL) child records: entry slots from the top of the query's range down
        int xc = 0, xg = 0, xsl = 0;  // ... how many there are, how many there were before the current list's, and how many of the
                                      // current list's are left to lean_poly_S (runs that a merge step's end cut)
        int lw_node = 0, lw_par = 0;
        double lw_e = 0, lw_dist = 0;
        if (n_leaf <= WAVE && lane < n_leaf) { lw_node = o_node[lo + lane]; lw_par = t.LP[lo + lane]; lw_e = t.LE[lo + lane]; lw_dist = o_dist[lo + lane]; }
        LEAN_TICK(1);
        while (true) {
            if (n_par + n_leaf == 1 && hi == n) break;  // one node left in the frontier: the LCA (Subtree.py:36-43), entry `base`
            if ((int64_t)base + 2 * (int64_t)n_par + n_leaf + (PL ? (int64_t)xc + n_par + n_leaf : 0) > cap) { overflow = true; break; }
            if (lane == 0) { grp_off[G] = base; if (PL) xoff[G + 1] = xc; }
            const int next_base = base + n_par;
            // ---- S tuples of this level's internal nodes
            if (n_par > 0 && n_par <= WAVE) {
                double r[6];
                // (PL) a polytomy's tuple is in its slot already (lean_poly_fast; one that a step's end cut: lean_poly_S below) -- its lane
                // forms nothing from the entry (marked: no descriptors there) and picks the tuple up
                const bool ispoly = PL && lane < n_par && (cK & LEAN_POLY_KID) != 0;
                if (lane < n_par) {
                    node_S<M, PL>(ispoly ? make_int2(0, 0) : cD, cE, cDD, t, stage, prev_staged, kid_base, r);
                    if (!ispoly) {
                        t.T0[base + lane] = make_double2(r[0], r[1]);
                        t.T1[base + lane] = make_double2(r[2], r[3]);
                        t.T2[base + lane] = make_double2(r[4], r[5]);
                    }
                }
                if (PL && __ballot(ispoly) != 0ull) {  // (never on a binary tree)
                    if (ispoly) {
                        const double2 a0 = t.T0[base + lane], a1 = t.T1[base + lane], a2 = t.T2[base + lane];
                        r[0] = a0.x; r[1] = a0.y; r[2] = a1.x; r[3] = a1.y; r[4] = a2.x; r[5] = a2.y;
                    }
                }
                __builtin_amdgcn_wave_barrier();  // (every lane's reads of the stage precede these stores)
                if (lane < n_par) {
                    stage[0][lane] = make_double2(r[0], r[1]);
                    stage[1][lane] = make_double2(r[2], r[3]);
                    stage[2][lane] = make_double2(r[4], r[5]);
                }
            } else if (n_par > WAVE) {
                lean_S_chunks<M, PL>(t, base, next_base, lane, WAVE, stage, prev_staged, kid_base);
            }
            if (PL && xsl > 0 && !LEAN_EXP_NO_FIX) {  // this list's polytomies that a merge step's end cut: their tuples over all their children
                // (the lane state the next steps need waits in the merge windows, which are free here: nothing but scalars lives
                // across the call)
                L.ka[lane] = cK; L.pa[lane] = pf_par; L.ea[lane] = pf_e;
                L.kb[lane] = lw_node; L.pb[lane] = lw_par; L.eb[lane] = lw_e; L.db[lane] = lw_dist;
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                const LeanTeam tc = t;
                lean_poly_S<M>(tc, xtop, xg, xc, lane, WAVE, n_par <= WAVE ? stage : nullptr, base);
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                cK = L.ka[lane]; pf_par = L.pa[lane]; pf_e = L.ea[lane];
                lw_node = L.kb[lane]; lw_par = L.pb[lane]; lw_e = L.eb[lane]; lw_dist = L.db[lane];
                __builtin_amdgcn_wave_barrier();
            }
            xg = xc; xsl = 0;
            // ---- the next level's list
            int merged;
            if (n_par <= WAVE && n_par + n_leaf <= WAVE) {
                if (lane < n_par) { L.ka[lane] = cK; L.pa[lane] = pf_par; L.ea[lane] = pf_e; }  // (PL: cK may carry LEAN_POLY_KID)
                if (lane < n_leaf) { L.kb[lane] = lw_node; L.pb[lane] = lw_par; L.eb[lane] = lw_e; L.db[lane] = lw_dist; }
                __builtin_amdgcn_wave_barrier();
                const int tot = n_par + n_leaf;
                const bool active = lane < tot;
                int i_lo = max(0, lane - n_leaf), i_hi = min(lane, n_par);  // i = keys of the list among the lane smallest
                while (i_lo < i_hi) {
                    const int i = (i_lo + i_hi) >> 1;
                    if (lean_key_node<PL>(L.ka[i]) < L.kb[lane - 1 - i]) i_lo = i + 1; else i_hi = i;
                }
                const int i = i_lo, j = lane - i_lo;
                const int kaf = i < n_par ? L.ka[i] : 0x7fffffff;
                const int ka = i < n_par ? lean_key_node<PL>(kaf) : 0x7fffffff, kb = j < n_leaf ? L.kb[j] : 0x7fffffff;
                const bool from_a = ka < kb;
                const int key = from_a ? ka : kb;
                const int desc = from_a ? ((base + i + 1) | (PL ? (kaf & LEAN_POLY_KID) : 0)) : -(lo + j) - 2;
                int par = -3;
                double e = 0, dist = 0;
                if (active) {
                    if (from_a) { par = L.pa[i]; e = L.ea[i]; }
                    else { par = L.pb[j]; e = L.eb[j]; dist = L.db[j]; }
                }
                const int prev = __shfl_up(par, 1, WAVE);
                const bool first = active && (lane == 0 || par != prev);
                const int next_desc = __shfl_down(desc, 1, WAVE), next_key = __shfl_down(key, 1, WAVE);
                const int next_first = __shfl_down(first ? 1 : 0, 1, WAVE);
                const double next_e = shfl_down_f64(e, 1), next_dist = shfl_down_f64(dist, 1);
                const bool two = lane + 1 < tot && !next_first;
                const unsigned long long fm = __ballot(first);
                merged = __popcll(fm);
                int kflag = 0;  // (PL) the run this lane begins has a third key: the node is a polytomy
                bool fhead = false;
                int fx0 = 0, fm_ = 0;
                if (PL) {
                    const LeanRunMasks rm = lean_run_masks(fm, tot, 0);
                    kflag = ((rm.third >> lane) & 1ull) ? LEAN_POLY_KID : 0;
                    if (rm.third != 0ull) {  // (never on a binary tree) every run lies inside this one step: finished here
                        const LeanTeam tc = t;
                        unsigned long long pm_ = 0ull;
                        bool h_ = false;
                        int a_ = 0, b_ = 0;
                        xc += lean_poly_fast<M>(tc, xtop, xc, fm, rm.third, tot, true, next_base, desc, key, e, dist, stage, n_par > 0, base, lane,
                                                0ull, 0ull, pm_, h_, a_, b_);
                        fhead = h_; fx0 = a_; fm_ = b_;
                    }
                }
                if (first) {
                    const int c = __popcll(fm & below);
                    const int2 eD = (PL && fhead) ? make_int2(fx0, LEAN_POLY_SELF | fm_) : make_int2(desc, two ? next_desc : 0);
                    const int2 eN = make_int2(key, two ? next_key : -1);
                    const double2 eE = make_double2(e, two ? next_e : 0.0), eDD = make_double2(dist, two ? next_dist : 0.0);
                    L.oK[c] = par | kflag; L.oD[c] = eD; L.oN[c] = eN; L.oE[c] = eE; L.oDD[c] = eDD;
                    t.K[next_base + c] = par | kflag; t.D[next_base + c] = eD; t.N[next_base + c] = eN; t.E[next_base + c] = eE; t.DD[next_base + c] = eDD;
                }
                __builtin_amdgcn_wave_barrier();
                if (lane < merged) { cK = L.oK[lane]; cD = L.oD[lane]; cN = L.oN[lane]; cE = L.oE[lane]; cDD = L.oDD[lane]; }
            } else {
                merged = lean_merge<M, PL>(t, base, n_par, o_node, o_dist, lo, n_leaf, next_base, pe, L.ka, L.kb, lane, xtop, xc, xsl);
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                if (merged <= WAVE && lane < merged) {
                    cK = t.K[next_base + lane]; cD = t.D[next_base + lane]; cN = t.N[next_base + lane];
                    cE = t.E[next_base + lane]; cDD = t.DD[next_base + lane];
                }
            }
            // {parent, edge length} of the new list's keys: on the way while the next step computes its S tuples
            if (merged <= WAVE && lane < merged) {
                const int4 r = pe[lean_key_node<PL>(cK)];
                pf_par = r.x;
                pf_e = pe_len(r);
            }
            if (prof) { if (n_par <= WAVE && n_par + n_leaf <= WAVE) { LEAN_TICK(2); ++pn[0]; } else { LEAN_TICK(3); ++pn[1]; } }
            prev_staged = n_par > 0 && n_par <= WAVE;
            kid_base = base;
            base = next_base;
            n_par = merged;
            ++G;
            --lvl;
            if (deep && lvl < wlo) {  // (the walk left the window: the next LEAN_MAX_LEVELS levels up)
                wlo = max(0, lvl + 2 - LEAN_MAX_LEVELS);
                __builtin_amdgcn_wave_barrier();
                for (int i = lane; i < min(T.height + 2 - wlo, LEAN_MAX_LEVELS); i += WAVE) L.cg[i] = cg[wlo + i];
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            }
            lo = L.cg[lvl + 1 - wlo]; hi = L.cg[lvl - wlo];
            n_leaf = hi - lo;
            if (n_leaf <= WAVE && lane < n_leaf) { lw_node = o_node[lo + lane]; lw_par = t.LP[lo + lane]; lw_e = t.LE[lo + lane]; lw_dist = o_dist[lo + lane]; }
        }
        if (overflow) {  // hand the query to the workgroup-sized teams with full-size scratch
            if (lane == 0) { a.lean_meta[q] = make_int4(0, -1, 0, 0); a.overflow_list[atomicAdd(a.overflow_count, 1)] = (int32_t)q; }
            if (prof && lane == 0) atomicAdd(a.prof + 13, 1ull);
            continue;
        }
        // internal valid nodes: base; the LCA's entry sits at that index.  What the top-down kernel needs of this query:
        if (PL && xsl > 0 && !LEAN_EXP_NO_FIX) {  // the LCA is a polytomy that a merge step's end cut: its children's tuples into their records
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            const LeanTeam tc = t;
            lean_poly_S<M>(tc, xtop, xg, xc, lane, WAVE, nullptr, 0);
        }
        if (lane == 0) {
            grp_off[G] = base; grp_off[G + 1] = base + 1; a.lean_meta[q] = make_int4((int)off, G, base, xc);
            if (PL) xoff[G + 1] = xc;
        }
        if (FUSED) {
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");  // (the group offsets and the entries: this wavefront's own writes)
            if (a.debug_phase != 1) lean_down_one<M, false, PL>(a, lds_pow, stage, q, n, off, G, base, lane, xc);
        }
        if (prof) {
            if (lane == 0) {
                for (int k = 0; k < 4; ++k) { atomicAdd(a.prof + k, pc[k]); pc[k] = 0; }
                for (int k = 0; k < 2; ++k) { atomicAdd(a.prof + 8 + k, pn[k]); pn[k] = 0; }
                atomicAdd(a.prof + 12, 1ull);
            }
            tk = __builtin_readcyclecounter();
        }
    }
#undef LEAN_TICK
}

// Top-down kernel of the wavefront-sized teams: a node forms R for each valid child (all_R_values), solves it
// (placement_per_edge) and evaluates its residual (error_per_edge); an internal child's tuple becomes lift(R) over its own
// edge; then the query's arg-min (apples/Algorithm.py:74-91)
template <int M, bool HY = false, bool PL = false>
__device__ void lean_down_loop(const SweepArgs &a, LeanDownShared &sh, double (*pu)[WAVE] = nullptr) {
    const int lane = threadIdx.x & (WAVE - 1);
    const int wave = threadIdx.x / WAVE;
    const double *lds_pow = sh.pow;
    double2 (*stage)[WAVE] = sh.stage[wave];
    const DevTree &T = a.tree;
    const LeanQueue qu = lean_queue(a);
    unsigned long long pc[7] = {0, 0, 0, 0, 0, 0, 0}, pn[4] = {0, 0, 0, 0}, tk = 0;
    const bool prof = a.prof != nullptr && !HY && !PL;  // (the cycle counters cover the MLSE / ME form on binary trees)
#define LEAN_TICK(slot) do { if (prof) { const unsigned long long now_ = __builtin_readcyclecounter(); pc[slot] += now_ - tk; tk = now_; } } while (0)
    if (prof) tk = __builtin_readcyclecounter();
    while (true) {
        int wq = 0;
        if (lane == 0) wq = atomicAdd(a.cursor, 1);
        int64_t w = __builtin_amdgcn_readfirstlane(wq);  // (scalar: the query's pointers and counts then live in SGPRs)
        if (a.w_mod > 1) w = w * a.w_mod + a.w_rem;
        if (w >= qu.n_work) break;
        const int64_t q = lean_queue_at(a, qu, w);
        const int n = a.n_obs[q];
        if (n == 0) continue;
        const int4 meta = a.lean_meta[q];
        const int G = meta.y, VI = meta.z;
        if (G < 0) continue;  // handed to the workgroup-sized teams by the bottom-up kernel
        if (!prof) { lean_down_one<M, HY, PL>(a, lds_pow, stage, q, n, meta.x & 0xffffffffll, G, VI, lane, meta.w, pu); continue; }
        const int V = VI + n;  // Subtree.num_nodes
        LeanTeam t = lean_pool_view(a.lean, a.lean_cap1, meta.x & 0xffffffffll, nullptr, 0, 0);
        t.BP = a.blk_pool;
        const int32_t *grp_off = a.grp_off + q * (int64_t)a.grp_stride;
        LEAN_TICK(0);
        LeanBest best;
        lean_best_init(best);
        bool hand_in = false;  // this level's lifted R tuples wait in LDS (handed over by the level above)
        for (int g = G; g >= 1; --g) {
            const int g0 = grp_off[g], g1 = grp_off[g + 1], k0 = grp_off[g - 1];
            const int ng = g1 - g0;
            const bool hand_out = g0 - k0 <= WAVE;  // the children's level has at most 64 nodes: their tuples go through LDS
            if (ng <= 32) {
                lean_td_pairs<M>(t, g0, ng, VI, lane, a.negative, a.criterion, lds_pow, hand_in ? stage : nullptr,
                                 hand_out ? stage : nullptr, k0, best);
            } else {
                lean_td_chunks<M, false>(t, g0, g1, lane, WAVE, VI, a.negative, a.criterion, lds_pow, hand_in ? stage : nullptr,
                                  hand_out ? stage : nullptr, k0, best);
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            hand_in = hand_out;
            if (prof) { if (ng <= 32) { LEAN_TICK(4); ++pn[2]; } else { LEAN_TICK(5); ++pn[3]; } }
        }

        // ------------------------------------------------------------ selection (apples/Algorithm.py:74-91)
        const int my_best = best.v;
        double wkey = best.key;
        int win = best.v;
        team_argmin<WAVE>(wkey, win, nullptr, nullptr);
        lean_write_placement(a.out, q, V, win, my_best == win, lane == 0, best);
        if (prof) {
            LEAN_TICK(6);
            if (lane == 0) {
                atomicAdd(a.prof + 0, pc[0]); pc[0] = 0;
                for (int k = 4; k < 7; ++k) { atomicAdd(a.prof + k, pc[k]); pc[k] = 0; }
                for (int k = 2; k < 4; ++k) { atomicAdd(a.prof + 8 + k, pn[k]); pn[k] = 0; }
            }
            tk = __builtin_readcyclecounter();
        }
    }
#undef LEAN_TICK
}

// ---------------------------------------------------------------------------------------------------------------------
// Workgroup-sized team (TEAM threads on one query): the queries with many observed leaves, whose levels hold thousands of
// nodes.  The same passes in their general form: S tuples and top-down steps strided over the team, TEAM keys per merge
// step (neighbours and counts through LDS instead of shuffles), a barrier where the wavefront-sized team has a fence.
template <int TEAM>
struct LeanBigShared {
    double pow[384 + 256];
    int cg[LEAN_MAX_LEVELS];
    int ka[TEAM], kb[TEAM];                        // the two key windows of a merge step
    int m_par[TEAM], m_desc[TEAM], m_key[TEAM];    // what a step's lanes tell their neighbours
    double m_e[TEAM], m_dist[TEAM];
    int mcnt[2][TEAM / WAVE];
    int xcnt[2][TEAM / WAVE];                      // (polytomies) a step's third / later siblings per wavefront
    double pu[TEAM / WAVE][6][WAVE];               // (polytomies) per wavefront: a polytomy's children's lifted tuples (lean_poly_td)
    double d[TEAM / WAVE];
    int i[TEAM / WAVE];
    int w;
};

template <int TEAM, bool PL>
__device__ __forceinline__ int lean_merge_wg(const LeanTeam &t, int base, int nA, const int32_t *__restrict__ o_node,
                                             const double *__restrict__ o_dist, int lo, int nB, int next_base,
                                             const int4 *__restrict__ pe, LeanBigShared<TEAM> &sh, int xtop, int &xc) {
    const int tid = threadIdx.x, lane = tid & (WAVE - 1), wave = tid / WAVE;
    int out = 0, ia = 0, ib = 0, carry = -2, ccnt = 0;
    const unsigned long long below = (1ull << lane) - 1ull;
    int wk_a = tid < nA ? t.K[base + tid] : 0x7fffffff, wk_b = tid < nB ? o_node[lo + tid] : 0x7fffffff;
    while (ia < nA || ib < nB) {
        const int wa = min(nA - ia, TEAM), wb = min(nB - ib, TEAM);
        sh.ka[tid] = wk_a;
        sh.kb[tid] = wk_b;
        __syncthreads();
        const int tot = min(wa + wb, TEAM);
        const bool active = tid < tot;
        int i_lo = max(0, tid - wb), i_hi = min(tid, wa);
        while (i_lo < i_hi) {
            const int i = (i_lo + i_hi) >> 1;
            if (lean_key_node<PL>(sh.ka[i]) < sh.kb[tid - 1 - i]) i_lo = i + 1; else i_hi = i;
        }
        const int i = i_lo, j = tid - i_lo;
        const int kaf = i < wa ? sh.ka[i] : 0x7fffffff;  // (PL: a polytomy's key carries LEAN_POLY_KID)
        const int ka = i < wa ? lean_key_node<PL>(kaf) : 0x7fffffff, kb = j < wb ? sh.kb[j] : 0x7fffffff;
        const bool from_a = ka < kb;
        const int key = from_a ? ka : kb;
        const int desc = from_a ? ((base + ia + i + 1) | (PL ? (kaf & LEAN_POLY_KID) : 0)) : -(lo + ib + j) - 2;
        const unsigned long long am = __ballot(active && from_a);
        if (lane == 0) sh.mcnt[1][wave] = __popcll(am);
        __syncthreads();  // (the windows are read: the next step may overwrite them after its own first barrier)
        int ca = 0;
#pragma unroll
        for (int w = 0; w < TEAM / WAVE; ++w) ca += sh.mcnt[1][w];
        const int nia = ia + ca, nib = ib + tot - ca;
        wk_a = nia + tid < nA ? t.K[base + nia + tid] : 0x7fffffff;  // the next step's windows, before this step's gather is waited for
        wk_b = nib + tid < nB ? o_node[lo + nib + tid] : 0x7fffffff;
        int par = -3;
        double e = 0, dist = 0;
        if (active) {
            if (from_a) {
                const int4 r = pe[key];
                par = r.x;
                e = pe_len(r);
            } else {
                par = t.LP[lo + ib + j];
                e = t.LE[lo + ib + j];
                dist = o_dist[lo + ib + j];
            }
        }
        sh.m_par[tid] = par; sh.m_desc[tid] = desc; sh.m_key[tid] = key; sh.m_e[tid] = e; sh.m_dist[tid] = dist;
        __syncthreads();
        const bool first = active && par != (tid == 0 ? carry : sh.m_par[tid - 1]);
        const bool two = tid + 1 < tot && sh.m_par[tid + 1] == par;
        const unsigned long long fm = __ballot(first);
        if (lane == 0) sh.mcnt[0][wave] = __popcll(fm);
        const int last_par = sh.m_par[tot - 1];
        // (PL) the lane's place in its run of siblings, as lean_run_of with the neighbours in LDS
        bool extra = false, xfirst = false, third = false;
        unsigned long long m3 = 0, m1 = 0;
        int ncc = 0;
        if (PL) {
            const bool c = carry == par;
            const bool s1 = tid >= 1 ? sh.m_par[tid - 1] == par : (c && ccnt >= 1);
            const bool s2 = s1 && (tid >= 2 ? sh.m_par[tid - 2] == par : (c && ccnt >= 2 - tid));
            const bool s3 = s2 && (tid >= 3 ? sh.m_par[tid - 3] == par : (c && ccnt >= 3 - tid));
            extra = active && s2;
            xfirst = extra && !s3;
            third = first && tid + 2 < tot && sh.m_par[tid + 2] == par;
            m3 = __ballot(xfirst);
            m1 = __ballot(extra && !xfirst);
            if (lane == 0) { sh.xcnt[0][wave] = __popcll(m3); sh.xcnt[1][wave] = __popcll(m1); }
            // keys of the step's last run (capped at 3; with the carried ones when the whole step belongs to the run before)
            const bool a1 = tot >= 2 && sh.m_par[tot - 2] == last_par, a2 = a1 && tot >= 3 && sh.m_par[tot - 3] == last_par;
            ncc = a2 ? 3 : (a1 ? 2 : 1);
            if (ncc == tot && carry == last_par) ncc = min(ncc + ccnt, 3);
        }
        __syncthreads();
        int before = 0, total = 0, xb3 = 0, xb1 = 0, xt3 = 0, xt1 = 0;
#pragma unroll
        for (int w = 0; w < TEAM / WAVE; ++w) {
            if (w < wave) before += sh.mcnt[0][w];
            total += sh.mcnt[0][w];
            if (PL) {
                if (w < wave) { xb3 += sh.xcnt[0][w]; xb1 += sh.xcnt[1][w]; }
                xt3 += sh.xcnt[0][w]; xt1 += sh.xcnt[1][w];
            }
        }
        if (PL && extra) {  // a child record (lean_poly_records, with the counts of the wavefronts before)
            const int en = next_base + out + before + __popcll(fm & (below | (1ull << lane))) - 1;
            const int x0 = xc + 3 * (xb3 + __popcll(m3 & below)) + xb1 + __popcll(m1 & below);
            const int x = xfirst ? x0 + 2 : x0;
            t.D[xtop - x] = make_int2(desc, xfirst ? 2 : 3);
            t.N[xtop - x] = make_int2(key, en);
            t.E[xtop - x] = make_double2(e, dist);
            if (xfirst) {
                t.D[xtop - x0] = make_int2(0, 0);
                t.N[xtop - x0] = make_int2(-1, en);
                t.D[xtop - x0 - 1] = make_int2(0, 1);
                t.N[xtop - x0 - 1] = make_int2(-1, en);
                // (the entry's own store of its key: this step's carries the mark already -- `third` -- an earlier step's is long done)
                t.K[en] = par | LEAN_POLY_KID;
            }
        }
        if (PL) { xc += 3 * xt3 + xt1; ccnt = ncc; }
        if (first) {
            const int at = next_base + out + before + __popcll(fm & below);
            t.K[at] = (PL && third) ? (par | LEAN_POLY_KID) : par;
            t.D[at] = make_int2(desc, two ? sh.m_desc[tid + 1] : 0);
            t.N[at] = make_int2(key, two ? sh.m_key[tid + 1] : -1);
            t.E[at] = make_double2(e, two ? sh.m_e[tid + 1] : 0.0);
            t.DD[at] = make_double2(dist, two ? sh.m_dist[tid + 1] : 0.0);
        } else if (tid == 0 && active && !(PL && extra)) {  // the second child of the previous step's last entry
            const int at = next_base + out - 1;
            reinterpret_cast<int *>(t.D + at)[1] = desc;
            reinterpret_cast<int *>(t.N + at)[1] = key;
            reinterpret_cast<double *>(t.E + at)[1] = e;
            reinterpret_cast<double *>(t.DD + at)[1] = dist;
        }
        carry = last_par;
        ia = nia;
        ib = nib;
        out += total;
        // (no barrier here: the next step writes the windows, which everybody has finished reading two barriers ago, and
        // reaches m_par / mcnt[0] only after its own barriers)
    }
    return out;
}

// team-wide arg-min over (key, id) for a team of TEAM threads (sweep_math.h:team_argmin is tied to 256)
template <int TEAM>
__device__ __forceinline__ void lean_argmin_wg(double &d, int &i, LeanBigShared<TEAM> &sh) {
    for (int o = WAVE / 2; o > 0; o >>= 1) {
        const double d2 = shfl_down_f64s(d, o);
        const int i2 = __shfl_down(i, o, WAVE);
        if (d2 < d || (d2 == d && i2 < i)) { d = d2; i = i2; }
    }
    const int w = threadIdx.x / WAVE;
    __syncthreads();
    if ((threadIdx.x & (WAVE - 1)) == 0) { sh.d[w] = d; sh.i[w] = i; }
    __syncthreads();
    d = sh.d[0]; i = sh.i[0];
    for (int k = 1; k < TEAM / WAVE; ++k) {
        const double d2 = sh.d[k];
        const int i2 = sh.i[k];
        if (d2 < d || (d2 == d && i2 < i)) { d = d2; i = i2; }
    }
}

template <int M, int TEAM, bool HY = false, bool PL = false>
__device__ void lean_big_loop(const SweepArgs &a, int64_t nq, LeanBigShared<TEAM> &sh) {
    const int tid = threadIdx.x;
    const double *lds_pow = sh.pow;
    const DevTree &T = a.tree;
    const int4 *__restrict__ pe = reinterpret_cast<const int4 *>(T.pe);
    LeanTeam t = lean_team(a.lean, blockIdx.x, a.lean_cap1, a.lean_leaf1);
    t.BP = a.blk_pool;
    int32_t *grp_off = a.grp_off + (int64_t)blockIdx.x * a.grp_stride;
    int32_t *xoff = grp_off + (T.height + 4);  // (PL) child records before each group's
    const int xtop = (int)a.lean_cap1 - 1;     // (PL) child records: the team's entry slots from the top down
    const bool deep = T.height + 2 > LEAN_MAX_LEVELS;  // the per-level offsets in LDS as a window that follows the walk
    // work: a device-side list, or (routed queries) three lists by size class, largest first
    const int r0 = a.route_classes ? a.work_count[4] : 0, r1 = a.route_classes ? a.work_count[5] : 0;
    const int64_t n_work = a.route_classes ? (int64_t)r0 + r1 + a.work_count[6] : (a.work_count ? *a.work_count : nq);
    while (true) {
        if (tid == 0) sh.w = atomicAdd(a.cursor, 1);
        __syncthreads();
        const int64_t w = sh.w;
        __syncthreads();
        if (w >= n_work) break;
        int64_t q;
        if (a.route_classes) q = w < r0 ? a.work_list[w] : (w < r0 + r1 ? a.work_list[a.cls_stride + (w - r0)] : a.work_list[2 * a.cls_stride + (w - r0 - r1)]);
        else q = a.work_list ? a.work_list[w] : w;
        const int n = a.n_obs[q];
        if (n == 0) continue;
        const int32_t *o_node = a.obs_node + row_start(a.row_off, q, a.obs_cap);
        const double *o_dist = a.obs_dist + row_start(a.row_off, q, a.obs_cap);
        const int32_t *cg = a.cnt_gt + q * (int64_t)(T.height + 2);
        const int lvl_first = T.level[o_node[0]];
        int wlo = deep ? max(0, lvl_first + 2 - LEAN_MAX_LEVELS) : 0;
        for (int i = tid; i < min(T.height + 2 - wlo, LEAN_MAX_LEVELS); i += TEAM) sh.cg[i] = cg[wlo + i];
        for (int j = tid; j < n; j += TEAM) {
            const int4 r = pe[o_node[j]];
            t.LP[j] = r.x;
            t.LE[j] = pe_len(r);
        }
        __syncthreads();
        int lvl = lvl_first, base = 0, n_par = 0, G = 0, xc = 0, xg = 0;
        while (true) {
            if (deep && lvl < wlo) {  // (the walk left the window: the next LEAN_MAX_LEVELS levels up)
                wlo = max(0, lvl + 2 - LEAN_MAX_LEVELS);
                __syncthreads();
                for (int i = tid; i < min(T.height + 2 - wlo, LEAN_MAX_LEVELS); i += TEAM) sh.cg[i] = cg[wlo + i];
                __syncthreads();
            }
            const int lo = sh.cg[lvl + 1 - wlo], hi = sh.cg[lvl - wlo];
            const int n_leaf = hi - lo;
            if (n_par + n_leaf == 1 && hi == n) break;  // the LCA (Subtree.py:36-43), entry `base`
            if (tid == 0) { grp_off[G] = base; if (PL) xoff[G + 1] = xc; }
            const int next_base = base + n_par;
            lean_S_chunks<M, PL>(t, base, next_base, tid, TEAM, nullptr, false, 0);  // S tuples of this level (the level below's are complete)
            if (PL && xc > xg) {  // this list's polytomies: their tuples over all their children
                __syncthreads();
                const LeanTeam tc = t;
                lean_poly_S<M>(tc, xtop, xg, xc, tid, TEAM, nullptr, 0);
            }
            xg = xc;
            const int merged = lean_merge_wg<TEAM, PL>(t, base, n_par, o_node, o_dist, lo, n_leaf, next_base, pe, sh, xtop, xc);
            __syncthreads();
            base = next_base;
            n_par = merged;
            ++G;
            --lvl;
        }
        const int VI = base, V = base + n;
        if (PL && xc > xg) {  // the LCA is a polytomy: its children's tuples into their records
            __syncthreads();
            const LeanTeam tc = t;
            lean_poly_S<M>(tc, xtop, xg, xc, tid, TEAM, nullptr, 0);
        }
        if (tid == 0) { grp_off[G] = VI; grp_off[G + 1] = VI + 1; if (PL) xoff[G + 1] = xc; }
        __syncthreads();
        if (a.debug_phase == 1) continue;
        LeanBest best;
        lean_best_init(best);
        for (int g = G; g >= 1; --g) {
            const int g0 = grp_off[g], g1 = grp_off[g + 1];
            lean_td_chunks<M, true, HY, PL>(t, g0, g1, tid, TEAM, VI, a.negative, a.criterion, lds_pow, nullptr, nullptr, 0, best, xtop, sh.pu[tid / WAVE]);
            __syncthreads();
        }
        if (HY) {
            bool mine = false;
            const int none = lean_hybrid_pick(t, grp_off[1], VI + 1, V, tid, TEAM,
                                              [&sh](double &d, int &i) { lean_argmin_wg<TEAM>(d, i, sh); }, best, mine, xtop, PL ? xc : 0);
            lean_write_placement(a.out, q, V, mine ? best.v : none, mine, tid == 0, best);
            __syncthreads();
            continue;
        }
        const int my_best = best.v;
        double wkey = best.key;
        int win = best.v;
        lean_argmin_wg<TEAM>(wkey, win, sh);
        lean_write_placement(a.out, q, V, win, my_best == win, tid == 0, best);
        __syncthreads();
    }
}

template <int M, int TEAM, bool HY = false, bool PL = false>
__global__ __launch_bounds__(TEAM, TEAM / 256 * 2 > 2 ? 2 : TEAM / 256 * 2) void k_sweep_lean_big(SweepArgs a, int64_t nq) {
    __shared__ LeanBigShared<TEAM> sh;
    for (int i = threadIdx.x; i < 384; i += TEAM) sh.pow[i] = (&kPowLogTab[0][0])[i];
    for (int i = threadIdx.x; i < 256; i += TEAM) sh.pow[384 + i] = __longlong_as_double((long long)kExpTab[i]);
    __syncthreads();
    lean_big_loop<M, TEAM, HY, PL>(a, nq, sh);
}

template <int TEAM, bool HY, bool PL>
void launch_lean_big_m(const SweepArgs &a, int64_t nq, dim3 grid, hipStream_t st) {
    const dim3 block(TEAM);
    switch (a.method) {
        case APPLES_FM: hipLaunchKernelGGL((k_sweep_lean_big<APPLES_FM, TEAM, HY, PL>), grid, block, 0, st, a, nq); break;
        case APPLES_BME: hipLaunchKernelGGL((k_sweep_lean_big<APPLES_BME, TEAM, HY, PL>), grid, block, 0, st, a, nq); break;
        case APPLES_BE: hipLaunchKernelGGL((k_sweep_lean_big<APPLES_BE, TEAM, HY, PL>), grid, block, 0, st, a, nq); break;
        default: hipLaunchKernelGGL((k_sweep_lean_big<APPLES_OLS, TEAM, HY, PL>), grid, block, 0, st, a, nq); break;
    }
}

template <int TEAM>
void launch_lean_big_t(const SweepArgs &a, int64_t nq, dim3 grid, hipStream_t st) {
    const bool poly = a.tree.max_children > 2 || a.tree.force_poly != 0;  // (the kernels for trees with polytomies: child records, lean_poly_S / lean_poly_td; the knob: timing experiments on binary trees)
    if (a.criterion == APPLES_HYBRID) {
        if (poly) launch_lean_big_m<TEAM, true, true>(a, nq, grid, st);
        else launch_lean_big_m<TEAM, true, false>(a, nq, grid, st);
    } else {
        if (poly) launch_lean_big_m<TEAM, false, true>(a, nq, grid, st);
        else launch_lean_big_m<TEAM, false, false>(a, nq, grid, st);
    }
}

template <int M, bool PL = false>
__global__ __launch_bounds__(APPLES_TPB, PL ? LEAN_UP_WAVES_PL : LEAN_UP_WAVES) void k_lean_up(SweepArgs a) {
    __shared__ LeanUpShared sh;
    lean_up_loop<M, false, PL>(a, sh);
}

struct LeanBothShared {
    LeanUpShared up;
    double pow[384 + 256];  // libm pow tables (sweep_math.h)
};

template <int M>
__global__ __launch_bounds__(APPLES_TPB, LEAN_UP_WAVES) void k_lean_both(SweepArgs a) {
    __shared__ LeanBothShared sh;
    for (int i = threadIdx.x; i < 384; i += APPLES_TPB) sh.pow[i] = (&kPowLogTab[0][0])[i];
    for (int i = threadIdx.x; i < 256; i += APPLES_TPB) sh.pow[384 + i] = __longlong_as_double((long long)kExpTab[i]);
    __syncthreads();
    lean_up_loop<M, true>(a, sh.up, sh.pow);
}

template <int M, bool HY = false, bool PL = false>
__global__ __launch_bounds__(APPLES_TPB, LEAN_DOWN_WAVES) void k_lean_down(SweepArgs a) {
    __shared__ typename std::conditional<PL, LeanDownSharedPL, LeanDownShared>::type sh;
    for (int i = threadIdx.x; i < 384; i += APPLES_TPB) sh.pow[i] = (&kPowLogTab[0][0])[i];
    for (int i = threadIdx.x; i < 256; i += APPLES_TPB) sh.pow[384 + i] = __longlong_as_double((long long)kExpTab[i]);
    __syncthreads();
    if constexpr (PL) lean_down_loop<M, HY, PL>(a, sh, sh.pu[threadIdx.x / WAVE]);
    else lean_down_loop<M, HY, PL>(a, sh);
}


// ---------------------------------------------------------------------------------------------------------------------
// Clade blocks (DevAlign::blk_*): the part of the sweep that lies inside whole subtrees of one cluster.  A tile = up to 64
// of the queries that accepted one cluster; a wavefront takes a tile, a lane = a query, and walks the internal nodes of the
// cluster's blocks on the static schedule of their records -- bottom-up in post-order (k_blocks_up: all_S_values,
// apples/OLS.py:12-44 ...), top-down in reverse (k_blocks_down: all_R_values, placement_per_edge, error_per_edge, the
// running arg-min of apples/Algorithm.py:74-91).  The structure (children, edge lengths) is wave-uniform: scalar loads; the
// tuples live in the pool as [slot][component][lane] -- a wavefront's loads and stores are whole 512-byte rows; the
// distances of the blocks' leaves are the member distances the cluster-major distance pass left in the queries' rows.
// Arithmetic and order are node_S's and lean_td_kid's: every valid child in file order, the parent term last.
// The first NS components of an S tuple inside a block do not depend on the query (every leaf of the block is observed: the count
// and the path-length sums are the block's own): OLS / BME S, Sd, Sd2; BE S, Sd; FM S.  The host formed them with node_S's
// operations (api.hip:build_blocks, BlockArgs::stat): the walks store and load the other components only.
template <int M>
struct BlkStat {
    static constexpr int NS = (M == APPLES_OLS || M == APPLES_BME) ? 3 : (M == APPLES_BE ? 2 : 1);
};
template <int NS>
__device__ __forceinline__ void blk_load_dyn(const double *p, double *S) {
#pragma unroll
    for (int x = NS; x < 6; ++x) S[x] = p[x * 64];
}
template <int NS>
__device__ __forceinline__ void blk_store_dyn(double *p, const double *S) {
#pragma unroll
    for (int x = NS; x < 6; ++x) p[x * 64] = S[x];
}
__device__ __forceinline__ void blk_load(const double *p, double *S) {
#pragma unroll
    for (int x = 0; x < 6; ++x) S[x] = p[x * 64];
}
__device__ __forceinline__ void blk_store(double *p, const double *S) {
#pragma unroll
    for (int x = 0; x < 6; ++x) p[x * 64] = S[x];
}

// What a step of the walks reads besides its record: the distances of the node's leaf children, the tuple of an internal child
// that is not the node before (bottom-up: only a left child can be) and, top-down, the node's own lift(R) where the step before
// did not form it.  All of it is requested ONE STEP AHEAD and UNCONDITIONALLY (what a node does not need comes from slot 0 /
// member 0 and is dropped): the loads of a step are then a fixed number, the wait before their use is a counted one that leaves
// the step's own stores and the next step's loads in flight.  (First form: loads where needed, waited for where used -- every
// step then drained the memory queue, stores included, 8 us per step: the kernel was a queue of round trips.)  The records
// come through a per-wavefront window in LDS (one coalesced load per 128 nodes): reading them costs no place in that queue.
#define BLK_WIN 128
struct BlkOps {
    double d0, d1;
    double p0[6], p1[6], pl[6];  // (p0 / p1: the query-independent components come from the static table, the others from the pool)
};
struct BlkWin {
    int4 ri[APPLES_TPB / WAVE][BLK_WIN];
    double2 re[APPLES_TPB / WAVE][BLK_WIN];
};
struct BlkWinC {  // (BME bottom-up: the records' coefficients too -- a node of m children inside a block weighs them 1 / m)
    double2 rc[APPLES_TPB / WAVE][BLK_WIN];
};

template <int M>
__global__ __launch_bounds__(APPLES_TPB) void k_blocks_up(BlockArgs a) {
    constexpr bool BME = (M == APPLES_BME);
    __shared__ BlkWin win;
    __shared__ typename std::conditional<BME, BlkWinC, int>::type winc;
    const int lane = threadIdx.x & (WAVE - 1), wv = threadIdx.x / WAVE;
    int4 *w_ri = win.ri[wv];
    double2 *w_re = win.re[wv];
    double2 *w_rc = nullptr;
    if constexpr (BME) w_rc = winc.rc[wv];
    const int n_tiles = *a.n_tiles;
    while (true) {
        int tq = 0;
        if (lane == 0) tq = atomicAdd(a.cursor, 1);
        const int ti = __builtin_amdgcn_readfirstlane(tq);
        if (ti >= n_tiles) break;
        const int4 tile = a.tiles[ti];
        const int c = tile.x, nqt = tile.z;
        const bool in = lane < nqt;
        const int item = tile.y + (in ? lane : 0);
        const int2 it = a.items[item];
        if (tile.w < 0) continue;  // no room in the pool: the tile's queries take these leaves one by one (phase 2 told them)
        // (which of the tile's items go without blocks -- a member the reference drops, an exact match, the query's own row -- is
        // k_cluster_dist's finding, item_bad: this kernel forms every lane's tuples, so that it can run beside the selection's last phase)
        const int64_t q = it.x;
        const double *dbase = a.tmp_d + row_start(a.row_off, q, a.stride) + it.y;
        const int rb = a.rep_soff[c], ns = a.rep_soff[c + 1] - rb;
        double *pool = a.pool + ((int64_t)tile.w + 1) * 384 + lane;  // (slot 0 of the tile: the lanes' best edges, k_blocks_down)
        // the members' distances as [member][lane] behind the tuples: a lane reads its query's row eight values (a sector) at a
        // time here, and the walks then read a leaf's distance as one 512-byte row for the wavefront -- read leaf by leaf from
        // the queries' rows a visit cost 64 sectors for 64 values, half of all the bytes the walks moved
        double *dT = pool + (int64_t)ns * 384;
        const int sz = a.rep_moff[c + 1] - a.rep_moff[c];
        for (int m0 = 0; m0 < sz; m0 += 8) {
            double v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = dbase[m0 + k < sz ? m0 + k : sz - 1];
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (m0 + k < sz) dT[(int64_t)(m0 + k) * 64] = v[k];
        }
        // (what a node does not need is read from one place, the same for every lane: a sector, not a row)
        constexpr int NS = BlkStat<M>::NS;
        const double *stat = a.stat + (int64_t)rb * 3;
        auto fetch = [&](const int4 &ri, int j, BlkOps &o) __attribute__((always_inline)) {
            o.d0 = dT[(int64_t)(ri.x < 0 ? -ri.x - 1 : 0) * 64];
            o.d1 = dT[(int64_t)(ri.y < 0 ? -ri.y - 1 : 0) * 64];
#pragma unroll
            for (int x = 0; x < NS; ++x) o.p0[x] = stat[(ri.x >= 0 ? ri.x : 0) * 3 + x];  // (the same address for every lane)
#ifdef BLK_EXP_NO_LOAD
            blk_load_dyn<NS>(a.pool, o.p0);
#else
            blk_load_dyn<NS>((ri.x >= 0 && ri.x != j - 1) ? pool + (int64_t)ri.x * 384 : a.pool, o.p0);
#endif
        };
        double r[6] = {0, 0, 0, 0, 0, 0};  // the tuple of the node before: in post-order an internal right child (or an internal left child beside a
                                            // leaf) is the node just formed -- it does not come back from the pool
        for (int w0 = 0; w0 < ns; w0 += BLK_WIN) {
            const int wn = ns - w0 < BLK_WIN ? ns - w0 : BLK_WIN;
            __builtin_amdgcn_wave_barrier();  // (the window's last readers)
            for (int k = lane; k < wn; k += WAVE) {
                w_ri[k] = a.rec_i[rb + w0 + k]; w_re[k] = a.rec_e[rb + w0 + k];
                if (BME) w_rc[k] = a.rec_c[rb + w0 + k];
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            // the query-independent components of the blocks' roots in this window (the host's values, the same bits the walk forms:
            // build_blocks): stored here, outside the steps, whose stores stay a fixed number (the counted waits) -- until round 5's
            // end every step stored them, into a dummy row for the nodes that are no roots: a third of the kernel's stores
            for (int k = 0; k < wn; ++k) {
                if (!(w_ri[k].w & BLK_F_ROOT)) continue;  // (wave-uniform)
#pragma unroll
                for (int x = 0; x < NS; ++x) pool[(int64_t)(w0 + k) * 384 + x * 64] = stat[(w0 + k) * 3 + x];
            }
            // two operand sets in turn (a step uses one and requests the next node's into the other): copying a set would wait for
            // the loads it had just requested -- the first form did, and a step cost a memory round trip again
            auto step = [&](int k, const BlkOps &o, BlkOps &o1) __attribute__((always_inline)) {
                const int j = w0 + k;
                const int4 ri = w_ri[k];
                const double2 re = w_re[k];
                fetch(w_ri[k + 1 < wn ? k + 1 : k], j + 1, o1);  // the next node's operands (its slots lie below j: written)
                double S0[6], S1[6], L0[6], L1[6], u[6];
                leaf_tuple<M>(o.d0, L0);
                leaf_tuple<M>(o.d1, L1);
                const bool leaf0 = ri.x < 0, leaf1 = ri.y < 0, prev0 = ri.x == j - 1;
#pragma unroll
                for (int x = 0; x < 6; ++x) {
                    S0[x] = leaf0 ? L0[x] : (prev0 ? r[x] : o.p0[x]);
                    S1[x] = leaf1 ? L1[x] : r[x];  // (an internal right child is the node before)
                }
                // apples/BME.py:20: every child is valid here: 1 / #children each (1 / 2 on a binary node; a chain record of a node of
                // m children: 1 / m, and 1 for the chain's partial sum on the left: api.hip:build_blocks)
                double2 coef = make_double2(1.0, 1.0);
                if (BME) coef = w_rc[k];
                lift<M>(S0, re.x, u);
#pragma unroll
                for (int x = 0; x < 6; ++x) r[x] = 0;
#pragma unroll
                for (int x = 0; x < 6; ++x) r[x] += BME ? coef.x * u[x] : u[x];
                lift<M>(S1, re.y, u);
#pragma unroll
                for (int x = 0; x < 6; ++x) r[x] += BME ? coef.y * u[x] : u[x];
#ifdef BLK_EXP_NO_STORE  // (timing experiments: scripts/r05_blk_parts_exp.sh)
                if (r[0] == 123456.789) blk_store(pool + (int64_t)j * 384, r);
#else
                // the query's components of the tuple (the others: the window loop below stores them for a block's root, where the
                // sweep above the blocks reads a whole tuple; nobody reads them anywhere else)
                blk_store_dyn<NS>(pool + (int64_t)j * 384, r);
#endif
            };
            BlkOps oa, ob;
            fetch(w_ri[0], w0, oa);
            int k = 0;
            for (; k + 2 <= wn; k += 2) {
                step(k, oa, ob);
                step(k + 1, ob, oa);
            }
            if (k < wn) step(k, oa, ob);
        }
    }
}

#ifndef BLK_DOWN_WAVES
#define BLK_DOWN_WAVES 2  // wavefronts per SIMD k_blocks_down is compiled for (186 registers unconstrained)
#endif
template <int M>
__global__ __launch_bounds__(APPLES_TPB, BLK_DOWN_WAVES) void k_blocks_down(BlockArgs a) {
    constexpr bool BME = (M == APPLES_BME);
    __shared__ double sh_pow[384 + 256];
    __shared__ BlkWin win;
    for (int i = threadIdx.x; i < 384; i += APPLES_TPB) sh_pow[i] = (&kPowLogTab[0][0])[i];
    for (int i = threadIdx.x; i < 256; i += APPLES_TPB) sh_pow[384 + i] = __longlong_as_double((long long)kExpTab[i]);
    __syncthreads();
    const int lane = threadIdx.x & (WAVE - 1), wv = threadIdx.x / WAVE;
    int4 *w_ri = win.ri[wv];
    double2 *w_re = win.re[wv];
    const int n_tiles = *a.n_tiles;
    while (true) {
        int tq = 0;
        if (lane == 0) tq = atomicAdd(a.cursor, 1);
        const int ti = __builtin_amdgcn_readfirstlane(tq);
        if (ti >= n_tiles) break;
        const int4 tile = a.tiles[ti];
        if (tile.w < 0) continue;
        const int c = tile.x, nqt = tile.z;
        const bool in = lane < nqt;
        const int item = tile.y + (in ? lane : 0);
        const int2 it = a.items[item];
        const int64_t q = it.x;
        // (a lane whose query goes without blocks -- a member it drops, its own row, a top-up or exact-match query -- computes on
        // whatever the pool holds and writes nothing that is read)
        const bool mine = in && a.item_sbase[item] >= 0 && !a.item_bad[item] && a.q_blk[q] == 1;
        if (__ballot(mine) == 0ull) continue;
        const int rb = a.rep_soff[c], ns = a.rep_soff[c + 1] - rb;
        double *pool = a.pool + ((int64_t)tile.w + 1) * 384 + lane;
        const double *dT = pool + (int64_t)ns * 384;  // the members' distances as [member][lane] (k_blocks_up)
        // The lane's running best starts from what the sweep above the blocks found for the query (k_blocks_finish keeps the smaller
        // of the two anyway: smallest key, ties to the smaller edge index), so that an edge's residual with the reference's bits --
        // two libm squares, a third of this kernel's arithmetic -- is formed only where it could matter: MLSE takes the residual
        // with plain squares first, and unless that minus its proven slack (sweep_math.h:edge_residual) lies above the lane's best
        // the edge cannot win; ME needs the residual of an edge only when its x_1 wins.  The branch is the wavefront's: one lane
        // that needs the exact value sends all 64 through it (measured with plain squares everywhere, results then differ in the last
        // place: 2.42 -> 1.66 ms per launch, profiles/r05_plain_sq_exp.txt).
        LeanBest best;
        lean_best_init(best);
        bool found = false;
        if (mine) {
            const apples_placement up = a.out[q];
            if (up.edge >= 0) { best.key = a.criterion == APPLES_ME ? up.pendant : up.error; best.v = up.edge; }
        }
        const double coef = BME ? 1.0 / (double)(1 + 2 - 1) : 1.0;  // apples/BME.py:36-37: the node is not the LCA, one valid sibling
        constexpr int NS = BlkStat<M>::NS;
        const double *stat = a.stat + (int64_t)rb * 3;
        auto fetch = [&](const int4 &ri, int j, bool chained, BlkOps &o) __attribute__((always_inline)) {
            o.d0 = dT[(int64_t)(ri.x < 0 ? -ri.x - 1 : 0) * 64];
            o.d1 = dT[(int64_t)(ri.y < 0 ? -ri.y - 1 : 0) * 64];
            blk_load(chained ? a.pool : pool + (int64_t)j * 384, o.pl);  // (what a node does not need: one place for every lane, a sector)
#pragma unroll
            for (int x = 0; x < NS; ++x) {  // the children's query-independent components (the same address for every lane)
                o.p0[x] = stat[(ri.x >= 0 ? ri.x : 0) * 3 + x];
                o.p1[x] = stat[(ri.y >= 0 ? ri.y : 0) * 3 + x];
            }
            blk_load_dyn<NS>(ri.x >= 0 ? pool + (int64_t)ri.x * 384 : a.pool, o.p0);
            blk_load_dyn<NS>(ri.y >= 0 ? pool + (int64_t)ri.y * 384 : a.pool, o.p1);
        };
        double nxt[6] = {0, 0, 0, 0, 0, 0};  // lift(R) of node j - 1 where this step forms it (j - 1 is then a child of j): it does not go through the pool
        bool have = false;                  // (wave-uniform)
        for (int w1 = ns; w1 > 0; w1 -= BLK_WIN) {  // windows from the top: nodes [w0, w1)
            const int w0 = w1 > BLK_WIN ? w1 - BLK_WIN : 0, wn = w1 - w0;
            __builtin_amdgcn_wave_barrier();
            for (int k = lane; k < wn; k += WAVE) { w_ri[k] = a.rec_i[rb + w0 + k]; w_re[k] = a.rec_e[rb + w0 + k]; }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            auto step = [&](int k, const BlkOps &o, BlkOps &o1) __attribute__((always_inline)) {
                const int j = w0 + k;
                const int4 ri = w_ri[k];
                const double2 re = w_re[k];
                // the next node's operands: its own lift(R) (unless this step forms it: then what comes is not used), its children's
                // tuples (slots below j - 1: this step writes none of them -- an internal child of j that is not j - 1 roots another subtree)
                // (round 6: a chain's inner record -- a partial sum, not a node -- does nothing on the way down; the chain's last record,
                // a node of more than two children, forms every child's R itself, below: of its children only the last can be j - 1)
                const bool part = (ri.w & BLK_F_PART) != 0, poly = (ri.w & BLK_F_POLY) != 0;  // (wave-uniform)
                const bool forms_next = !part && (poly ? ri.y == j - 1 : (ri.x == j - 1 || ri.y == j - 1));
                fetch(w_ri[k > 0 ? k - 1 : 0], k > 0 ? j - 1 : j, forms_next, o1);
                if (part) { have = false; return; }
                double plift[6], S0[6], S1[6], L0[6], L1[6];
                leaf_tuple<M>(o.d0, L0);
                leaf_tuple<M>(o.d1, L1);
                const bool leaf0 = ri.x < 0, leaf1 = ri.y < 0;
#pragma unroll
                for (int x = 0; x < 6; ++x) {
                    plift[x] = have ? nxt[x] : o.pl[x];  // from its parent's step (a root: from the sweep above)
                    S0[x] = leaf0 ? L0[x] : o.p0[x];
                    S1[x] = leaf1 ? L1[x] : o.p1[x];
                }
                // the edge above one child from its R (`acc`): solve, the lazy residual, lift(R) to an internal child
                // (defer: an internal child's lift(R) goes there instead of its slot -- the polytomy step, whose children's S tuples must
                // stay in the pool until every sibling has read them)
                auto kid_acc = [&](const double *Sk, const double *acc, double ek, int kd, int kn, double *defer) __attribute__((always_inline)) {
                    double u[6];
                    Sol r = solve_x<M>(Sk, acc, ek, a.negative);
                    bool need;
                    if (a.criterion == APPLES_ME) {
                        need = mine && (r.x1 < best.key || (r.x1 == best.key && kn < best.v));
                    } else {
                        double mag;
                        const double lower = edge_residual<M, false>(Sk, acc, ek, r.x1, r.x2, nullptr, &mag);
                        need = mine && !(lower - 0x1p-40 * mag > best.key);  // (a NaN anywhere: the exact evaluation decides)
                    }
                    if (kd >= 0) {  // an internal child: lift(R) over its edge replaces its S (both S tuples are in registers)
                        lift<M>(acc, ek, u);
                        if (defer) {
#pragma unroll
                            for (int x = 0; x < 6; ++x) defer[x] = u[x];
                        } else if (kd == j - 1) {  // the next node of the walk
#pragma unroll
                            for (int x = 0; x < 6; ++x) nxt[x] = u[x];
                        } else blk_store(pool + (int64_t)kd * 384, u);
                    }
                    if (__ballot(need) != 0ull) {
                        r.err = edge_residual<M, true>(Sk, acc, ek, r.x1, r.x2, sh_pow);
                        const double key = (a.criterion == APPLES_ME) ? r.x1 : r.err;
                        if (need && (key < best.key || (key == best.key && kn < best.v))) {
                            best.key = key; best.v = kn; best.x1 = r.x1; best.x2 = r.x2; best.err = r.err; best.x1_int = r.x1_int; best.e = ek;
                            found = true;
                        }
                    }
                };
                auto kid = [&](const double *Sk, const double *Ss, double ek, double es, int kd, int kn) __attribute__((always_inline)) {
                    double acc[6], u[6];
#pragma unroll
                    for (int x = 0; x < 6; ++x) acc[x] = 0;
                    lift<M>(Ss, es, u);  // the one valid sibling (apples/OLS.py:59-69)
#pragma unroll
                    for (int x = 0; x < 6; ++x) acc[x] += BME ? coef * u[x] : u[x];
#pragma unroll
                    for (int x = 0; x < 6; ++x) acc[x] += BME ? coef * plift[x] : plift[x];  // parent term last (apples/OLS.py:70-80)
                    kid_acc(Sk, acc, ek, kd, kn, nullptr);
                };
                if (poly) {
                    // a node of m > 2 children (all of them observed, the node not the LCA): child i's R = every other child's lifted S in
                    // file order, then the node's own lifted R (apples/OLS.py:59-80; BME: all times 1 / m, apples/BME.py:36-38).  Rare:
                    // the children's tuples are fetched here, one at a time, again for every sibling (m <= BLK_MAX_DEG)
                    const int2 pp = a.rec_p[rb + j];
                    const double cf = BME ? 1.0 / (double)pp.y : 1.0;
                    auto child = [&](int c, double *S) __attribute__((always_inline)) {
                        const int ref = a.pk_i[pp.x + c].x;
                        if (ref >= 0) {
#pragma unroll
                            for (int x = 0; x < NS; ++x) S[x] = stat[ref * 3 + x];
                            blk_load_dyn<NS>(pool + (int64_t)ref * 384, S);
                        } else {
                            leaf_tuple<M>(dT[(int64_t)(-ref - 1) * 64], S);
                        }
                    };
                    // (the children's lift(R) wait in registers until every child has read its siblings' S tuples from the pool)
                    double U[BLK_MAX_DEG][6];
                    int KD[BLK_MAX_DEG];
#pragma unroll
                    for (int i = 0; i < BLK_MAX_DEG; ++i) {
                        KD[i] = -1;
                        if (i < pp.y) {  // (wave-uniform)
                            double acc[6] = {0, 0, 0, 0, 0, 0}, Sk[6], u[6];
#pragma unroll 1
                            for (int c = 0; c < pp.y; ++c) {
                                if (c == i) continue;
                                child(c, Sk);
                                lift<M>(Sk, a.pk_e[pp.x + c], u);
#pragma unroll
                                for (int x = 0; x < 6; ++x) acc[x] += BME ? cf * u[x] : u[x];
                            }
#pragma unroll
                            for (int x = 0; x < 6; ++x) acc[x] += BME ? cf * plift[x] : plift[x];
                            child(i, Sk);
                            const int2 ki = a.pk_i[pp.x + i];
                            KD[i] = ki.x;
                            kid_acc(Sk, acc, a.pk_e[pp.x + i], ki.x, ki.y, U[i]);
                        }
                    }
#pragma unroll
                    for (int i = 0; i < BLK_MAX_DEG; ++i) {
                        if (KD[i] >= 0) {  // (wave-uniform)
                            if (KD[i] == j - 1) {
#pragma unroll
                                for (int x = 0; x < 6; ++x) nxt[x] = U[i][x];
                            } else blk_store(pool + (int64_t)KD[i] * 384, U[i]);
                        }
                    }
                    have = forms_next;
                    return;
                }
                kid(S0, S1, re.x, re.y, ri.x, ri.z);
                kid(S1, S0, re.y, re.x, ri.y, ri.w & BLK_NODE_MASK);  // (the bits above: BLK_F_*)
                have = forms_next;
            };
            BlkOps oa, ob;
            fetch(w_ri[wn - 1], w1 - 1, have, oa);
            int k = wn - 1;
            for (; k >= 1; k -= 2) {
                step(k, oa, ob);
                step(k - 1, ob, oa);
            }
            if (k == 0) step(0, oa, ob);
        }
        if (in) {  // slot 0 of the tile: key, x1, x2, err, e, (x1 is the int 0, edge)
            double *b = a.pool + (int64_t)tile.w * 384 + lane;
            b[0] = (mine && found) ? best.key : INF_D; b[64] = best.x1; b[128] = best.x2; b[192] = best.err; b[256] = best.e;
            b[320] = __hiloint2double(best.x1_int, best.v);
        }
    }
}

// the query's placement = the better of what the sweep above the blocks found and the best edge inside its blocks
// (apples/Algorithm.py:74-101: smallest key, ties to the smaller edge_index).  A wavefront per query.
__global__ __launch_bounds__(APPLES_TPB) void k_blocks_finish(BlockArgs a) {
    const int lane = threadIdx.x & (WAVE - 1);
    const int64_t q = (int64_t)blockIdx.x * (APPLES_TPB / WAVE) + threadIdx.x / WAVE;
    if (q >= a.nq || a.q_blk[q] != 1) return;
    const int2 qi = a.q_items[q];
    double key = INF_D;
    int v = 0x7fffffff, at = -1;
    for (int k = lane; k < qi.y; k += WAVE) {
        const int item = a.q_item[qi.x + k], sb = a.item_sbase[item];
        if (sb < 0 || a.item_bad[item]) continue;
        const double *b = a.pool + (int64_t)(sb >> 6) * 384 + (sb & 63);
        const double k2 = b[0];
        const int v2 = __double2loint(b[320]);
        if (k2 < key || (k2 == key && v2 < v)) { key = k2; v = v2; at = sb; }
    }
    const int my_v = v;
    double wkey = key;
    int win = v;
    team_argmin<WAVE>(wkey, win, nullptr, nullptr);
    if (win == 0x7fffffff || my_v != win || at < 0) return;  // (edge ids are distinct: one lane holds the winner)
    apples_placement pl = a.out[q];
    const double ukey = pl.edge < 0 ? INF_D : (a.criterion == APPLES_ME ? pl.pendant : pl.error);
    if (!(wkey < ukey || (wkey == ukey && win < pl.edge))) return;
    const double *b = a.pool + (int64_t)(at >> 6) * 384 + (at & 63);
    const double x1 = b[64], x2 = b[128], err = b[192], e = b[256];
    pl.edge = win;
    pl.error = err;
    pl.distal = e - x2;
    pl.pendant = x1;
    pl.flags = 0;
    if (__double2hiint(b[320])) pl.flags |= APPLES_F_PENDANT_INT;
    if (x1 == 0 && err > 0 && (x2 == 0 || x2 == e)) pl.flags |= APPLES_F_MISPLACED;
    a.out[q] = pl;
}

}  // namespace

// clade blocks: S tuples of the blocks of every (query, accepted cluster) item, before the selection's last phase names the
// block roots in the observation lists
int launch_blocks_up(apples_ctx *ctx, const BlockArgs &a, hipStream_t st) {
    const int cus = ctx->n_cu > 0 ? ctx->n_cu : 256;
    // tuning knob (82 registers: up to five or six wavefronts per SIMD -- but the kernel is bound by its pool traffic, not by the
    // wavefronts in flight: 6 / 4 / 3 / 2 workgroups per CU give 34.6 / 34.7 / 34.3 / 33.7 ms on config 3's clustered pass,
    // profiles/r05_blk_order_exp.txt, and the fewer there are the more room the selection's last phase has beside them)
    // then one per CU (32.6 against 33.5 ms at two; a grid of 64 / 128 / 192 workgroups: 46.2 / 35.7 / 32.4 ms)
    const int per_cu = (int)knob(ctx, "APPLES_BLK_UP_WGS", 1);
    const dim3 grid((unsigned)(cus * std::max(per_cu, 1))), block(APPLES_TPB);
    HIP_TRY(ctx, hipMemsetAsync(a.cursor, 0, sizeof(int32_t), st));
    switch (a.method) {
        case APPLES_FM: hipLaunchKernelGGL((k_blocks_up<APPLES_FM>), grid, block, 0, st, a); break;
        case APPLES_BME: hipLaunchKernelGGL((k_blocks_up<APPLES_BME>), grid, block, 0, st, a); break;
        case APPLES_BE: hipLaunchKernelGGL((k_blocks_up<APPLES_BE>), grid, block, 0, st, a); break;
        default: hipLaunchKernelGGL((k_blocks_up<APPLES_OLS>), grid, block, 0, st, a); break;
    }
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}

// ... and after the sweep above the blocks: the top-down pass inside them, then the queries' placements
int launch_blocks_down(apples_ctx *ctx, const BlockArgs &a, hipStream_t st) {
    const int cus = ctx->n_cu > 0 ? ctx->n_cu : 256;
    // (193 registers: two wavefronts per SIMD; one workgroup per CU instead of two: the sweep phase 15.2 -> 18.7 ms; compiled for
    // three with 23 registers spilled: 16.5, profiles/r05_blk_order_exp.txt)
    const dim3 grid((unsigned)(cus * BLK_DOWN_WAVES)), block(APPLES_TPB);
    HIP_TRY(ctx, hipMemsetAsync(a.cursor, 0, sizeof(int32_t), st));
    switch (a.method) {
        case APPLES_FM: hipLaunchKernelGGL((k_blocks_down<APPLES_FM>), grid, block, 0, st, a); break;
        case APPLES_BME: hipLaunchKernelGGL((k_blocks_down<APPLES_BME>), grid, block, 0, st, a); break;
        case APPLES_BE: hipLaunchKernelGGL((k_blocks_down<APPLES_BE>), grid, block, 0, st, a); break;
        default: hipLaunchKernelGGL((k_blocks_down<APPLES_OLS>), grid, block, 0, st, a); break;
    }
    hipLaunchKernelGGL(k_blocks_finish, dim3((unsigned)((a.nq + APPLES_TPB / WAVE - 1) / (APPLES_TPB / WAVE))), block, 0, st, a);
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}

// workgroup-sized teams over a device-side list (routed or overflow queries); a.lean = the big teams' field arrays
int launch_sweep_lean_big(apples_ctx *ctx, const SweepArgs &a, int64_t nq, int wgs, hipStream_t st) {
    if (nq == 0) return 0;
    // Team size: 256 threads, and 512 in a small device batch (a shard of a multi-GPU job, a -d block: the batches route_threshold
    // lowers the routing cut for).  There the routed queries are a launch of their own length -- config 3's 12 500-query shards
    // with a 34 000-leaf query: this kernel 2.74 -> 1.61 ms, the sweep 2.75 -> 2.24, the shard 7.65 -> 7.05 ms; a shard without such
    // a query 6.86 -> 6.91 (profiles/r04_shard_bigteam_exp.txt) -- while in the big batches of a full pass the wavefront-sized teams'
    // kernels beside it are as long as it is either way (no change at config 3, 35.8 -> 37.2 ms on the clustered route: half as many
    // teams per compute unit).  512-thread teams for the few largest queries only, in a launch of their own beside a launch of
    // 256-thread teams, measured worse than either (2.86 / 2.57 ms on the two shards).  APPLES_LEAN_BIG_TEAM: tuning knob.
    const int team_env = (int)knob(ctx, "APPLES_LEAN_BIG_TEAM", 0);
    const int team = team_env > 0 ? team_env : (ctx->cur_batch_queries > 0 && ctx->cur_batch_queries <= LEAN_SMALL_BATCH ? 512 : 256);
    const dim3 grid((unsigned)std::min<int64_t>(nq, wgs));
    if (team == 512) launch_lean_big_t<512>(a, nq, grid, st);
    else launch_lean_big_t<256>(a, nq, grid, st);
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}

// wavefront-sized teams over the size-class queues of one device batch: the bottom-up kernel, then (unless the timing
// knob says bottom-up only) the top-down kernel, each a persistent grid of the workgroups its registers let a CU hold
// (x 1.5: a second round shortens the tail).  `up` / `down` differ in their work cursor only.
int launch_sweep_lean(apples_ctx *ctx, const SweepArgs &up, const SweepArgs &down, int64_t nq, hipStream_t st, int32_t *halves,
                      hipStream_t side, hipEvent_t *ev) {
    if (nq == 0) return 0;
    const int cus = ctx->n_cu > 0 ? ctx->n_cu : 256;
    const int wg_up = (int)knob(ctx, "APPLES_LEAN_UP_WGS", 0);      // tuning knobs
    const int wg_down = (int)knob(ctx, "APPLES_LEAN_DOWN_WGS", 0);
    const int64_t need = (nq + 3) / 4;
    const dim3 block(APPLES_TPB);
    const bool pl_up = up.tree.max_children > 2 || up.tree.force_poly == 1 || up.tree.force_poly == 2;  // (k_lean_up<M, true>)
    const dim3 gu((unsigned)std::min<int64_t>(need, wg_up > 0 ? wg_up : cus * (pl_up ? LEAN_UP_WAVES_PL : LEAN_UP_WAVES) * 3 / 2));
    const dim3 gd((unsigned)std::min<int64_t>(need, wg_down > 0 ? wg_down : cus * LEAN_DOWN_WAVES * 3 / 2));
    if ((int64_t)gu.x * 4 > up.lean_teams) { ctx->err = "lean sweep: more bottom-up teams than per-leaf scratch"; return 1; }
    // APPLES_LEAN_FUSED=1: one kernel for both passes of a query (k_lean_both), the team's top-down pass right behind its bottom-up
    // pass.  Measured SLOWER than the two kernels in every workload (round 5, profiles/r05_lean_fused_exp.txt: config 3's sweep 15.6 -> 16.7 ms,
    // config 5's block 1.31 -> 1.44, config 4 2.77 -> 3.05): at three wavefronts per SIMD the fused body spills 85 vector registers
    // where the bottom-up kernel alone spills 19 and the top-down kernel 8, and the launch it saves is worth less than that.
    const bool fused = knob_on(ctx, "APPLES_LEAN_FUSED");  // experiment knob
    if (fused && !up.prof && up.debug_phase != 1 && up.criterion != APPLES_HYBRID && up.tree.max_children <= 2) {
        switch (up.method) {
            case APPLES_FM: hipLaunchKernelGGL((k_lean_both<APPLES_FM>), gu, block, 0, st, up); break;
            case APPLES_BME: hipLaunchKernelGGL((k_lean_both<APPLES_BME>), gu, block, 0, st, up); break;
            case APPLES_BE: hipLaunchKernelGGL((k_lean_both<APPLES_BE>), gu, block, 0, st, up); break;
            default: hipLaunchKernelGGL((k_lean_both<APPLES_OLS>), gu, block, 0, st, up); break;
        }
        HIP_TRY(ctx, hipGetLastError());
        return 0;
    }
    // (the kernels for trees with polytomies; the knob: timing experiments on binary trees -- 1: both kernels, 2: bottom-up only, 3: top-down only)
    const int force = (int)knob(ctx, "APPLES_LEAN_FORCE_POLY", 0);
    const bool poly_up = up.tree.max_children > 2 || force == 1 || force == 2, poly = up.tree.max_children > 2 || force == 1 || force == 3;
    auto launch_up = [&](const SweepArgs &x, hipStream_t s_) {
        if (poly_up) {
            switch (x.method) {
                case APPLES_FM: hipLaunchKernelGGL((k_lean_up<APPLES_FM, true>), gu, block, 0, s_, x); break;
                case APPLES_BME: hipLaunchKernelGGL((k_lean_up<APPLES_BME, true>), gu, block, 0, s_, x); break;
                case APPLES_BE: hipLaunchKernelGGL((k_lean_up<APPLES_BE, true>), gu, block, 0, s_, x); break;
                default: hipLaunchKernelGGL((k_lean_up<APPLES_OLS, true>), gu, block, 0, s_, x); break;
            }
            return;
        }
        switch (x.method) {
            case APPLES_FM: hipLaunchKernelGGL((k_lean_up<APPLES_FM>), gu, block, 0, s_, x); break;
            case APPLES_BME: hipLaunchKernelGGL((k_lean_up<APPLES_BME>), gu, block, 0, s_, x); break;
            case APPLES_BE: hipLaunchKernelGGL((k_lean_up<APPLES_BE>), gu, block, 0, s_, x); break;
            default: hipLaunchKernelGGL((k_lean_up<APPLES_OLS>), gu, block, 0, s_, x); break;
        }
    };
    auto launch_down = [&](const SweepArgs &x, hipStream_t s_) {
        if (poly) {
            if (x.criterion == APPLES_HYBRID) {
                switch (x.method) {
                    case APPLES_FM: hipLaunchKernelGGL((k_lean_down<APPLES_FM, true, true>), gd, block, 0, s_, x); break;
                    case APPLES_BME: hipLaunchKernelGGL((k_lean_down<APPLES_BME, true, true>), gd, block, 0, s_, x); break;
                    case APPLES_BE: hipLaunchKernelGGL((k_lean_down<APPLES_BE, true, true>), gd, block, 0, s_, x); break;
                    default: hipLaunchKernelGGL((k_lean_down<APPLES_OLS, true, true>), gd, block, 0, s_, x); break;
                }
            } else {
                switch (x.method) {
                    case APPLES_FM: hipLaunchKernelGGL((k_lean_down<APPLES_FM, false, true>), gd, block, 0, s_, x); break;
                    case APPLES_BME: hipLaunchKernelGGL((k_lean_down<APPLES_BME, false, true>), gd, block, 0, s_, x); break;
                    case APPLES_BE: hipLaunchKernelGGL((k_lean_down<APPLES_BE, false, true>), gd, block, 0, s_, x); break;
                    default: hipLaunchKernelGGL((k_lean_down<APPLES_OLS, false, true>), gd, block, 0, s_, x); break;
                }
            }
            return;
        }
        if (x.criterion == APPLES_HYBRID) {  // (per-edge records in the entries' dead tuple slots + lean_hybrid_pick)
            switch (x.method) {
                case APPLES_FM: hipLaunchKernelGGL((k_lean_down<APPLES_FM, true>), gd, block, 0, s_, x); break;
                case APPLES_BME: hipLaunchKernelGGL((k_lean_down<APPLES_BME, true>), gd, block, 0, s_, x); break;
                case APPLES_BE: hipLaunchKernelGGL((k_lean_down<APPLES_BE, true>), gd, block, 0, s_, x); break;
                default: hipLaunchKernelGGL((k_lean_down<APPLES_OLS, true>), gd, block, 0, s_, x); break;
            }
            return;
        }
        switch (x.method) {
            case APPLES_FM: hipLaunchKernelGGL((k_lean_down<APPLES_FM>), gd, block, 0, s_, x); break;
            case APPLES_BME: hipLaunchKernelGGL((k_lean_down<APPLES_BME>), gd, block, 0, s_, x); break;
            case APPLES_BE: hipLaunchKernelGGL((k_lean_down<APPLES_BE>), gd, block, 0, s_, x); break;
            default: hipLaunchKernelGGL((k_lean_down<APPLES_OLS>), gd, block, 0, s_, x); break;
        }
    };
    // A small device batch (a shard of a multi-GPU job, a -d block) leaves most of the chip idle during either kernel -- a team has
    // three or four queries and a launch lasts as long as its longest: the queue is then taken in two halves (entries of even and
    // of odd index: both get their share of every size class), and the top-down kernel of the first half runs on `side` beside
    // the bottom-up kernel of the second.  halves = the two extra work cursors (cleared with the batch's counters), or nullptr.
    if (halves && side && up.debug_phase != 1 && !up.prof) {
        SweepArgs u0 = up, u1 = up, d0 = down, d1 = down;
        u0.w_mod = u1.w_mod = d0.w_mod = d1.w_mod = 2;
        u1.w_rem = d1.w_rem = 1;
        u1.cursor = halves; d1.cursor = halves + 1;
        // (the second bottom-up launch needs per-leaf scratch of its own: the teams of the two launches are alive together)
        // (the THIRD set of the workspace: the second belongs to the overlapped top-up chain's sweep, api.hip:run_sweep_second)
        u1.lean_leaf = reinterpret_cast<char *>(up.lean_leaf) + 2 * up.lean_teams * up.lean_leaf1 * LEAN_BYTES_PER_LEAF;
        launch_up(u0, st);
        HIP_TRY(ctx, hipEventRecord(ev[0], st));
        HIP_TRY(ctx, hipStreamWaitEvent(side, ev[0], 0));
        launch_down(d0, side);
        HIP_TRY(ctx, hipEventRecord(ev[1], side));
        launch_up(u1, st);
        launch_down(d1, st);
        HIP_TRY(ctx, hipStreamWaitEvent(st, ev[1], 0));
        HIP_TRY(ctx, hipGetLastError());
        return 0;
    }
    launch_up(up, st);
    if (up.debug_phase != 1) launch_down(down, st);
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}

// bottom-up teams the workspace must hold per-leaf scratch for
int sweep_lean_up_teams(const apples_ctx *ctx) {
    const int cus = ctx->n_cu > 0 ? ctx->n_cu : 256;
    const int wg_up = (int)knob(ctx, "APPLES_LEAN_UP_WGS", 0);
    return 4 * (wg_up > 0 ? wg_up : cus * LEAN_UP_WAVES * 3 / 2);
}
