// Shared declarations of libapples_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/apples_hip.h"

#define APPLES_TPB 256  // threads per workgroup in every kernel: 4 wave64

// Tree constants of one node in one 32-byte record (one memory transaction per visit).
// Tree constants of a node and what a parent needs to know about its first two children, in one
// 64-byte line: the sweep's level steps are chains of dependent loads, and a separate load of the
// children's records would add a link to every step.
struct __attribute__((aligned(64))) NodeRec {
    // 16-byte chunks grouped by reader, so that each pass of the sweep touches as few as possible:
    // top-down reads chunks 0-1, bottom-up chunks 0, 2 and the first word of 3
    double e0, e1;        // the first two children's edge lengths
    double e;             // edge length
    int32_t c0, c1;       // first two children in file order (-1 if absent)
    int32_t nchild;
    uint32_t kleaf;       // bit 0: c0 is a leaf, bit 1: c1 is a leaf
    int32_t c0pos, c1pos; // the children's positions in the level-ordered bit space (sweep.hip NodeBits)
    int32_t ppos;         // the parent's position (-1 for the root)
    int32_t parent;       // -1 for the root
    int32_t lpos;         // own position
    int32_t node;         // own id (what a reader of the position-ordered copy needs)
};

// one ancestor of a leaf (scan formulation of the sweep): edge length and node id
struct __attribute__((aligned(16))) AncRec {
    double e;
    int32_t node;
    int32_t pad;
};

// residue -> BLOSUM45 row/column (apples/distance.py:418-678: a2i, both cases; anything else counts as 'A'), 20 = gap
__device__ __forceinline__ uint8_t aa_index(uint8_t b) {
    switch (b) {
        case 'A': case 'a': return 0;
        case 'R': case 'r': return 1;
        case 'N': case 'n': return 2;
        case 'D': case 'd': return 3;
        case 'C': case 'c': return 4;
        case 'Q': case 'q': return 5;
        case 'E': case 'e': return 6;
        case 'G': case 'g': return 7;
        case 'H': case 'h': return 8;
        case 'I': case 'i': return 9;
        case 'L': case 'l': return 10;
        case 'K': case 'k': return 11;
        case 'M': case 'm': return 12;
        case 'F': case 'f': return 13;
        case 'P': case 'p': return 14;
        case 'S': case 's': return 15;
        case 'T': case 't': return 16;
        case 'W': case 'w': return 17;
        case 'Y': case 'y': return 18;
        case 'V': case 'v': return 19;
        case '-': return 20;
        default: return 0;
    }
}

struct DevTree {
    int32_t n_nodes = 0;
    int32_t height = 0;  // max level
    int32_t max_children = 0;
    int32_t force_poly = 0;  // experiment knob APPLES_LEAN_FORCE_POLY: the polytomy-capable kernels on a binary tree (1 both, 2 bottom-up, 3 top-down)
    int64_t poly_kids = 0;  // children of the nodes with more than two (sweep_lean.hip: the child records a full-size team may need)
    // the merged level lists (sweep.hip:merge_parents, sweep_lean.hip) need node ids in left-to-right post-order and observed
    // leaves that are distinct nodes; a tree or an alignment / table that does not comply gets the node map or the node bits
    bool merge_ok = false;
    bool lean_small = false;  // criterion other than HYBRID: a small tree, too, takes the merged lists + lean sweep (sweep.hip:sweep_merge_lists)
    uint32_t dbg = 0;  // APPLES_DBG_* switches of the context (apples_params.debug | environment), fixed at creation
    int32_t *parent = nullptr;
    double *edge_len = nullptr;
    int32_t *child_off = nullptr;
    int32_t *child_idx = nullptr;
    int32_t *level = nullptr;
    NodeRec *rec = nullptr;
    // level-ordered bit space of the sweep: per level one block for its internal nodes and one for
    // its leaves (each in node-id order, each starting on a 64-bit word)
    int32_t bm_words = 0;        // 64-bit words of the whole space
    int32_t *lvlw = nullptr;     // [2*(height+1)+1] first word of level l's internal block at 2l, leaf block at 2l+1
    int32_t *lnode = nullptr;    // [bm_words*64] node at a bit position (-1 = padding)
    NodeRec *rec_l = nullptr;    // [bm_words*64] the records in bit-position order: position -> record in one load
    int32_t *npos = nullptr;     // [n_nodes][2] lpos, ppos (what a leaf needs, without its 64-byte record)
    void *pe = nullptr;          // [n_nodes] {int32 parent, int32 -, double edge length}: the one tree gather of sweep_lean.hip
    // scan formulation of the sweep (sweep_scan.hip): per leaf, by level m = 0..level(leaf), the ancestor at
    // that level.  Built when the tree is shallow enough for the tables to fit (else the level-loop sweep runs).
    bool scan = false;
    int4 *leaf_info = nullptr;   // [n_nodes] {offset of the leaf's rows in `anc` (-1: not a leaf), level, Euler position, -}
    AncRec *anc = nullptr;       // [sum(level+1)] the ancestor's edge length and node id
    // level of the lowest common ancestor of two leaves = minimum of the Euler tour's levels between
    // their positions: sparse table of range minima, rmq[k][t] = min level over tour steps [t, t + 2^k)
    uint8_t *rmq = nullptr;      // [rmq_k][euler_len]
    int32_t euler_len = 0, rmq_k = 0;
};


// Slot = physical position of an alignment row on the device.  Member slots [0, n_refs) are the
// reference rows sorted by tree level, deepest first (rows that are not tree leaves last);
// consensus slots follow.  Sorting by level lets the selection kernel's ordered compaction
// hand the sweep kernel its leaves already grouped by level.
// threshold of the GEMM-form distance pass as a linear function of the valid count (dist_gemm.hip): 4 mism <=
// slope * (8192 valid) + off and 8192 valid >= vmin8 <=> mism <= mmax[valid], verified for every valid in [0, L]
struct GemmThreshold {
    float slope = 0.f, off = 0.f, vmin8 = 0.f;
    bool ok = false;
};
#define GEMM_MAX_L13 2046  // the merged accumulator with the validity sum at 2^13 decodes exactly while valid <= 2046 ...
#define GEMM_MAX_L 4092    // ... and at 2^12, by the congruence of dist_gemm.hip's header, while valid <= 4092

struct DevAlign {
    int64_t n_rows = 0, n_refs = 0, n_reps = 0;
    int64_t slots_pad = 0;  // n_rows rounded up to a multiple of APPLES_TPB
    int32_t L = 0, W = 0, G = 0;  // sites, 32-site words, groups of 4 words
    int32_t planes = 0;           // code planes in the packed layout: 2 (ACGT fast path) or 8 (raw byte)
    bool all_singleton = true;
    uint8_t *raw = nullptr;       // [n_rows*L] bytes in the caller's row order (kept for lazy repacking)
    int32_t *d_slot_row = nullptr;// [n_rows] slot -> caller's row (the packing kernels gather through it)
    uint4 *packed = nullptr;      // [G][planes+1][slots_pad] uint4 = 4 consecutive 32-site words
    uint8_t *ref_f4 = nullptr;    // [slots_pad][2G][4][32 B] fp4 operand image of the reference (dist_gemm.hip), ACGT- singleton contexts
    // Bytes beyond ACGT- (`.`, `?`, `*`: ordinary symbols to apples/distance.py:733-737) in a singleton JC69 context: the 2-plane
    // rows and the fp4 images hold them AS GAPS, so that the matrix-core passes stay what they are; exactness comes back through
    //   * packed8: the same rows once more in the 8-plane form (the raw byte), for everything that writes full rows (the listed /
    //     top-up pass, apples_distances, the unfused route) and for query blocks that carry such bytes themselves (QueryBlock::exact8);
    //   * ex_off / ex_site: per slot the sites that hold such a byte -- k_exotic_fix adds them back to the counts of the fused
    //     pass's survivors (a site where the row has such a byte and the query a letter is valid and a mismatch).
    bool ex_ok = false;           // the context takes this route (singleton clusters, JC69, matrix-core passes enabled, L < 8192)
    uint4 *packed8 = nullptr;     // [G][9][slots_pad]; built with the first such byte, in the reference or in a query block
    int32_t *ex_off = nullptr;    // [n_rows + 1] CSR over slots, or nullptr: the reference rows hold no such byte
    uint16_t *ex_site = nullptr;  // [ex_off[n_rows]] sites, ascending per slot
    int32_t ex_max = 0;           // most such sites in one row
    // clustered references (fused selection by representatives, select.hip k_select_clusters):
    uint4 *packed_rm = nullptr;   // the member rows cluster by cluster, interleaved per plane word (dist.hip:k_cluster_major): [n_refs * G*3]
    uint4 *rep_packed = nullptr;  // [G][3][reps_pad] the representatives' rows, in representative order
    int64_t reps_pad = 0;
    // ... and of scoredist contexts (the default route of -p): the representatives' rows in representative order, in the layout
    // of aa_idx / aa_mask with reps_pad in place of slots_pad (what k_scoredist runs over in place of every slot)
    uint8_t *aa_rep_idx = nullptr;   // [Lpad16/16][reps_pad][16]
    uint16_t *aa_rep_mask = nullptr; // [Lpad16/16][reps_pad]
    uint8_t *aa_cm_idx = nullptr;    // [Lpad16/16][slots_pad][16]: the member rows cluster by cluster (position rep_moff[c] + m: mem_slot's order), k_cluster_dist_sd
    uint16_t *aa_cm_mask = nullptr;  // [Lpad16/16][slots_pad]
    uint8_t *aa_idx = nullptr;    // scoredist: [Lpad16/16][slots_pad][16] residue index * 8 (0..152, 160 = gap)
    uint16_t *aa_mask = nullptr;  // scoredist: [Lpad16/16][slots_pad] bit k = site 16*s16+k is not a gap
    uint8_t *aa_rows = nullptr;   // scoredist, beside sd_ref4: the same residue bytes slot-major, [slots_pad][aa_Lrow] (dist_sd.hip:sd_eval64)
    uint16_t *aa_mrows = nullptr; // ... and the gap masks, [slots_pad][aa_Lrow / 16]
    int32_t aa_Lrow = 0;          // bytes per row of aa_rows: L rounded up to a multiple of 64
    uint8_t *sd_ref4 = nullptr;   // scoredist, singleton clusters: one-hot fp4 operand image of the reference rows (dist_sd.hip),
                                  // [slots_pad / 256][steps][1024 chunks of 16 B], 20 values per site
    float *sd_nvr = nullptr;      // [slots_pad] sites of the row that are not gaps (-1: no row in the slot)
    bool sd_fp6 = false;          // the query images hold fp6 table values (24 bytes per 32), not fp4 (APPLES_DBG_SD_FP6; default fp4)
    // Clade blocks of a clustered reference (sweep_lean.hip: k_blocks_up / k_blocks_down): a block = a whole subtree of the
    // backbone all of whose leaves are members of one cluster (at least two leaves, binary inside).  A query that accepts the
    // cluster observes every leaf of the block (apples/Reference.py:146-152 keeps every member with a valid distance), so the
    // induced subtree inside the block is the block itself: its S / R passes run on a static schedule, cluster-major, with the
    // queries in the lanes, and the per-query merged sweep sees the block's root as one leaf that carries a tuple.
    int32_t n_blocks = 0;
    int4 *blk_rec_i = nullptr;    // per internal node of the clusters' blocks, cluster by cluster, a cluster's blocks by root id, a block's
                                  // internal nodes in post-order ("slot" = index inside the cluster): {left, right, node id of left, of right};
                                  // left / right >= 0: the child's slot, < 0: a leaf, member position -(x + 1) of the cluster's flat member list
    double2 *blk_rec_e = nullptr; // ... and the edge lengths of the two children
    // (round 6) a node of up to BLK_MAX_DEG children inside a block = a chain of records (api.hip:build_blocks): the flags in the
    // fourth component of blk_rec_i (BLK_F_*), BME's coefficients of a record's two operands, and for the chain's last record -- the
    // node itself -- all its children once more for the top-down walk
    double2 *blk_rec_c = nullptr; // [record] BME: coefficient of the left / right operand (1 / #children; 1 for the chain's partial sum)
    int2 *blk_rec_p = nullptr;    // [record] (first entry of blk_pk_*, children) where BLK_F_POLY is set
    int2 *blk_pk_i = nullptr;     // a polytomy's children in file order: (slot, or -(member position + 1) for a leaf; node id)
    double *blk_pk_e = nullptr;   // ... their edge lengths
    double *blk_stat[2] = {nullptr, nullptr};  // ... and [record][3] the first three components of the node's S tuple, which do not depend on the
                                  // query inside a block (api.hip:build_blocks): [0] OLS / BE / FM, [1] BME
    int32_t *rep_soff = nullptr;  // [n_reps + 1] first record of every cluster (records = internal nodes of its blocks)
    int32_t *cl_order = nullptr;  // [n_reps] the clusters by falling number of records
    int32_t *rep_boff = nullptr, *rep_loff = nullptr, *loose_mp = nullptr;  // [n_reps + 1] the clusters' blocks (numbered cluster by cluster), [n_reps + 1] + list:
                                  // their members outside every block (positions in the cluster's member list)
    int32_t *mem_block = nullptr; // [members] block of the member (index into blk_*) x 2 + (1: the block's first leaf), -1: none
    int32_t *blk_root = nullptr;  // [n_blocks] root node
    int32_t *blk_rslot = nullptr; // [n_blocks] slot of the root inside its cluster
    int32_t *blk_nodes = nullptr; // [n_blocks] nodes strictly below the root (2 x leaves - 2)
    // emission order of the clustered fast path with blocks: tree-leaf slots and block roots together, by level (deepest first),
    // then node id -- what the level-sorted observation list of the sweep wants
    int32_t *e_of_slot = nullptr; // [n_refs] emission index of a slot (-1: not a tree leaf)
    int32_t *e_of_blk = nullptr;  // [n_blocks]
    int32_t *e_node = nullptr;    // [n_e] node of an emission index
    int32_t *lvl_e = nullptr;     // [height + 2] entry l + 1: emission indices with a level above l
    int64_t n_e = 0;
    int32_t *slot_node = nullptr; // [n_refs] tree node or -1
    int32_t *slot_level = nullptr;// [n_refs] level or -1
    int32_t *lvl_slots = nullptr; // [height + 2] entry l + 1: slots with a level above l (slots are sorted by level, deepest first); null
                                  // where they are not (scan layout)
    int32_t *slot_rep = nullptr;  // [n_refs] representative index of the member's cluster
    int32_t *slot_mpos = nullptr; // [n_refs] position inside its cluster
    int32_t *rep_slot = nullptr;  // [n_reps] slot holding the representative's sequence
    int32_t *rep_moff = nullptr;  // [n_reps+1]
    int32_t *mem_slot = nullptr;  // members in stored order, as slots
    std::vector<int32_t> row_slot;  // host: row -> slot
    std::vector<int32_t> slot_row;  // host: slot -> row
};

struct QueryBlock {
    int64_t n = 0, n_pad = 0;
    double *table = nullptr;      // distance-table block: [n][n_cols] fp64 in slot order (columns sorted by level)
    int64_t n_cols = 0;
    uint8_t *raw = nullptr;       // [n*L]
    uint4 *packed = nullptr;      // [n_pad/16][G][16][planes+1] uint4
    uint4 *packed8 = nullptr;     // the 8-plane form beside a 2-plane `packed` (DevAlign::packed8 exists: exact full rows), or nullptr
    int32_t *q_ex = nullptr;      // [n_pad] 1: the query carries a byte beyond ACGT- (k_pack_rows<2>)
    bool exact8 = false;          // the block's fused pass runs on the 8-plane forms (its queries carry such bytes): no matrix cores
    QueryBlock *ex_block = nullptr;  // the block's few queries with such bytes once more as an exact8 block of their own ...
    int32_t *ex_idx = nullptr;       // ... and which queries they are [ex_block->n] (run_block places them last and copies the structs over)
    uint8_t *qf4 = nullptr;       // fp4 operand image for the matrix-core distance kernels: [n_pad128][2G][4][32 B] (t1, t2, t3, v) for
                                  // k_jc69_mfma; beside a reference image (DevAlign::ref_f4) the compact tiled form of dist_gemm.hip
    uint8_t *aa_idx = nullptr;    // [n_pad][Lpad16] residue index (20 = gap)
    uint16_t *aa_mask = nullptr;  // [n_pad][Lpad16/16]
    uint8_t *sd_q4 = nullptr;     // scoredist beside a reference image (DevAlign::sd_ref4): the rounded-down table rows of the query's
                                  // residues as an fp4 operand image, same tiled form
    float *sd_nvq = nullptr;      // [image rows] the query's sites that are not gaps (-1: padding row)
    int32_t *self_slot = nullptr; // [n]
    apples_placement *out = nullptr;  // [n] device
    int planes = 0;
    int64_t col_gen = 0;          // distance-table block: generation of the column layout it was permuted with
    bool live = false;
};

// per-batch buffers that exist twice so that the distance/selection kernels of batch i+1 overlap
// the sweep of batch i (front stream vs back stream)
struct BatchBuf {
    double *dist = nullptr; uint32_t *counts = nullptr; int32_t *obs_node = nullptr; double *obs_dist = nullptr;
    int32_t *cnt_gt = nullptr, *n_obs = nullptr, *seg_slot = nullptr, *seg_cnt = nullptr; double *dist_slow = nullptr;
    int32_t *slow_list = nullptr, *slow_count = nullptr, *route_list = nullptr, *route_count = nullptr;
    int32_t *overflow_list = nullptr, *overflow_count = nullptr;
    int32_t *cls_list = nullptr, *cls_count = nullptr;
};

struct Workspace {
    int64_t batch = 0;            // queries per device batch
    int64_t dist_rows = 0;        // rows of dist / dist_slow: `batch`, or a slice of it (slim: only listed queries use full rows)
    bool slim = false;
    bool cslim = false;           // clustered fused route: seg_slot / seg_cnt rows of reps_pad entries, dist_slow a slice, dist whole
    int64_t stride = 0;           // row stride of dist/counts
    double *dist = nullptr;       // [batch][slots_pad] fp64 distances in slot order
    uint32_t *counts = nullptr;   // [batch][slots_pad] (mism<<16|valid), only when requested
    int32_t *obs_node = nullptr;  // [batch][obs_cap]
    double *obs_dist = nullptr;   // [batch][obs_cap]
    int32_t *cnt_gt = nullptr;    // [batch][height+2]: #observed leaves with level > l, at index l+1
    int32_t *n_obs = nullptr;     // [batch] emitted observed leaves (tree leaves, self removed)
    int64_t obs_cap = 0;
    // Ragged rows (the clustered fused route on the lean sweep, round 6).  The observation rows and the rows of member distances
    // (obs_node / obs_dist / dist) are sized for a query that observes every leaf -- 20 bytes x n_refs per query, 4 MB at 200 000
    // leaves, which is what bounds the device batch -- where the mean query's flat member list is a few thousand entries.  Ragged:
    // every query of the batch has a row of `row_small` entries at q x row_small; a query whose list is longer (k_select_clusters
    // phase 2), or that leaves the fast phases (phase 4, k_select), takes one of `row_big` rows of `stride` entries behind them, handed
    // out by a cursor (cls_count[22]); row_off[q] = where query q's rows start, in entries.  When the big rows run out the queries share
    // the last one (garbage, in bounds), cls_count[23] is raised and run_block repeats the block with full rows for good (ctx->no_ragged).
    bool ragged = false;
    int64_t row_small = 0, row_big = 0;
    int64_t *row_off = nullptr;   // [batch]
    // sweep scratch.  Small teams (one wavefront per query): `cap` nodes each; big teams (one
    // workgroup per query, for subtrees beyond cap): n_nodes each.
    struct Sweep {
        int32_t wgs = 0;          // workgroups of the launch
        int64_t teams = 0, cap = 0, leaf_cap = 0;
        uint32_t *map = nullptr;  // [teams][n_nodes] big trees: node -> tagged descriptor (sweep.hip NodeMap)
        uint32_t *ver = nullptr;  // [teams] last tag used in the team's map
        int32_t *order = nullptr; // [teams][cap+1] big trees: node ids in compact order
        int4 *ent = nullptr;      // [teams][cap+1] merge layout (sweep.hip): node, first two valid children, the first one's node id
        int32_t *grp_off = nullptr; // [teams][height+4] level groups in compact order, deepest first
        void *A = nullptr;        // [teams][cap+1] Rec (64 B): S then R tuple, first two valid children, node
        void *B = nullptr;        // [teams][cap+1][6] R values in waiting; trees with polytomies only
        void *lean = nullptr;     // sweep_lean.hip: [teams][LEAN_BYTES_PER_NODE * lean_cap1] field arrays (in place of ent and A)
        int64_t lean_cap1 = 0;    // entries per field array: cap + 1 rounded up to a multiple of 4
        int64_t lean_leaf1 = 0;   // observed leaves a team's per-leaf arrays hold (a multiple of 4)
        void *lean_leaf = nullptr; // wavefront-sized teams: [teams][lean_leaf1] per-leaf scratch of the bottom-up kernel (then `lean` is the
                                  // batch's pool of lean_cap1 entries, grp_off is per query and lean_meta holds what the top-down kernel needs)
        int4 *lean_meta = nullptr;
        double *xe = nullptr;     // [teams][cap+leaf_cap][18] per-edge x, err, R, S (HYBRID / inspection)
        // scan formulation: per team `cap` entries (one per subtree node) as component arrays
        double *ent_f = nullptr;  // [teams][13][cap]: S[6], R[6] as pairs, edge length
        int32_t *ent_i = nullptr; // [teams][5][cap]: ancestor-table row then node id, parent entry, first child entry, number of children, leaf index
        uint16_t *leaf_g = nullptr; // [teams][leaf_cap] per-leaf (level, lca level) when a team's LDS cannot hold them
        int32_t *meta = nullptr;  // [teams][4]: V, lca node, top level, -
    } small, big;
    int32_t *seg_slot = nullptr;       // [batch][stride] fused path: slots of the kept entries
    int32_t *seg_cnt = nullptr;        // [batch][stride/64]
    double *dist_slow = nullptr;       // [batch][stride] full rows for the top-up path
    int32_t *slow_list = nullptr;      // [batch]
    int32_t *slow_count = nullptr;     // [1]
    int32_t *route_list = nullptr;     // [3][batch] queries routed to big teams by the selection kernel (one list, or three by size)
    int32_t *route_count = nullptr;    // [1]
    int32_t *overflow_list = nullptr;  // [batch]
    int32_t *overflow_count = nullptr; // [1]
    int32_t *cls_list = nullptr;       // [4][batch] placeable queries by size class, largest class first
    int32_t *cls_count = nullptr;      // [8]: 4 class counts, then 3 dynamic-queue cursors (small / routed / overflow)
    BatchBuf alt;                      // the other buffer set (swapped in by the pipelined driver)
    bool has_alt = false;
};

struct apples_ctx {
    int device = 0;
    // tuning / experiment / test knobs (NAME -> value): apples_params.knobs ("NAME=value;NAME=value") over the process environment's
    // APPLES_* variables, both read ONCE, at context creation -- two contexts of one process may differ in every one of them, and no
    // launcher keeps a function-local static of its own any more (round 6; DESIGN.md lists them)
    std::unordered_map<std::string, long long> knobs;
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr;   // spare
    hipEvent_t ev_sel = nullptr, ev_big = nullptr;
    hipEvent_t ev_top[2] = {nullptr, nullptr};  // run_block's overlapped top-up chain: k_select_fast done / chain and its sweep done
    hipStream_t stream3 = nullptr;   // back stream: sweeps of batch i while the front stream works on batch i+1
    hipStream_t stream_big = nullptr; // the workgroup-sized sweep teams of a batch, beside sweep_lean.hip's wavefront-sized ones
    hipEvent_t ev_front[2] = {}, ev_back[2] = {}, ev_bigfree = nullptr;
    hipEvent_t ev_cl[2] = {};        // clustered fast path: the few-query second form of its last phase runs beside the first on stream2
    hipEvent_t ev_blk_time[2] = {};  // timing events around k_blocks_up for the device batch under way (run_block's pool)
    hipEvent_t ev_half[2] = {};      // the lean sweep of a small device batch in two halves (launch_sweep_lean)
    hipEvent_t ev_blk[2] = {};       // clade blocks: k_blocks_up runs on stream_big beside the selection's last phase (distances ready / tuples ready)
    std::string err;
    std::string desc;
    apples_params params{};
    uint32_t dbg = 0;  // APPLES_DBG_* (apples_params.debug at creation | the environment variables of the same names)
    DevTree tree;
    DevAlign aln;
    bool has_aln = false;
    double *jc_lut = nullptr;
    int64_t jc_lut_len = 0;
    GemmThreshold gemm_thr;
    int n_cu = 0;  // compute units of the device (dist_gemm.hip's persistent grid)
    int64_t cur_batch_queries = 0;  // queries of the device batch under way (api.hip:route_threshold)
    int32_t *jc_mmax = nullptr;  // [L+1] largest mismatch count with 0 <= lut <= threshold, per valid count
    int32_t *jc_mmax_true = nullptr;  // jc_mmax is the FILTER's table where the reference holds bytes beyond ACGT- (a pair's counts may still
                                      // grow by up to ex_max: set_params); this is the rule itself, what k_exotic_fix tests.  nullptr: jc_mmax is
    double *blosum = nullptr;  // 21x21 table (row/col 20 = gap -> 0)
    uint8_t *sd_tq4 = nullptr; // [20][20] fp4 codes of the table rounded down to the grid {0, .5, 1, 1.5, 2, 3, 4, 6} / 4 (dist_sd.hip)
    Workspace ws;
    std::vector<QueryBlock> blocks;
    // buffers of freed query blocks, kept for the next block (a host-buffer call would otherwise pay
    // hipMalloc + hipFree of a few hundred MB every time): size -> pointer, plus the size of every
    // block buffer handed out
    std::vector<std::pair<size_t, void *>> blk_cache;
    std::unordered_map<void *, size_t> blk_size;
    int *d_exotic = nullptr;         // device flag: a packed query block carried a symbol beyond ACGT-
    std::vector<hipEvent_t> ev_feed; // "chunk i of a streamed block is uploaded and packed"
    int32_t *d_slice_cnt = nullptr;  // [64] list lengths of the top-up slices (slim workspaces)
    uint8_t *sd_list_img = nullptr; int64_t sd_list_rows = 0;  // scoredist top-up: operand image of the listed queries (dist_sd.hip)
    int32_t *sd_list_ints = nullptr;  // ... [sd_list_rows] compact row lengths, [1] count of list2, [sd_list_rows] list2
    // scratch of the clustered fast path's cluster-major distance pass (select.hip), grown on demand
    int32_t *cl_ints = nullptr; int64_t cl_ints_cap = 0;
    int2 *cl_items = nullptr; int64_t cl_items_cap = 0;
    int4 *cl_tiles = nullptr; int64_t cl_tiles_cap = 0;
    int32_t *cl_big_scr = nullptr;  // k_select_clusters' third form: its lists (SELECT_CLUSTERS_BIG_LIST workgroups x 3 x HUGE_CAP)
    bool blk_active = false;   // the device batch under way names block roots in its observation lists (run_block)
    int32_t *d_rag = nullptr;  // [2] ragged rows: [0] a batch ran out of big rows
    bool no_ragged = false;    // a block ran out of big rows once (Workspace::ragged): this context keeps full rows from then on
    int32_t *blk_counters = nullptr;  // [3] inside blk_ints: items, work cursor, tiles of the last device batch
    double *blk_pool = nullptr; int64_t blk_pool_cap = 0;   // clade blocks: the batch's tuples, [tile][slot][6][64 lanes] doubles
    int32_t *blk_ints = nullptr; int64_t blk_ints_cap = 0;  // ... per item: storage base (-1: no blocks for it); per query: {first, count} of its items; the items
    int4 *blk_tiles = nullptr; int64_t blk_tiles_cap = 0;   // ... tiles of up to 64 items of one cluster: {cluster, first item, items, storage base / 64}
    double *sd_rep_d = nullptr; int64_t sd_rep_d_cap = 0;  // scoredist: [batch][reps_pad] the queries' distances to every representative
    unsigned long long *scan_prof = nullptr;  // APPLES_SCAN_PROFILE: per-phase cycle sums of the scan sweep
    unsigned long long *lean_prof = nullptr;  // APPLES_LEAN_PROFILE: the same for sweep_lean.hip's wavefront-sized teams
    // -d path: column layout cache
    int64_t col_gen = 0;             // bumped whenever the column layout below is replaced
    int64_t dcols = 0;
    int32_t *d_col_perm = nullptr;   // [n_cols] slot -> column, level-sorted
    int32_t *d_col_node = nullptr;   // [n_cols] slot -> node
    int32_t *d_col_level = nullptr;
    std::vector<int32_t> h_col_node;
    std::vector<int32_t> h_col_perm;
    // timing
    hipEvent_t ev[8] = {};
    std::vector<hipEvent_t> ev_pool;  // timing events of run_block, created once and reused
    double t_ms[APPLES_T_COUNT] = {};
};

extern thread_local std::string g_create_error;

// a knob's value (`def` where it is not set) / whether it is set at all
inline long long knob(const apples_ctx *ctx, const char *name, long long def) {
    const auto it = ctx->knobs.find(name);
    return it == ctx->knobs.end() ? def : it->second;
}
inline bool knob_on(const apples_ctx *ctx, const char *name) { return ctx->knobs.find(name) != ctx->knobs.end(); }

#define HIP_TRY(ctx, call)                                                                            \
    do {                                                                                              \
        hipError_t e__ = (call);                                                                      \
        if (e__ != hipSuccess) {                                                                      \
            (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e__);                          \
            return 1;                                                                                 \
        }                                                                                             \
    } while (0)

// ---- kernels' host launchers (defined in the .hip files) -----------------------------------------
// pack.hip
int launch_pack_rows(apples_ctx *ctx, const uint8_t *d_raw, int64_t n_rows, int L, int planes, uint4 *d_out,
                     int64_t slots_pad, bool query_layout, int *d_exotic, hipStream_t st = nullptr,
                     const int32_t *d_src_row = nullptr, int32_t *d_row_bad = nullptr);
int launch_pack_aa(apples_ctx *ctx, const uint8_t *d_raw, int64_t n_rows, int L, uint8_t *d_out, uint16_t *d_mask,
                   int64_t slots_pad, bool query_layout, hipStream_t st = nullptr, const int32_t *d_src_row = nullptr);
// dist.hip
int launch_counts(apples_ctx *ctx, const QueryBlock &qb, int64_t q0, int64_t nq, int tile, double *d_dist,
                  uint32_t *d_counts);
int launch_scoredist(apples_ctx *ctx, const QueryBlock &qb, int64_t q0, int64_t nq, double *d_dist,
                     uint32_t *d_counts);
int launch_scoredist_fused(apples_ctx *ctx, const QueryBlock &qb, int64_t q0, int64_t nq, double *seg_d, int32_t *seg_slot,
                           int32_t *seg_cnt, double *full_rows);
int launch_build_cluster_panels_aa(apples_ctx *ctx);  // scoredist: the representatives' rows in representative order
// scoredist of queries q0.. to every representative: full rows of reps_pad values (rep_d), then the survivors 0 <= d <= threshold
// per 64-representative segment as k_select_clusters reads them (position << 26 in seg_slot, counts in seg_cnt)
int launch_scoredist_reps(apples_ctx *ctx, const QueryBlock &qb, int64_t q0, int64_t nq, double *rep_d, int32_t *seg_slot,
                          int32_t *seg_cnt);
int launch_scoredist_listed(apples_ctx *ctx, const QueryBlock &qb, int64_t q0, int64_t nq_max, const int32_t *qlist,
                            const int32_t *qcount, double *d_dist);
// dist_sd.hip: the fused scoredist pass as a lower bound on the matrix cores + exact evaluation of the candidates
#define SD_GEMM_MAX_THRESHOLD 0.25  // -f beyond this: too many pairs pass for a filter to pay (full rows instead)
void sd_table_codes(const double *blosum20x20, uint8_t *codes, bool fp6);
int64_t sd_query_image_bytes(const apples_ctx *ctx, int64_t rows256);  // bytes of a query operand image of that many rows
int sd_steps(int L);                // 128-value K steps of the operand images
bool sd_gemm_usable(const apples_ctx *ctx);
int launch_sd_expand(apples_ctx *ctx, const uint8_t *d_raw, int64_t n, uint8_t *d_out, int64_t n_img, hipStream_t st,
                     const int32_t *d_src_row, int64_t row0, bool query, float *d_nv, const int32_t *d_n = nullptr,
                     const uint16_t *d_mask = nullptr, int64_t ref_stride = 0);
int launch_sd_topup(apples_ctx *ctx, const QueryBlock &qb, int64_t q0, int64_t nq_max, const int32_t *qlist, const int32_t *qcount,
                    uint8_t *img, double *lbrows, double *out_rows, int32_t *len = nullptr, int32_t *list2 = nullptr,
                    int32_t *count2 = nullptr);
int launch_sd_topup_rows_again(apples_ctx *ctx, const QueryBlock &qb, int64_t q0, int64_t nq_max, const int32_t *qlist,
                               const int32_t *qcount, double *lbrows, double *out_rows);
int sd_compact_cap(const apples_ctx *ctx);  // entries a compact row of launch_sd_topup holds
int launch_sd_rows(apples_ctx *ctx);
int launch_sd_filter(apples_ctx *ctx, const QueryBlock &qb, int64_t q0, int64_t nq, int32_t *seg_slot, int32_t *seg_cnt);
int launch_sd_exact(apples_ctx *ctx, const QueryBlock &qb, int64_t q0, int64_t nq, double *seg_d, const int32_t *seg_slot,
                    const int32_t *seg_cnt, int32_t *n_surv);
// select.hip
// where query q's row starts (entries): ragged rows (Workspace::ragged) or q x pitch
__device__ __forceinline__ int64_t row_start(const int64_t *row_off, int64_t q, int64_t pitch) { return row_off ? row_off[q] : q * pitch; }

struct SelectArgs {
    const double *dist;       // [nq][stride]
    int64_t stride;
    const int32_t *gather;    // slot -> column (distance-table path) or nullptr
    const int32_t *slot_node, *slot_level, *slot_rep, *slot_mpos, *rep_slot, *rep_moff, *mem_slot;
    int64_t n_members, n_reps;
    int all_singleton;
    int table_mode;           // 1: -d semantics (rows not in the tree are ignored entirely)
    int cols_all_in_tree;     // table mode: every column is a tree leaf (no per-entry node test while streaming)
    const int32_t *self_slot; // [nq] or nullptr
    double thr;
    int baseobs;
    int height;
    int32_t *obs_node; double *obs_dist; int64_t obs_cap; int32_t *cnt_gt; int32_t *n_obs;
    // ragged rows (Workspace::ragged): where a query's observation row and its row of member distances start; nullptr: q x obs_cap / q x stride
    int64_t *row_off; int64_t row_small, row_big_base, row_big_pitch; int32_t row_big_n; int32_t *row_big_cursor, *row_fail;
    apples_placement *out;    // [nq]
    // fused fast path (k_jc69 MODE 1 -> k_select_fast)
    const int32_t *seg_slot, *seg_cnt;  // [nq][stride], [nq][stride/64]
    const double *seg_lut;              // non-null: seg_slot holds position << 26 | valid << 13 | mism, distances are seg_lut[...]
    const int32_t *seg_surv;            // non-null (dist_sd.hip): the segments hold CANDIDATES, the ones that failed marked by a negative
                                        // distance; seg_surv[query] = how many passed
    // k_select_fast behind a matrix-core pass over reference rows with bytes beyond ACGT- (DevAlign::ex_off): the survivors on such
    // rows get those sites counted (valid and mismatching under a letter of the query) and 0 <= d <= threshold tested once more
    const int32_t *ex_off; const uint16_t *ex_site;  // CSR over slots
    const int32_t *ex_mmax;             // non-null: do it; the rule's own table (apples_ctx::jc_mmax_true, or jc_mmax where the pass's was not loosened)
    int ex_all;                         // the pass's table was loosened: every survivor is tested again, not only those on such rows
    const uint8_t *q_raw;               // the batch's queries as bytes ([nq][L])
    const int32_t *node_level;          // tree level by node id
    int32_t *slow_list, *slow_count;    // queries that need the top-up rule
    int32_t *slow_hint;                 // [list position] or nullptr: what the listing kernel already knows about the query -- its
                                        // count of observations inside the threshold (k_select then skips the pass that
                                        // would only count them again), -1 = not known
    int32_t *cls_list, *cls_count;      // size-class work lists for the small-team sweep
    int64_t cls_stride;
    int big_threshold;                  // n_obs above this -> straight to the big-team sweep list
    int32_t *overflow_list, *overflow_count;
    // k_select_stream on compact rows (dist_sd.hip:k_sd_topup<true>): row r holds row_len[r] distances in slot order and, from
    // double `row_cap` on, their slots as 32-bit integers; row_len[r] < 0: not this launch's row.  nullptr: rows of n_members values
    const int32_t *row_len;
    int row_cap;
    int32_t *row_cursor;                // k_select_stream: the next row to hand out (cleared by launch_select), or nullptr = rows dealt out in advance
    int third_pass;                     // k_select_stream: a third pass over a row that needs the top-up rule instead of the merge in LDS (diagnostic)
    int route_classes;                  // the routed queries go to three lists by size (each cls_stride long, counts at overflow_count[4..6]):
                                        // sweep_lean.hip's workgroup-sized teams take the largest first
    // listed mode of k_select: block r handles query qlist[r] with distances in row r (rows_by_query: in row qlist[r])
    const int32_t *qlist, *qcount;
    const int32_t *qhint;     // [list position] or nullptr: see slow_hint
    int rows_by_query;
    // top-up by segment minima (k_jc69 MODE 2 -> k_select_topup): [listed row][stride]
    const double *segmin_d; const int32_t *segmin_i;
    // clustered fast path (k_select_clusters): survivors among the representatives come in seg_slot/seg_cnt with
    // row stride rep_stride; member rows are read row-major, the query's words from its packed tile
    const uint4 *packed_rm; const uint4 *qpacked; int G; int L; double overlap; int64_t rep_stride;
    int cl_mfma;                        // the member distances of accepted clusters on the matrix cores (k_cluster_dist_mfma), not by bit counts (k_cluster_dist)
    double *tmp_d;            // [nq][stride] member distances before the ordered emission
    int flat_pref;            // k_select_fast: the segment counts' prefix fits LDS (set by the launcher)
    // cluster-major member distances (k_select_clusters phases 1-3 around k_cluster_tiles / k_cluster_dist): per cluster the
    // queries that accepted it, cut into tiles
    int32_t *cl_count, *cl_start, *cl_fill, *cl_ntiles;  // [n_reps], [n_reps + 1], [n_reps], [1]
    int2 *cl_items;           // [sum of the counts] (query, offset of the cluster's members in the query's flat member list)
    int4 *cl_tiles;           // (cluster, first item, items, -)
    int64_t cl_tiles_cap;
    const int32_t *lvl_slots; // see DevAlign
    // phase 4 of k_select_clusters (the slow list of the clustered fast path): the representative panel ([(g, plane)][rep_stride]),
    // and where it forwards what it cannot serve (then full rows + k_select)
    const uint4 *rep_panel; int32_t *slow2_list, *slow2_count;
    int32_t *big_list, *big_count;  // queries with more than ACC_CAP accepted clusters: served by the phases' second form (CAP = BIG_CAP)
    int32_t *gen_list, *gen_count;  // phase 3 with clade blocks: the queries the short-form launch leaves to the general form (k_select_clusters, SHORT_ONLY)
    int32_t *big_scr;  // the third form's lists (CAP = HUGE_CAP): 3 x HUGE_CAP ints per workgroup of its launches
    // scoredist contexts on that path (k_cluster_dist_sd, phase 4): the representatives' distances come as full rows (the survivors
    // in seg_slot carry their position only), the members' from the packed residue bytes
    const double *rep_dist;   // [nq][rep_stride] or nullptr (JC69: seg_lut)
    const uint8_t *aa_idx; const uint16_t *aa_mask;   // DevAlign's, row stride `stride`
    const uint8_t *aa_cm_idx; const uint16_t *aa_cm_mask; int64_t cm_pad;  // the same rows in cluster-major member order (row stride cm_pad), or nullptr
    const uint8_t *q_aa; const uint16_t *q_aam;       // the batch's queries: QueryBlock::aa_idx / aa_mask from its first query on
    int Lpad; const double *table;                    // 21 x 21
    // clade blocks (DevAlign::blk_*): k_cluster_tiles also cuts every cluster's items into tiles of up to 64 for the block kernels,
    // phase 2 notes every query's items, k_blocks_up leaves the blocks' S tuples in the pool, phase 3 emits block roots
    const int4 *blk_rec_i; const double2 *blk_rec_e; const double2 *blk_rec_c; const double *blk_stat; const int32_t *rep_soff, *mem_block, *blk_root, *blk_rslot, *blk_nodes;
    const int32_t *e_of_slot, *e_of_blk, *e_node, *lvl_e; int64_t n_e;
    const int32_t *rep_boff, *rep_loff, *loose_mp;  // (the short form of the last phase)
    double *blk_pool; int64_t blk_pool_cap;  // (doubles)
    int4 *blk_tiles; int64_t blk_tiles_cap; int32_t *blk_ntiles;
    int32_t *q_blk;           // [nq] 1: the query's observation list names block roots (k_blocks_down / k_blocks_finish serve it)
    int32_t *item_sbase;      // [items] first slot of the item's tile in blk_pool x 64 + the item's lane (slot s, component x, lane l at
                              // ((base + s) * 6 + x) * 64 + l); -1: no room in the pool / no blocks in the cluster (phase 2)
    int32_t *item_bad;        // [items] 1: a member the reference drops, the query's own row or an exact match among the cluster's
                              // members -- the item goes without blocks (k_cluster_dist)
    const int32_t *cl_order;  // [n_reps] the clusters by falling number of block-internal nodes (k_cluster_tiles walks them in this order), or nullptr
    int32_t *cl_bbase;        // [n_reps] first slot of the cluster's tiles in the pool, -1: the cluster has no blocks (k_cluster_tiles)
    int2 *q_items;            // [nq] {first entry of q_item, accepted clusters}
    int32_t *q_item;          // [items] the queries' items, a query's in the order of its accepted clusters
    int32_t *q_item_cursor;   // [1]
    int method;               // (k_blocks_up)
    int rep_cache;            // k_select, clustered rows: representatives whose distances are staged in LDS (set by the launcher; 0 = none)
    int64_t n_rows_plain;     // k_select / k_select_stream without a list: rows to select (set by the launcher; the grid may be smaller)
};
int launch_select(apples_ctx *ctx, const SelectArgs &a, int64_t nq);
int launch_select_fast(apples_ctx *ctx, const SelectArgs &a, int64_t nq);
int launch_select_topup(apples_ctx *ctx, const SelectArgs &a, int64_t nq);
int launch_select_clusters(apples_ctx *ctx, const SelectArgs &a, int64_t nq);  // needs n_members <= SELECT_CLUSTERS_MAX_SLOTS
int launch_select_clusters_listed(apples_ctx *ctx, const SelectArgs &a, int64_t nq_max);  // its slow list (qlist / qcount / qhint), top-up rule included
#define SELECT_CLUSTERS_ACC_CAP 512   // accepted clusters per query on the fast path (more: the query takes the general route)
#define SELECT_CLUSTERS_MIN_TILE 16   // fewest queries a full tile of k_cluster_dist holds
#define SELECT_CLUSTERS_BIG_CAP 5120  // ... and on its second form, for the few queries beyond ACC_CAP (one workgroup per CU: 60 KB of lists)
#define SELECT_CLUSTERS_HUGE_CAP 16384 // ... and on the third, for references of more than BIG_CAP clusters: the offsets in LDS (64 KB), the other lists in global scratch
#define SELECT_CLUSTERS_BIG_LIST 1024 // queries a batch may send to that form (more: the general route)
#define SELECT_CLUSTERS_MAX_SLOTS 524288  // k_select_clusters' bitmap: 256 threads x runs of 32 words (80 KB of LDS at that size; 229 376 until round 6)
int launch_counts_reps(apples_ctx *ctx, const QueryBlock &qb, int64_t q0, int64_t nq, int32_t *seg_slot, int32_t *seg_cnt);
int launch_build_cluster_panels(apples_ctx *ctx);  // listed queries, needs segmin_d/segmin_i; baseobs <= 256
int launch_permute_cols(apples_ctx *ctx, const double *in, double *out, const int32_t *perm, int64_t nq, int64_t n_cols);
bool dist_mfma_enabled(const apples_ctx *ctx);
bool fused_counts_format(const apples_ctx *ctx, const QueryBlock &qb);
// (contexts with a reference image keep compact, tiled images: d_out is then the image's base and row0 the image row
// of d_raw's first row)
int launch_expand_queries_f4(apples_ctx *ctx, const uint8_t *d_raw, int64_t n, uint8_t *d_out, int64_t n_pad,
                             hipStream_t st, const int32_t *d_src_row = nullptr, int64_t row0 = 0);
// dist_gemm.hip
bool dist_gemm_usable(const apples_ctx *ctx);
int launch_counts_gemm(apples_ctx *ctx, const QueryBlock &qb, int64_t q0, int64_t nq, int32_t *seg_slot, int32_t *seg_cnt);
int launch_counts_fused(apples_ctx *ctx, const QueryBlock &qb, int64_t q0, int64_t nq, int tile, double *seg_d,
                        int32_t *seg_slot, int32_t *seg_cnt);
int launch_counts_listed(apples_ctx *ctx, const QueryBlock &qb, int64_t q0, int64_t nq_max, const int32_t *qlist,
                         const int32_t *qcount, double *d_dist, double *segmin_d, int32_t *segmin_i);
// the fused matrix-core pass's survivors (packed words in seg_slot) on reference rows with bytes beyond ACGT-: exact counts, the
// threshold once more, n_surv[query] = how many stay (k_select_fast: SelectArgs::seg_surv)
int launch_exotic_fix(apples_ctx *ctx, const QueryBlock &qb, int64_t q0, int64_t nq, int32_t *seg_slot, const int32_t *seg_cnt,
                      int32_t *n_surv);  // (diagnostic form: api.hip does the same inside k_select_fast)
bool exact8_rows(const apples_ctx *ctx, const QueryBlock &qb);  // the bit-plane kernels take the 8-plane forms for this block
// sweep.hip
struct SweepArgs {
    DevTree tree;
    const int32_t *obs_node; const double *obs_dist; int64_t obs_cap; const int32_t *cnt_gt; const int32_t *n_obs;
    const int64_t *row_off;   // ragged rows (Workspace::ragged) or nullptr
    int32_t *grp_off; void *A, *B; double *xe;
    int grp_stride;           // sweep_lean.hip: ints per query / team in grp_off (height + 4; twice that on a tree with polytomies: the child records' offsets)
    uint32_t *map;            // big trees: [teams][n_nodes] tagged node map; nullptr = node bits in LDS
    uint32_t *map_ver;        // [teams] version tags of the maps
    int32_t *order;           // [teams][cap+1]
    int4 *ent;                // [teams][cap+1] merge layout: nullptr = node map or node bits
    void *lean;               // sweep_lean.hip: the teams' field arrays (workgroup-sized teams) or the batch's pool (wavefront-sized)
    void *lean_leaf;          // wavefront-sized bottom-up teams: per-leaf scratch, lean_leaf1 entries per team
    int64_t lean_teams;       // ... for this many teams
    int4 *lean_meta;          // [batch] per query: {offset in the pool, number of level groups G (-1: not swept here), internal nodes, -}
    unsigned int *pool_cursor; // next free pool entry (cleared per batch)
    double *blk_pool;          // clade blocks: an observed "leaf" whose distance is a boxed index (lean_is_block) is a block root, its tuple at
                               // blk_pool[index + x * 64] (S after k_blocks_up; the top-down pass leaves lift(R) there for k_blocks_down)
    unsigned long long *prof; // diagnostic (APPLES_LEAN_PROFILE): [16] cycles and step counts per phase summed over teams, or nullptr
    int64_t lean_cap1, lean_leaf1;
    int map_bits;             // payload bits of a map entry; the tag sits above them
    int method, criterion, negative;
    int keep_edges;           // store per-edge x/err (inspection or HYBRID)
    int debug_phase;          // timing experiments only: 1 = stop after the bottom-up pass
    int64_t cap;              // internal nodes of A/B/xe scratch per team
    int64_t leaf_cap;         // observed leaves a team's xe area can hold beyond `cap`
    int big_threshold;        // small teams skip queries with more observed leaves (already listed for big teams)
    const int32_t *work_list; // queries to process (nullptr = 0..nq-1)
    const int32_t *work_count;// device count of work_list entries (nullptr = nq)
    int route_classes;        // work_list = three lists of cls_stride entries by size class, counts at work_count[4..6] (largest first)
    const int32_t *cls_list;  // size-class lists (small teams): queue index -> query, largest class first
    const int32_t *cls_count; // [4] class counts
    int64_t cls_stride;
    int32_t *cursor;          // dynamic work queue: teams take the next entry with one atomic add
    int w_mod, w_rem;         // sweep_lean.hip's wavefront-sized teams: this launch takes the queue entries w with w % w_mod == w_rem (0, 0: all)
    int32_t *overflow_list;   // queries whose subtree exceeded `cap`
    int32_t *overflow_count;
    apples_placement *out;
};
// clade blocks (sweep_lean.hip: k_blocks_up / k_blocks_down / k_blocks_finish; DevAlign::blk_* for the static part)
struct BlockArgs {
    const int4 *tiles; const int32_t *n_tiles;  // {cluster, first item, items (<= 64), first slot of the tile's tuples in the pool or -1}
    const int2 *items;                          // (query, where the cluster's members start in the query's flat member list)
    const int4 *rec_i; const double2 *rec_e; const double *stat; const int32_t *rep_soff, *rep_moff, *slot_rep, *slot_mpos;
    const double2 *rec_c; const int2 *rec_p; const int2 *pk_i; const double *pk_e;  // (polytomies inside blocks: DevAlign::blk_rec_c ...)
    const int32_t *self_slot;                   // [nq] the queries' own rows as slots, or nullptr
    const double *tmp_d; int64_t stride;        // the queries' rows of member distances
    const int64_t *row_off;                     // ... ragged (Workspace::ragged) or nullptr
    double *pool;                               // [slot][6][64 lanes]; a tile's slot 0: the lanes' best edges inside the blocks (key, x1, x2,
                                                // err, e, (x1 is the int 0, edge)), its slots 1 ..: the tuples of the cluster's block-internal nodes
    const int32_t *item_sbase, *item_bad;       // [items] first slot x 64 + lane (-1: none); 1: the item goes without blocks
    const int32_t *q_blk; const int2 *q_items; const int32_t *q_item;
    int32_t *cursor;
    int method, criterion, negative;
    apples_placement *out; int64_t nq;
};
int launch_blocks_up(apples_ctx *ctx, const BlockArgs &a, hipStream_t st);
int launch_blocks_down(apples_ctx *ctx, const BlockArgs &a, hipStream_t st);
// sweep_scan.hip
struct ScanArgs {
    const int4 *leaf_info; const AncRec *anc; const uint8_t *rmq;
    int32_t euler_len;
    int32_t n_nodes, height;
    const int32_t *obs_node; const double *obs_dist; int64_t obs_cap; const int32_t *n_obs;
    const int64_t *row_off;   // ragged rows (Workspace::ragged) or nullptr
    double *ent_f; int32_t *ent_i; double *xe; uint16_t *leaf_g; int32_t *meta;
    int64_t cap;              // entries of scratch per team
    int64_t leaf_cap;         // leaves a team's global leaf-state area holds (0: none)
    int lds_leaves;           // leaves a team's LDS leaf-state area holds
    int method, criterion, negative, keep_edges;
    int big_threshold;
    const int32_t *work_list; const int32_t *work_count;
    const int32_t *cls_list; const int32_t *cls_count; int64_t cls_stride;
    int32_t *cursor;
    int32_t *overflow_list; int32_t *overflow_count;
    apples_placement *out;
    unsigned long long *prof;  // diagnostic (APPLES_SCAN_PROFILE): [8] cycles per phase summed over teams, or nullptr
};
int launch_scan_mixed(apples_ctx *ctx, const ScanArgs &small, const ScanArgs &big, int64_t nq, int wgs, int n_big, hipStream_t st);
int launch_scan(apples_ctx *ctx, const ScanArgs &a, int64_t nq, int wgs, int team, hipStream_t st);
#define SCAN_LDS_LEAVES_SMALL 2048   // per wavefront-sized team (4 per workgroup); 6 bytes of LDS per leaf
#define SCAN_LDS_LEAVES_BIG 8192     // per workgroup-sized team
bool sweep_merge_lists(const DevTree &t);  // big binary trees, wavefront-sized teams: level lists by merging, no node map
bool sweep_bits_in_lds(const DevTree &t);  // the sweep's node bits fit in LDS (else: tagged node map in global scratch)
int launch_sweep(apples_ctx *ctx, const SweepArgs &a, int64_t nq, int wgs, int team, hipStream_t stream = nullptr);
// sweep_lean.hip: the three-pass form for big binary trees (wavefront-sized teams over the size-class queues)
// An observed "leaf" that is the root of a clade block travels through the observation list as a boxed index into the blocks' pool:
// a negative quiet NaN with the tag 0b101 under the quiet bit (no arithmetic produces that payload: the default NaN has none) and
// the index in the low 48 bits (select.hip boxes, sweep_lean.hip:lean_is_block tests the whole 16-bit prefix)
#define APPLES_BLOCK_BOX 0xFFFD000000000000ull
// clade blocks: flags in the fourth component of a record (DevAlign::blk_rec_i) above the right child's node id; the third component
// of a chain's inner / last record whose left operand is the partial sum: BLK_F_NOEDGE
#define BLK_MAX_DEG 5              // children of a node inside a block (more: the block is cut there, as every polytomy was until round 6)
#define BLK_F_ROOT 0x40000000      // the record is a block's root
#define BLK_F_POLY 0x20000000      // the record is a node of more than two children (its chain's last record): blk_rec_p names them all
#define BLK_F_PART 0x10000000      // the record is an inner record of a chain: a partial sum, not a node
#define BLK_NODE_MASK 0x0fffffff   // node ids below
#define BLK_F_NOEDGE 0x0fffffff    // "node id" of a left operand that is a partial sum
#define LEAN_BYTES_PER_NODE 100  // T0 T1 T2 E DD (16 B each), D N (8 B each), K (4 B)
#define LEAN_BYTES_PER_LEAF 12   // per observed leaf: edge length, parent
#define LEAN_SMALL_BATCH 13312    // device batches up to this many queries: routing cut halved (api.hip:route_threshold), 512-thread routed teams
#define LEAN_BIG_THRESHOLD 8192  // observed leaves above which a query goes to the lean sweep's workgroup-sized teams
#define LEAN_MAX_LEVELS 256      // per-level offsets of a query in LDS: a window of this many levels (deeper trees: it follows the walk)
bool sweep_lean_layout(const DevTree &t, bool per_edge_records);
int launch_sweep_lean(apples_ctx *ctx, const SweepArgs &up, const SweepArgs &down, int64_t nq, hipStream_t st, int32_t *halves = nullptr,
                      hipStream_t side = nullptr, hipEvent_t *ev = nullptr);
int sweep_lean_up_teams(const apples_ctx *ctx);
int launch_sweep_lean_big(apples_ctx *ctx, const SweepArgs &a, int64_t nq, int wgs, hipStream_t st);
int launch_sweep_mixed(apples_ctx *ctx, const SweepArgs &small, const SweepArgs &big, int64_t nq, int wgs, int n_big,
                       hipStream_t st);
