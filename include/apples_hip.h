/*
 * apples_hip.h -- C ABI of the MI355X-native APPLES hot path (libapples_hip.so).
 *
 * The reference (balabanmetin/apples, pure Python) has no FFI layer; its seams are Python
 * call signatures.  Each entry point below names the reference interface it
 * replaces (file:line relative to the reference tree).  Plain C: int status returns
 * (0 = ok), caller-owned contiguous buffers, one opaque context per device, no globals.
 * A context is thread-compatible: one host thread at a time.
 *
 *   seam B1  pool.starmap(PoolQueryWorker.runquery, queries)      run_apples.py:94-102
 *   seam B2  Reference.get_obs_dist / dist_function               apples/Reference.py:22-29,117-157
 *   seam B3  Algorithm.dp_frag / placement_per_edge / placement   apples/Algorithm.py:16-101
 */
#ifndef APPLES_HIP_H
#define APPLES_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct apples_ctx apples_ctx;

/* -m / --method (apples/PoolQueryWorker.py:104-111: anything else means OLS) */
enum { APPLES_OLS = 0, APPLES_FM = 1, APPLES_BME = 2, APPLES_BE = 3 };
/* -c / --criterion (apples/Algorithm.py:76-91: anything else means MLSE) */
enum { APPLES_MLSE = 0, APPLES_ME = 1, APPLES_HYBRID = 2 };
/* distance model: jc69 (apples/distance.py:718-745) or scoredist (:681-715), chosen by -p
 * as in apples/Reference.py:22-25 */
enum { APPLES_JC69 = 0, APPLES_SCOREDIST = 1 };

/* Backbone tree as arrays indexed by edge_index = left-to-right post-order number
 * (apples/util.py:57-69); level = BFS depth (apples/util.py:72-88); children in file order. */
typedef struct {
    int32_t n_nodes;
    const int32_t *parent;    /* [n_nodes], -1 for the root (= n_nodes-1) */
    const double *edge_len;   /* [n_nodes] */
    const int32_t *child_off; /* [n_nodes+1] CSR */
    const int32_t *child_idx; /* [n_nodes-1] */
    const int32_t *level;     /* [n_nodes] */
} apples_tree;

/* Reference alignment in the reference's own encoding: 1 byte per site, '-' = gap, any other
 * byte an ordinary symbol (apples/fasta2dic.py:56-71).  Rows 0..n_refs-1 are the reference
 * sequences (Reference.refs, apples/Reference.py:19); rows n_refs..n_rows-1 are consensus
 * sequences of multi-member clusters.  Clusters (Reference.representatives,
 * apples/Reference.py:107) in representative order i: rep_row[i] = row holding the
 * representative sequence, members = member_row[member_off[i]..member_off[i+1]) in stored
 * order.  rep_row == NULL means every reference is its own singleton cluster in row order
 * (apples/PoolRepresentativeWorker.py:99-101). */
typedef struct {
    int64_t n_rows;
    int64_t n_refs;
    int32_t length;           /* L */
    const uint8_t *rows;      /* [n_rows * L] row-major */
    const int32_t *row_node;  /* [n_refs] tree node of each reference row, -1 if not a leaf of the tree */
    int64_t n_reps;
    const int32_t *rep_row;   /* [n_reps] or NULL */
    const int32_t *member_off;/* [n_reps+1] or NULL */
    const int32_t *member_row;/* [member_off[n_reps]] or NULL */
} apples_alignment;

/* Options that reach the per-query worker (apples/OptionsRun.py, apples/OptionsBasic.py). */
typedef struct {
    int32_t model;            /* APPLES_JC69 | APPLES_SCOREDIST (-p) */
    int32_t method;           /* -m */
    int32_t criterion;        /* -c */
    int32_t negative_branch;  /* -n */
    double filt_threshold;    /* -f (default 0.2) */
    int32_t base_observation; /* -b (default 25) */
    double overlap_frac;      /* -V (default 0.001) */
    /* Optional JC69 table: jc_lut[valid*(valid+1)/2 + mism] = distance for the integer pair,
     * valid in [0, L]; lets the host fill it with numpy's own log so distances carry the
     * reference's bits.  NULL: the kernel evaluates -0.75*log(1-4p/3) itself. */
    const double *jc_lut;
    int64_t jc_lut_len;
    int64_t max_batch;        /* queries per device batch; 0 = choose from free memory */
    /* Diagnostic switches (APPLES_DBG_*), read when the context is created (the device layouts depend on them) and
     * ignored by apples_set_params: alternative routes to the same placements, for tests that cross them inside one
     * process.  0 = the measured-best defaults.  The environment variables of the same names (APPLES_NO_FUSE ...) set
     * the same bits for a whole process. */
    uint32_t debug;
    /* Cap of the device batch buffers in GiB; 0 = none.  The library sizes them from free memory (at most 144 GiB or half of
     * what is free); the cap only ever lowers that: min(cap, free-memory budget).  A one-shot command-line run sets 24 (beyond
     * a few tens of GiB the allocation itself takes seconds, apples_amd/worker.py). */
    int32_t batch_gib;
    /* Tuning, experiment and test knobs of THIS context, "NAME=value;NAME=value" (the APPLES_ prefix optional, a bare NAME means 1),
     * or NULL.  Read once, at apples_ctx_create, over the process environment's APPLES_* variables (which give the same knobs to
     * every context of a process); apples_set_params ignores it.  The library keeps no other hidden state: two contexts of one
     * process may differ in every knob.  The names and defaults: DESIGN.md, "Knobs". */
    const char *knobs;
} apples_params;

#define APPLES_DBG_NO_FUSE       1u   /* full distance rows + general selection instead of the fused epilogue */
#define APPLES_DBG_SWEEP_SCAN    2u   /* scan formulation of the sweep (sweep_scan.hip) */
#define APPLES_DBG_NODE_MAP      4u   /* tagged node map of the level loop also for small trees */
#define APPLES_DBG_SWEEP_MERGE   8u   /* merged level lists also for small trees */
#define APPLES_DBG_NO_SWEEP_MERGE 16u /* tagged node map instead of merged level lists on big trees */
#define APPLES_DBG_NO_DIST_GEMM  32u  /* fused distance pass through the bit-plane-fed MFMA kernel, no reference image */
#define APPLES_DBG_NO_SWEEP_LEAN 64u  /* level loop of sweep.hip where sweep_lean.hip would run */
#define APPLES_DBG_NO_SD_GEMM    128u /* scoredist: the fused pass evaluates every pair (k_scoredist), no matrix-core filter */
#define APPLES_DBG_CLUSTER_BY_QUERY 256u  /* clustered route: a thread per (query, member) pair (phase 0 of k_select_clusters) */
#define APPLES_DBG_NO_CLUSTER_TOPUP 512u  /* clustered route: its listed queries through full rows + general selection */
#define APPLES_DBG_NO_STREAM_SELECT 1024u /* singleton rows / -d tables: general selection kernel instead of the streaming one */
#define APPLES_DBG_NO_TOPUP_KERNEL 2048u  /* JC69 top-up list: general selection over the full rows, no segment minima */
#define APPLES_DBG_NO_CLUSTER_BIG 4096u   /* clustered route: queries beyond 512 accepted clusters to the general route */
#define APPLES_DBG_NO_SD_TOPUP   8192u    /* scoredist top-up list: full rows (k_scoredist listed), no lower-bound rows */
#define APPLES_DBG_SD_FP6        16384u   /* scoredist filter: query-side table values as fp6 (half the candidates, a slower filter: measured slower in all) */
#define APPLES_DBG_NO_TOPUP_OVERLAP 32768u /* the top-up chain of a device batch (full rows + selection of the listed queries) before the sweep, not beside it */
#define APPLES_DBG_STREAM_THIRD_PASS 65536u /* k_select_stream: a row that needs the top-up rule is streamed a third time instead of the merge in LDS */
#define APPLES_DBG_NO_SD_COMPACT  131072u  /* scoredist top-up: rows of n_slots values for the selection instead of compact lists */
#define APPLES_DBG_SD_COMPACT_TINY 262144u /* ... compact lists of 16 entries: nearly every listed query overflows into the row form */
#define APPLES_DBG_NO_BLOCKS     524288u  /* clustered route: no clade blocks (every observed leaf goes through the per-query merged sweep) */
#define APPLES_DBG_HYBRID_RECORDS 1048576u /* -c HYBRID: per-edge records + the level loop on every tree (the form of rounds 1 - 4), not the lean sweep's ranking */
#define APPLES_DBG_NO_CLUSTER_MFMA 2097152u /* clustered route, JC69: the member distances of accepted clusters by bit counts (k_cluster_dist), not on the matrix cores */
#define APPLES_DBG_ALL           4194303u /* every defined switch; other bits of apples_params.debug are ignored */

/* One placement = the p row runquery returns, [edge_num, likelihood(error), 1, distal, pendant]
 * (apples/Algorithm.py:98-101, apples/PoolQueryWorker.py:36-37,74,88,119-125). */
typedef struct {
    int32_t edge;             /* edge_index; -1 = cannot be placed (fewer than 3 distances) */
    uint32_t flags;           /* APPLES_F_* */
    double error;
    double distal;            /* edge_length - x_2 */
    double pendant;           /* x_1 */
    int32_t n_obs;            /* len(obs_dist) the worker saw (after removing the query's own entry) */
    int32_t n_valid;          /* Subtree.num_nodes (apples/Subtree.py:42) */
} apples_placement;

#define APPLES_F_EXACT        1u  /* a zero distance: p = [edge,0,1,0,0] (PoolQueryWorker.py:72-75) */
#define APPLES_F_INSUFFICIENT 2u  /* len(obs_dist) <= 2 (PoolQueryWorker.py:97-98), edge = -1 */
#define APPLES_F_MISPLACED    4u  /* potential_misplacement_flag (Algorithm.py:94-97) */
#define APPLES_F_PENDANT_INT  8u  /* x_1 is the clamped Python int 0, printed "0" not "0.0" (util.py:32-50) */
#define APPLES_F_ZERO_NOT_IN_TREE 16u /* zero distance to a reference that is not a tree leaf: the
                                         reference raises KeyError (PoolQueryWorker.py:74) */
#define APPLES_F_DEGENERATE  32u  /* >=3 distances but fewer than two of them on tree leaves */

/* ABI of this header: bumped whenever a struct above grows or an entry point changes (4 = apples_params.debug with the
 * switches up to APPLES_DBG_ALL; 5 = apples_params.batch_gib; 6 = APPLES_T_BLOCKS, APPLES_DBG_NO_BLOCKS; 7 = APPLES_DBG_HYBRID_RECORDS, APPLES_DBG_NO_CLUSTER_MFMA; 8 = apples_params.knobs, apples_device_log).  apples_params has no size field: a caller must zero-initialise it (memset / = {0}) and
 * be built against the header of the library it loads -- check apples_abi_version() == APPLES_ABI_VERSION and
 * apples_params_size() == sizeof(apples_params) once at start-up, as apples_amd/engine.py does.  Bits of `debug` beyond
 * APPLES_DBG_ALL are ignored. */
#define APPLES_ABI_VERSION 8u
uint32_t apples_abi_version(void);
size_t apples_params_size(void);

/* Build a context on HIP device `device`: uploads the tree, packs the alignment into the
 * bit-plane layout, allocates workspaces.  `aln` may be NULL for a distance-table-only context
 * (run_apples.py -d).  Replaces prepareTree + ReducedReference construction state that
 * PoolQueryWorker.set_class_attributes injects (apples/PoolQueryWorker.py:17-24). */
int apples_ctx_create(const apples_tree *tree, const apples_alignment *aln, const apples_params *params,
                      int device, apples_ctx **out);
void apples_ctx_destroy(apples_ctx *ctx);
/* Message for the last non-zero status on this context (or creation failure when ctx == NULL). */
const char *apples_last_error(const apples_ctx *ctx);

/* Change the per-query options without rebuilding the context (tree/alignment stay resident). */
int apples_set_params(apples_ctx *ctx, const apples_params *params);

/* Seam B2, element kernel: dist_function(query, row) for every row (apples/Reference.py:22-25).
 * queries: [n_queries * L] bytes.  out_counts (nullable): [n_queries * n_rows * 2] uint32 pairs
 * (mismatches, valid) -- jc69's two integers (apples/distance.py:733-737); for scoredist the
 * pair is (0, valid).  out_dist (nullable): [n_queries * n_rows] fp64, -1.0 = missing. */
int apples_distances(apples_ctx *ctx, const uint8_t *queries, int64_t n_queries,
                     uint32_t *out_counts, double *out_dist);

/* Seam B1, alignment input: runquery(name, seq, None) for a block of queries
 * (apples/PoolQueryWorker.py:28-141 = get_obs_dist -> Subtree -> dp_frag -> placement_per_edge
 * -> placement).  self_row[q] = reference row whose name equals the query's name when that
 * name is a tree leaf (the entry runquery deletes, PoolQueryWorker.py:63-66), else -1;
 * NULL = none. */
int apples_place_from_sequences(apples_ctx *ctx, const uint8_t *queries, int64_t n_queries,
                                const int32_t *self_row, apples_placement *out);

/* Seam B1, distance-table input (run_apples.py -d): runquery(name, None, obs_dist).
 * dist: [n_queries * n_cols] fp64 in the table's column order, <0 = missing
 * (apples/PoolQueryWorker.py:44-59); col_node[c] = tree leaf of column c or -1 when the name
 * is not in the tree; self_col as self_row above. */
int apples_place_from_distances(apples_ctx *ctx, const double *dist, int64_t n_queries, int64_t n_cols,
                                const int32_t *col_node, const int32_t *self_col, apples_placement *out);

/* Seam B3 for inspection/parity: per-edge results of the sweep for ONE observed set.
 * obs_node/obs_dist: n_obs observed leaves (tree node ids) and their distances.  Outputs are
 * indexed by edge_index over [0, n_nodes): valid (0/1), S[6], R[6] (tuple order as in
 * apples/OLS.py:27-33, FM.py:21-27, BE.py:11-17, BME.py:11-17), x[4] = x_1,x_2,x_1_neg,x_2_neg
 * (apples/util.py:51-54), err = error_per_edge.  Any output pointer may be NULL. */
int apples_sweep_edges(apples_ctx *ctx, const int32_t *obs_node, const double *obs_dist, int32_t n_obs,
                       uint8_t *valid, double *S, double *R, double *x, double *err, int32_t *lca,
                       apples_placement *out);

/* Seam B1 with the placements left on the device: same work as apples_place_from_sequences (the
 * caller's host buffer is uploaded and packed chunk by chunk on a second stream while the previous
 * chunk's kernels run), but the result stays in a block owned by the context; fetch it with
 * apples_fetch_placements, hand it to a collective through apples_placements_device_ptr, release it
 * with apples_queries_free.  apples_place_from_sequences = this + fetch + free. */
int apples_place_sequences_streamed(apples_ctx *ctx, const uint8_t *queries, int64_t n_queries,
                                    const int32_t *self_row, int64_t *handle);

/* ---- device-resident variants used by bench.py (inputs already in HBM when timing starts) ---- */
/* Upload and pack a block of queries; returns a handle owned by the context. */
int apples_queries_upload(apples_ctx *ctx, const uint8_t *queries, int64_t n_queries,
                          const int32_t *self_row, int64_t *handle);
int apples_queries_free(apples_ctx *ctx, int64_t handle);
/* Upload a distance table (run_apples.py -d input, as apples_place_from_distances takes it) and
 * keep it resident; apples_place_resident / apples_fetch_placements then work on the handle. */
int apples_table_upload(apples_ctx *ctx, const double *dist, int64_t n_queries, int64_t n_cols,
                        const int32_t *col_node, const int32_t *self_col, int64_t *handle);
/* One pass of the hot path over an uploaded block; placements stay on the device until fetched. */
int apples_place_resident(apples_ctx *ctx, int64_t handle);
int apples_fetch_placements(apples_ctx *ctx, int64_t handle, apples_placement *out);
/* Device address of the block's placement array (n_queries apples_placement structs), for
 * zero-copy hand-off to a collective (the end-of-run RCCL gather that replaces starmap's pickle
 * return, run_apples.py:101-102). */
int apples_placements_device_ptr(apples_ctx *ctx, int64_t handle, void **ptr);
/* Distance kernel alone over an uploaded block (roofline measurement); results stay on device. */
int apples_distances_resident(apples_ctx *ctx, int64_t handle, int32_t query_tile);

/* Per-kernel device time of the most recent apples_place_* / apples_distances* call, measured
 * with HIP events on the context's stream.  ms[APPLES_T_*]; n = number of entries filled. */
enum { APPLES_T_PACK = 0, APPLES_T_DIST = 1, APPLES_T_SELECT = 2, APPLES_T_SWEEP = 3, APPLES_T_TOTAL = 4,
       APPLES_T_DIST_LAUNCHES = 5,
       APPLES_T_FILTER = 6, /* scoredist: the part of APPLES_T_DIST spent in the matrix-core lower-bound filter, before the
                               exact evaluation of its candidates, summed over the device batches the filter ran in; 0 = no
                               filter ran (every other route) */
       APPLES_T_BLOCKS = 7, /* clustered route with clade blocks: k_blocks_up (it runs on a side stream beside the selection's last
                               phase, so APPLES_T_SELECT covers it too; k_blocks_down is part of APPLES_T_SWEEP); 0 otherwise */
       APPLES_T_COUNT = 8 };
int apples_last_timing(const apples_ctx *ctx, double *ms, int32_t n);

/* Introspection: device name, packed layout, workspace sizes (JSON text, owned by ctx).  Which route a context takes is in here:
 * "sweep_layout" (lean / merge / bits / map / scan), "fused_distance_pass" (fp4 gemm / fp4 mfma / valu), "code_planes",
 * "cluster_fused", "cluster_blocks", "max_children", "height", "exotic_symbols_as_gaps" / "eight_plane_copy" (bytes beyond ACGT-),
 * "batch" (queries per device batch), "ragged_rows" / "row_small" / "big_rows" / "big_rows_last_batch" / "full_rows_for_good" (the
 * clustered route's observation rows: DESIGN.md section 3). */
const char *apples_describe(apples_ctx *ctx);

/* Backbone branch lengths on a fixed topology, no context needed: replaces the external call
 * `FastTree -nosupport -nome -noml -intree tree [-nt] < ref.fa` (apples/reestimateBackbone.py:82-84) --
 * balanced minimum-evolution lengths from log-corrected profile distances (nucleotide: -3/4 ln(1 - 4d/3);
 * protein: BLOSUM45 dissimilarity, -1.3 ln(1 - d)).  The tree as parent[] (-1 at the root) and a CSR of
 * children; every internal node has two children, the root two or three (resolve_polytomies first, as
 * reestimateBackbone.py:40-46 does).  leaf_row[v] = row of `rows` ([n_rows x length], the FASTA's own bytes:
 * either case, U = T, anything outside the alphabet a gap) for leaves, ignored otherwise.  out_len[v] = the
 * branch above v (0 at the root; a two-child root: both children carry the length of the one edge between
 * them).  Negative estimates are kept, as FastTree prints them.  site_chunk = 0 sizes the site chunk from free
 * memory (tests pass a multiple of 64).  Errors: apples_last_error(NULL). */
int apples_backbone_lengths(int device, int32_t n_nodes, const int32_t *parent, const int32_t *child_off,
                            const int32_t *child_idx, const int32_t *leaf_row, const uint8_t *rows,
                            int64_t n_rows, int32_t length, int protein, int64_t site_chunk, double *out_len);

/* Diagnostic, no context needed: out[i] = the logarithm the distance kernels take (csrc/libm_log.h: GNU libm's double log
 * restated bit for bit -- the `np.log` of apples/distance.py:715,745 as the C oracle evaluates it), x[i] positive and normal.
 * tests/test_gpu_parity.py compares it with the host's libm on millions of arguments.  Errors: apples_last_error(NULL). */
int apples_device_log(int device, const double *x, int64_t n, double *out);

#ifdef __cplusplus
}
#endif
#endif /* APPLES_HIP_H */
