/* Host-side input scanning for the APPLES hot path's callers: plain C ABI, no
 * device code.  Built as apples_amd/libapples_io.so by `python -m apples_amd.build`.
 *
 * apples_fasta_scan replaces the per-record Python of the reference's reader
 * (apples/fasta2dic.py:4-39 readfq, :42-72 fasta2dic) on the way to the dense N x L byte matrix the
 * distance kernels take: one pass over the file image, sequences written straight into their rows
 * through the 256-entry byte translation the caller derived from the alphabet flags
 * (fasta2dic.py:56-67).  Reader semantics kept: a record starts at a line whose first byte is '>'
 * or '@'; its name is the header up to the first space; sequence lines run until a line starting
 * with '@', '+' or '>'; after '+' a FASTQ quality block of at least the sequence's length is
 * skipped; every line loses its last character (so an unterminated final line loses a base);
 * "\r\n" and lone "\r" end a line (the reference opens the file in text mode).
 */
#ifndef APPLES_IO_H
#define APPLES_IO_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* Pass 1 (rows == NULL): counts records, reports the first record's sequence length in *length.
 * Pass 2 (rows != NULL, n_rows x length bytes): fills rows and name_off/name_len (byte range of
 * every name inside `data`).  Returns 0, or 1 if a record's length differs from *length
 * (*bad_record = its index, *bad_length = its length), or 2 if n_rows is too small. */
int apples_fasta_scan(const uint8_t *data, int64_t n_bytes, const uint8_t *translate /*[256]*/,
                      uint8_t *rows, int64_t n_rows, int64_t *n_records, int64_t *length,
                      int64_t *name_off, int32_t *name_len, int64_t *bad_record, int64_t *bad_length);

/* The same in one threaded pass for the plain shape (FASTA only, "\n" line ends: no '\r' anywhere, no line starting with '@' or
 * '+'): headers indexed by byte ranges, records filled by record ranges, memchr per line.  Returns 3 when the input is not of that
 * shape (the caller then runs apples_fasta_scan); otherwise as apples_fasta_scan.  n_threads <= 0: one per hardware thread. */
int apples_fasta_scan_mt(const uint8_t *data, int64_t n_bytes, const uint8_t *translate /*[256]*/, uint8_t *rows, int64_t n_rows,
                         int64_t *n_records, int64_t *length, int64_t *name_off, int32_t *name_len, int64_t *bad_record,
                         int64_t *bad_length, int32_t n_threads);

/* apples_newick_scan replaces the token loop of the Newick reader (apples_amd/tree.py:parse_newick;
 * reader contract: SURVEY.md Appendix B, for apples/prepareTree.py:24-36 and apples/util.py:57-88)
 * for backbones of 10^5..10^6 leaves.  `text` is the tree string after the optional [&R]/[&U]
 * prefix, ASCII only.  Tokens: ( ) , : ; | '...' with '' for a quote | [...] comment, dropped |
 * any other run of characters, stripped of white space: a label, or after ':' a branch length.
 * Output, per node in creation (= pre-) order, arrays of `cap` entries (cap >= 1 + number of '('
 * and ',' in the text is always enough): parent (creation index, -1 for the root), depth, size
 * (nodes in the subtree), the label's byte range in `text` (label_len = -1: none; label_quoted:
 * the range is the inside of a quoted label, '' still doubled), the branch length
 * (length_state 0: none, 1: parsed into `length` -- plain decimal spellings only, on which strtod
 * and Python's float() agree, 2: the caller converts text[length_off : length_off + length_len]).
 * Returns 0 and *n_nodes, 1 on anything malformed (the caller's own parser then decides and words
 * the error), 2 if cap is too small. */
int apples_newick_scan(const uint8_t *text, int64_t n_bytes, int64_t cap, int32_t *parent, int32_t *depth,
                       int32_t *size, int64_t *label_off, int32_t *label_len, uint8_t *label_quoted,
                       double *length, uint8_t *length_state, int64_t *length_off, int32_t *length_len,
                       int64_t *n_nodes);

/* apples_max_clusters is the clustering sweep of apples_amd/treecluster.py:max_clusters (the
 * TreeCluster "max" method the reference runs as an external tool with -t 1.2*f,
 * apples/Reference.py:87-88; restated from the published algorithm, parity unpinned -- see that
 * module) on the tree's arrays: children in file order as CSR, edge lengths with 0 where the Newick
 * had none.  Output: the leaves (node ids) cluster by cluster in the order the sweep closes the
 * clusters, leaves of a cluster left to right (leaf_order[n_leaves]), and the clusters' end offsets
 * into it (cluster_end[n_leaves + 1], cluster_end[0] = 0).  Returns 0. */
int apples_max_clusters(int32_t n_nodes, const int32_t *child_off, const int32_t *child_idx,
                        const double *edge_len, int32_t root, double threshold, int32_t *leaf_order,
                        int32_t *cluster_end, int32_t *n_clusters);

/* Consensus rows of multi-member clusters (apples/PoolRepresentativeWorker.py:17-85): per column the most frequent symbol of
 * `alphabet` among the cluster's rows (member_row[member_off[c] .. member_off[c + 1]) of seqs[n_rows][L]), ties to the first in
 * alphabet order, other symbols not counted.  out = [n_clusters][L].  Returns 0. */
int apples_consensus(const uint8_t *seqs, int64_t L, const int32_t *member_row, const int64_t *member_off, int64_t n_clusters,
                     const uint8_t *alphabet, int32_t n_alpha, uint8_t *out, int32_t n_threads);

/* apples_dismat_scan replaces the per-value Python of the reference's distance-table reader
 * (run_apples.py:43-54 read_dismat) for tables of 10^8 and more values.  `data` is the file image.
 * Reader semantics kept: universal newlines; the header line is right-stripped and split on white
 * space and its first field dropped; every other line is stripped and split: first field the query
 * name, then one value per tag, surplus values ignored, a tag without a value stays -1 (absent from
 * the reference's dict, and negatives are dropped at apples/PoolQueryWorker.py:52 anyway).
 * Pass 1 (out == NULL): counts tags and rows.  Pass 2: fills out[n_rows][n_tags] and the byte ranges
 * of tags and names inside `data`.  Returns 0; 1 when some value is not a plain decimal spelling
 * (on which strtod and Python's float() agree: the caller then uses its own reader); 2 when a
 * capacity is too small.  Duplicate tag names are the caller's business (dict(zip(...)) semantics). */
int apples_dismat_scan(const uint8_t *data, int64_t n_bytes, double *out, int64_t n_tags_cap, int64_t n_rows_cap,
                       int64_t *n_tags, int64_t *n_rows, int64_t *tag_off, int32_t *tag_len, int64_t *name_off,
                       int32_t *name_len);

/* jplace placement rows as text (csrc_io/jplace_format.cpp): the bytes json.dumps(sort_keys=True, indent=4) gives for the
 * reference's per-query dicts {"n": [name], "p": [[edge, likelihood, 1, distal, pendant]]} (run_apples.py:106-118,
 * apples/PoolQueryWorker.py:36-37), numbers spelled as Python's repr spells them.  kind[i]: 0 = floats; 1 = [edge, 0, 1, 0, 0];
 * 2 = floats with the pendant as the int 0.  keep[i] == 0 rows are skipped; first != 0: the first row written opens the
 * "placements" list.  Names are copied verbatim between quotes (the caller checks that json.dumps would not escape them).
 * Returns the bytes written into out[cap], -1 if cap is too small (512 + name bytes per kept row is always enough). */
int64_t apples_jplace_rows(const uint8_t *names, const int64_t *name_off, const int32_t *name_len, int64_t n, const int32_t *edge,
                           const double *err, const double *distal, const double *pendant, const uint8_t *kind,
                           const uint8_t *keep, int first, char *out, int64_t cap);
/* The jplace tree string without its closing ';' (apples/jutil.py:22-96): Newick with "{edge_index}" after every node but the
 * root; nodes numbered in left-to-right post-order, children in file order (CSR), labels as byte ranges of `labels`
 * (label_len < 0: none).  Returns the bytes written, -1 if cap is too small (24 + 40 per node + the labels' bytes is enough),
 * -2 when a branch length is an integral value beyond 2^63 (the caller's own formatter prints those). */
int64_t apples_extended_newick(int32_t n_nodes, const int32_t *child_off, const int32_t *child_idx, int32_t root,
                               const double *edge_len, const uint8_t *has_len, const uint8_t *labels, const int64_t *label_off,
                               const int32_t *label_len, char *out, int64_t cap);
/* repr(float) of Python into out (at most 32 bytes); returns its length */
int64_t apples_format_double(double x, char *out);

#ifdef __cplusplus
}
#endif
#endif
