/*
 * oracle.c -- scalar C restatement of the APPLES per-query hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library;
 * the product (apples_amd + libapples_hip.so) never does.  It follows the reference one query
 * at a time, in the reference's operation order (file:line cited per function, relative to
 * /root/reference), so that it can check the HIP path bit for bit at sizes the Python oracle
 * (oracle/apples_oracle.py) is too slow for.  Pinned by tests/test_oracle_c.py against the
 * Python oracle, which is itself pinned against the reference's golden fixtures.
 *
 * `x ** 2` in the reference is libm pow(x, 2.0) (CPython float_pow); the same call is used here.
 * np.log in jc69 is numpy's SIMD log; callers pass the distance table the host fills with numpy
 * (apples_amd.engine.jc69_lut) so distances carry the same bits; without it libm log is used.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

enum { OLS = 0, FM = 1, BME = 2, BE = 3 };
enum { MLSE = 0, ME = 1, HYBRID = 2 };

typedef struct {
    int32_t n_nodes;
    const int32_t *parent;
    const double *edge_len;
    const int32_t *child_off;
    const int32_t *child_idx;
    const int32_t *level;
} otree;

typedef struct {
    int32_t edge;
    uint32_t flags; /* same bits as APPLES_F_* in include/apples_hip.h */
    double error, distal, pendant;
    int32_t n_obs, n_valid;
} oplacement;

#define F_EXACT 1u
#define F_INSUFFICIENT 2u
#define F_MISPLACED 4u
#define F_PENDANT_INT 8u
#define F_ZERO_NOT_IN_TREE 16u
#define F_DEGENERATE 32u

/* ---- distances ------------------------------------------------------------------------------ */
/* apples/distance.py:733-737 */
void orc_pair_counts(const uint8_t *a, const uint8_t *b, int L, uint32_t *mism, uint32_t *valid) {
    uint32_t m = 0, v = 0;
    for (int i = 0; i < L; ++i) {
        int nd = (a[i] != '-') & (b[i] != '-');
        v += nd;
        m += nd & (a[i] != b[i]);
    }
    *mism = m;
    *valid = v;
}

/* apples/distance.py:734-745 */
double orc_jc69_from_counts(uint32_t mism, uint32_t valid, int L, double overlap, const double *lut) {
    if (lut) return lut[(int64_t)valid * (valid + 1) / 2 + mism];
    if (!valid || (double)valid / (double)L < overlap) return -1.0;
    double p = mism * 1.0 / valid;
    if (p - 2.220446049250313e-16 < 0) return 0.0;
    double loc = 1 - (4 * p / 3);
    if (0 >= loc) return -1.0;
    return -0.75 * log(loc);
}

static int aa_index(uint8_t b) { /* apples/distance.py:418-678: ARNDCQEGHILKMFPSTWYV, else 0 */
    static const char *order = "ARNDCQEGHILKMFPSTWYV";
    if (b >= 'a' && b <= 'z') b -= 32;
    for (int i = 0; i < 20; ++i)
        if (order[i] == (char)b) return i;
    return 0;
}

/* apples/distance.py:681-715, sites summed left to right */
double orc_scoredist(const uint8_t *a, const uint8_t *b, int L, double overlap, const double *blosum400) {
    uint32_t v = 0;
    double tot = 0.0;
    for (int i = 0; i < L; ++i) {
        int nd = (a[i] != '-') & (b[i] != '-');
        v += nd;
        if (nd) tot += blosum400[20 * aa_index(a[i]) + aa_index(b[i])];
    }
    if (!v || (double)v / (double)L < overlap) return -1.0;
    double r = 1 - tot / v;
    if (0 >= r) return -1.0;
    return -log(r) * 1.3;
}

/* libm's log over an array: what the device's restatement of it (apples_amd/csrc/libm_log.h) is compared with */
void orc_log_array(const double *x, double *out, int64_t n) {
    for (int64_t i = 0; i < n; ++i) out[i] = log(x[i]);
}

/* one query against n_rows rows -> dist[n_rows] */
void orc_distance_row(const uint8_t *query, const uint8_t *rows, int64_t n_rows, int L, int model, double overlap,
                      const double *lut, const double *blosum400, double *dist) {
    for (int64_t r = 0; r < n_rows; ++r) {
        if (model == 1) dist[r] = orc_scoredist(query, rows + r * L, L, overlap, blosum400);
        else {
            uint32_t m, v;
            orc_pair_counts(query, rows + r * L, L, &m, &v);
            dist[r] = orc_jc69_from_counts(m, v, L, overlap, lut);
        }
    }
}

/* ---- observed set --------------------------------------------------------------------------- */
typedef struct { double d; int32_t i; } okey_t;
static int key_cmp(const void *x, const void *y) {
    const okey_t *a = x, *b = y;
    if (a->d < b->d) return -1;
    if (a->d > b->d) return 1;
    return (a->i > b->i) - (a->i < b->i);
}

/* apples/Reference.py:138-154 (alignment input, table = 0: dist is indexed by row, clusters given)
 * or apples/PoolQueryWorker.py:44-59 (table = 1: dist indexed by column, every column its own
 * cluster, columns whose node is -1 ignored).  Output: the dict in insertion order as
 * (row/column index, distance) pairs; returns their number. */
int64_t orc_select(const double *dist, int64_t n_refs, int64_t n_reps, const int32_t *rep_row,
                   const int32_t *member_off, const int32_t *member_row, const int32_t *row_node, int table,
                   double thr, int baseobs, int32_t *out_row, double *out_dist) {
    okey_t *keys = malloc(sizeof(okey_t) * (size_t)(n_reps > 0 ? n_reps : 1));
    int64_t nk = 0;
    for (int64_t j = 0; j < n_reps; ++j) {
        int64_t r = rep_row ? rep_row[j] : j;
        if (table && row_node[r] < 0) continue;
        double d = dist[r];
        if (d >= 0) { keys[nk].d = d; keys[nk].i = (int32_t)j; ++nk; }
    }
    qsort(keys, (size_t)nk, sizeof(okey_t), key_cmp);
    int64_t n = 0;
    int obs_num = 0;
    for (int64_t k = 0; k < nk; ++k) {
        if (keys[k].d <= thr || obs_num < baseobs) {
            int32_t j = keys[k].i;
            int m0 = member_off ? member_off[j] : j, m1 = member_off ? member_off[j + 1] : j + 1;
            for (int m = m0; m < m1; ++m) {
                int32_t r = member_row ? member_row[m] : m;
                double dm = dist[r];
                if (!(dm < 0)) { out_row[n] = r; out_dist[n] = dm; ++n; ++obs_num; }
            }
        } else break;
    }
    free(keys);
    return n;
}

/* ---- least squares -------------------------------------------------------------------------- */
static void leaf_tuple(int m, double D, double *t) {
    t[0] = 1; t[1] = 0; t[2] = 0; t[3] = 0;
    if (m == OLS || m == BME) { t[4] = D * D; t[5] = D; }        /* apples/OLS.py:27-33, BME.py:11-17 */
    else if (m == FM) { t[4] = 1.0 / D; t[5] = 1.0 / (D * D); }  /* apples/FM.py:21-27 */
    else { t[4] = D; t[5] = 1.0 / D; }                           /* apples/BE.py:11-17 */
}

static void lift(int m, const double *s, double e, double *t) {
    if (m == OLS || m == BME) { /* apples/OLS.py:36-44 */
        t[0] = s[0]; t[1] = s[0] * e + s[1]; t[2] = s[0] * e * e + s[2] + 2 * e * s[1];
        t[3] = e * s[5] + s[3]; t[4] = s[4]; t[5] = s[5];
    } else if (m == FM) { /* apples/FM.py:31-40 */
        t[0] = s[0]; t[1] = e * s[4] + s[1]; t[2] = e * s[5] + s[2];
        t[3] = s[5] * e * e + s[3] + 2 * e * s[2]; t[4] = s[4]; t[5] = s[5];
    } else { /* apples/BE.py:20-30 */
        t[0] = s[0]; t[1] = s[0] * e + s[1]; t[2] = e * s[5] + s[2];
        t[3] = s[5] * e * e + s[3] + 2 * e * s[2]; t[4] = s[4]; t[5] = s[5];
    }
}

typedef struct { double x1, x2, x1n, x2n, err; int x1_int; } sol;

static sol solve_edge(int m, const double *S, const double *R, double e, int negative) {
    int ols = (m == OLS || m == BME);
    int IA = ols ? 0 : 5, IC = ols ? 5 : (m == FM ? 4 : 0), IE = ols ? 0 : 5, ID = ols ? 1 : 2;
    double a11 = R[IA] + S[IA], a12 = R[IA] - S[IA], a21 = a12, a22 = a11;
    double c1 = R[IC] + S[IC] - e * S[IE] - R[ID] - S[ID];
    double c2 = R[IC] - S[IC] + e * S[IE] - R[ID] + S[ID];
    double det = 1 / (a11 * a22 - a12 * a21); /* apples/util.py:26-50 */
    sol r;
    r.x1n = (a22 * c1 - a12 * c2) * det;
    r.x2n = (-a21 * c1 + a11 * c2) * det;
    r.x1 = r.x1n; r.x2 = r.x2n; r.x1_int = 0;
    if (!negative) {
        double x1n = r.x1n, x2n = r.x2n;
        if (x1n < 0 && x2n < 0) { r.x1 = 0; r.x1_int = 1; r.x2 = 0; }
        else if (x1n > 0 && x2n < 0) {
            double t = c1 * 1.0 / a11;
            if (0 > t) { r.x1 = 0; r.x1_int = 1; } else r.x1 = t;
            r.x2 = 0;
        } else if (x1n < 0 && 0 <= x2n && x2n <= e) {
            r.x1 = 0; r.x1_int = 1;
            double u = c2 * 1.0 / a22;
            if (0 > u) u = 0;
            r.x2 = (e < u) ? e : u;
        } else if (x1n < 0 && x2n > e) { r.x1 = 0; r.x1_int = 1; r.x2 = e; }
        else if (x1n > 0 && x2n > e) {
            double t = (c1 * 1.0 - a12 * e) / a11;
            if (0 > t) { r.x1 = 0; r.x1_int = 1; } else r.x1 = t;
            r.x2 = e;
        }
    }
    /* error_per_edge: apples/OLS.py:121-128, FM.py:117-124, BE.py:73-80, BME.py:76-83 */
    int JA = (m == FM) ? 0 : 4, JB = ols ? 1 : 2, JC = ols ? 0 : 5, JD = ols ? 5 : (m == FM ? 4 : 0);
    int JE = ols ? 3 : 1, JF = ols ? 2 : 3;
    double up = r.x1 + r.x2, dn = e + r.x1 - r.x2;
    double A = R[JA] + S[JA];
    double B = 2 * up * R[JB] + 2 * dn * S[JB];
    double C = pow(up, 2.0) * R[JC] + pow(dn, 2.0) * S[JC];
    double D = -2 * up * R[JD] - 2 * dn * S[JD];
    double E = -2 * R[JE] - 2 * S[JE];
    double F = R[JF] + S[JF];
    r.err = A + B + C + D + E + F;
    return r;
}

/* binary heap of (-level, node): apples/PrioritySet.py */
typedef struct { int32_t lvl, node; } hent;
static void heap_push(hent *h, int *n, hent e) {
    int i = (*n)++;
    h[i] = e;
    while (i > 0) {
        int p = (i - 1) / 2;
        if (h[p].lvl > h[i].lvl || (h[p].lvl == h[i].lvl && h[p].node <= h[i].node)) break;
        hent t = h[p]; h[p] = h[i]; h[i] = t; i = p;
    }
}
static hent heap_pop(hent *h, int *n) {
    hent top = h[0];
    h[0] = h[--(*n)];
    int i = 0;
    for (;;) {
        int l = 2 * i + 1, r = l + 1, b = i;
        if (l < *n && (h[l].lvl > h[b].lvl || (h[l].lvl == h[b].lvl && h[l].node < h[b].node))) b = l;
        if (r < *n && (h[r].lvl > h[b].lvl || (h[r].lvl == h[b].lvl && h[r].node < h[b].node))) b = r;
        if (b == i) break;
        hent t = h[b]; h[b] = h[i]; h[i] = t; i = b;
    }
    return top;
}

static int cmp_i32(const void *a, const void *b) { return (*(const int32_t *)a > *(const int32_t *)b) - (*(const int32_t *)a < *(const int32_t *)b); }

typedef struct { /* per-thread scratch sized by the tree */
    uint8_t *valid; int32_t *inset; hent *heap; int32_t *vlist; int32_t *pos; double *S, *R, *leafD, *xe;
} scratch;

static scratch *scratch_new(int32_t n) {
    scratch *s = calloc(1, sizeof(scratch));
    s->valid = calloc((size_t)n, 1); s->inset = calloc((size_t)n, 4); s->heap = malloc(sizeof(hent) * (size_t)n);
    s->vlist = malloc(4 * (size_t)n); s->pos = malloc(4 * (size_t)n); s->S = malloc(48 * (size_t)n);
    s->R = malloc(48 * (size_t)n); s->leafD = malloc(8 * (size_t)n); s->xe = malloc(40 * (size_t)n);
    return s;
}
static void scratch_free(scratch *s) {
    free(s->valid); free(s->inset); free(s->heap); free(s->vlist); free(s->pos); free(s->S); free(s->R);
    free(s->leafD); free(s->xe); free(s);
}

/* apples/PoolQueryWorker.py:101-133 for one observed set (tree leaves only).  Optional per-edge
 * outputs indexed by edge_index (any may be NULL). */
static void place_observed(const otree *t, scratch *sc, const int32_t *obs_node, const double *obs_dist, int n_obs,
                           int method, int criterion, int negative, oplacement *out, uint8_t *o_valid, double *o_S,
                           double *o_R, double *o_x, double *o_err, int32_t *o_lca) {
    /* Subtree.validate_edges, apples/Subtree.py:23-43 */
    int hn = 0, nv = 0;
    for (int i = 0; i < n_obs; ++i) {
        int v = obs_node[i];
        sc->leafD[v] = obs_dist[i];
        if (!sc->inset[v]) { hent e = {t->level[v], v}; heap_push(sc->heap, &hn, e); sc->inset[v] = 1; }
    }
    while (hn > 1) {
        hent x = heap_pop(sc->heap, &hn);
        sc->inset[x.node] = 0;
        sc->valid[x.node] = 1;
        sc->vlist[nv++] = x.node;
        int p = t->parent[x.node];
        if (!sc->inset[p]) { hent e = {t->level[p], p}; heap_push(sc->heap, &hn, e); sc->inset[p] = 1; }
    }
    int lca = sc->heap[0].node;
    sc->inset[lca] = 0;
    qsort(sc->vlist, (size_t)nv, 4, cmp_i32); /* valid post-order == ascending edge_index */
    for (int k = 0; k < nv; ++k) sc->pos[sc->vlist[k]] = k;
    /* all_S_values */
    for (int k = 0; k < nv; ++k) {
        int v = sc->vlist[k];
        double *acc = sc->S + 6 * (size_t)k;
        int c0 = t->child_off[v], c1 = t->child_off[v + 1];
        if (c0 == c1) { leaf_tuple(method, sc->leafD[v], acc); continue; }
        for (int x = 0; x < 6; ++x) acc[x] = 0;
        double coef = 1.0;
        if (method == BME) { int n = 0; for (int ci = c0; ci < c1; ++ci) n += sc->valid[t->child_idx[ci]]; coef = 1.0 / n; }
        for (int ci = c0; ci < c1; ++ci) {
            int c = t->child_idx[ci];
            if (!sc->valid[c]) continue;
            double tt[6];
            lift(method, sc->S + 6 * (size_t)sc->pos[c], t->edge_len[c], tt);
            for (int x = 0; x < 6; ++x) acc[x] += (method == BME) ? coef * tt[x] : tt[x];
        }
    }
    /* all_R_values (parents before children = descending id), placement_per_edge, error_per_edge */
    for (int k = nv - 1; k >= 0; --k) {
        int v = sc->vlist[k], p = t->parent[v];
        double *acc = sc->R + 6 * (size_t)k;
        for (int x = 0; x < 6; ++x) acc[x] = 0;
        int c0 = t->child_off[p], c1 = t->child_off[p + 1];
        double coef = 1.0;
        if (method == BME) {
            int n = (p != lca) ? 1 : 0;
            for (int ci = c0; ci < c1; ++ci) { int c = t->child_idx[ci]; n += (c != v) && sc->valid[c]; }
            coef = 1.0 / n;
        }
        for (int ci = c0; ci < c1; ++ci) {
            int c = t->child_idx[ci];
            if (c == v || !sc->valid[c]) continue;
            double tt[6];
            lift(method, sc->S + 6 * (size_t)sc->pos[c], t->edge_len[c], tt);
            for (int x = 0; x < 6; ++x) acc[x] += (method == BME) ? coef * tt[x] : tt[x];
        }
        if (p != lca && sc->valid[p]) {
            double tt[6];
            lift(method, sc->R + 6 * (size_t)sc->pos[p], t->edge_len[p], tt);
            for (int x = 0; x < 6; ++x) acc[x] += (method == BME) ? coef * tt[x] : tt[x];
        }
    }
    for (int k = 0; k < nv; ++k) {
        int v = sc->vlist[k];
        sol r = solve_edge(method, sc->S + 6 * (size_t)k, sc->R + 6 * (size_t)k, t->edge_len[v], negative);
        double *xe = sc->xe + 5 * (size_t)k;
        xe[0] = r.x1; xe[1] = r.x2; xe[2] = r.x1n; xe[3] = r.x2n; xe[4] = r.err;
    }
    /* Algorithm.placement, apples/Algorithm.py:74-91 */
    int best = -1;
    if (criterion == HYBRID) {
        int kk = 0;
        while ((1 << (kk + 1)) <= nv) ++kk; /* floor(log2(num_nodes)) */
        double last_e = -INFINITY; int last_k = -1; double bx = INFINITY;
        for (int r = 0; r < kk; ++r) {
            int bk = -1;
            for (int k = 0; k < nv; ++k) {
                double e = sc->xe[5 * (size_t)k + 4];
                int after = (e > last_e) || (e == last_e && k > last_k);
                if (after && (bk < 0 || e < sc->xe[5 * (size_t)bk + 4])) bk = k;
            }
            if (bk < 0) break;
            last_e = sc->xe[5 * (size_t)bk + 4]; last_k = bk;
            if (best < 0 || sc->xe[5 * (size_t)bk] < bx) { bx = sc->xe[5 * (size_t)bk]; best = bk; }
        }
    } else {
        int off = (criterion == ME) ? 0 : 4;
        for (int k = 0; k < nv; ++k)
            if (best < 0 || sc->xe[5 * (size_t)k + off] < sc->xe[5 * (size_t)best + off]) best = k;
    }
    out->n_valid = nv;
    if (best < 0) { out->edge = -1; out->flags = F_DEGENERATE | F_PENDANT_INT; }
    else {
        int v = sc->vlist[best];
        double e = t->edge_len[v];
        sol r = solve_edge(method, sc->S + 6 * (size_t)best, sc->R + 6 * (size_t)best, e, negative);
        out->edge = v; out->error = r.err; out->distal = e - r.x2; out->pendant = r.x1; out->flags = 0;
        if (r.x1_int) out->flags |= F_PENDANT_INT;
        if (r.x1 == 0 && r.err > 0 && (r.x2 == 0 || r.x2 == e)) out->flags |= F_MISPLACED;
    }
    if (o_lca) *o_lca = lca;
    for (int k = 0; k < nv; ++k) {
        int v = sc->vlist[k];
        if (o_valid) o_valid[v] = 1;
        if (o_S) memcpy(o_S + 6 * (size_t)v, sc->S + 6 * (size_t)k, 48);
        if (o_R) memcpy(o_R + 6 * (size_t)v, sc->R + 6 * (size_t)k, 48);
        if (o_x) memcpy(o_x + 4 * (size_t)v, sc->xe + 5 * (size_t)k, 32);
        if (o_err) o_err[v] = sc->xe[5 * (size_t)k + 4];
        sc->valid[v] = 0; /* unroll_changes, apples/Subtree.py:72-76 */
    }
}

/* runquery after the observed dict exists (apples/PoolQueryWorker.py:63-133).  sel_row/sel_dist:
 * the dict in insertion order; row_node maps rows/columns to tree leaves (-1 = not in tree);
 * self_row = the query's own entry or -1. */
static void run_observed(const otree *t, scratch *sc, const int32_t *sel_row, const double *sel_dist, int64_t n_sel,
                         const int32_t *row_node, int32_t self_row, int method, int criterion, int negative,
                         oplacement *out, int32_t *tmp_node, double *tmp_dist) {
    memset(out, 0, sizeof *out);
    int n_total = 0, n_tree = 0;
    for (int64_t k = 0; k < n_sel; ++k) {
        if (sel_row[k] == self_row) continue;
        ++n_total;
        if (sel_dist[k] == 0) { /* first zero in dict order, :72-75 */
            int nd = row_node[sel_row[k]];
            out->flags = F_EXACT | F_PENDANT_INT;
            out->edge = nd;
            if (nd < 0) { out->flags |= F_ZERO_NOT_IN_TREE; out->edge = -1; }
            for (int64_t j = k + 1; j < n_sel; ++j) n_total += sel_row[j] != self_row;
            out->n_obs = n_total;
            return;
        }
        if (row_node[sel_row[k]] >= 0) { tmp_node[n_tree] = row_node[sel_row[k]]; tmp_dist[n_tree] = sel_dist[k]; ++n_tree; }
    }
    out->n_obs = n_total;
    if (n_total <= 2) { out->flags = F_INSUFFICIENT | F_PENDANT_INT; out->edge = -1; return; }
    if (n_tree < 2) { out->flags = F_DEGENERATE | F_PENDANT_INT; out->edge = -1; return; }
    place_observed(t, sc, tmp_node, tmp_dist, n_tree, method, criterion, negative, out, 0, 0, 0, 0, 0, 0);
    out->n_obs = n_total;
}

/* ---- exported drivers ------------------------------------------------------------------------ */
/* per-edge inspection, mirrors apples_sweep_edges */
int orc_sweep_edges(const otree *t, const int32_t *obs_node, const double *obs_dist, int n_obs, int method,
                    int criterion, int negative, uint8_t *valid, double *S, double *R, double *x, double *err,
                    int32_t *lca, oplacement *out) {
    scratch *sc = scratch_new(t->n_nodes);
    oplacement tmp;
    memset(&tmp, 0, sizeof tmp);
    if (valid) memset(valid, 0, (size_t)t->n_nodes);
    place_observed(t, sc, obs_node, obs_dist, n_obs, method, criterion, negative, &tmp, valid, S, R, x, err, lca);
    tmp.n_obs = n_obs;
    if (out) *out = tmp;
    scratch_free(sc);
    return 0;
}

/* alignment input: mirrors apples_place_from_sequences.  OpenMP over queries when built with it. */
int orc_place_from_sequences(const otree *t, const uint8_t *rows, int64_t n_rows, int64_t n_refs, int L,
                             const int32_t *row_node, int64_t n_reps, const int32_t *rep_row,
                             const int32_t *member_off, const int32_t *member_row, int model, int method,
                             int criterion, int negative, double thr, int baseobs, double overlap, const double *lut,
                             const double *blosum400, const uint8_t *queries, int64_t nq, const int32_t *self_row,
                             oplacement *out, int threads) {
    if (!rep_row) n_reps = n_refs;
#pragma omp parallel num_threads(threads > 0 ? threads : 1)
    {
        scratch *sc = scratch_new(t->n_nodes);
        double *dist = malloc(8 * (size_t)n_rows);
        int32_t *sel_row = malloc(4 * (size_t)n_refs), *tmp_node = malloc(4 * (size_t)n_refs);
        double *sel_dist = malloc(8 * (size_t)n_refs), *tmp_dist = malloc(8 * (size_t)n_refs);
#pragma omp for schedule(dynamic, 1)
        for (int64_t q = 0; q < nq; ++q) {
            orc_distance_row(queries + q * L, rows, n_rows, L, model, overlap, lut, blosum400, dist);
            int64_t n = orc_select(dist, n_refs, n_reps, rep_row, member_off, member_row, row_node, 0, thr, baseobs,
                                   sel_row, sel_dist);
            run_observed(t, sc, sel_row, sel_dist, n, row_node, self_row ? self_row[q] : -1, method, criterion,
                         negative, &out[q], tmp_node, tmp_dist);
        }
        free(dist); free(sel_row); free(tmp_node); free(sel_dist); free(tmp_dist);
        scratch_free(sc);
    }
    return 0;
}

/* distance-table input: mirrors apples_place_from_distances */
int orc_place_from_distances(const otree *t, const double *dist, int64_t nq, int64_t n_cols, const int32_t *col_node,
                             const int32_t *self_col, int method, int criterion, int negative, double thr,
                             int baseobs, oplacement *out, int threads) {
#pragma omp parallel num_threads(threads > 0 ? threads : 1)
    {
        scratch *sc = scratch_new(t->n_nodes);
        int32_t *sel_row = malloc(4 * (size_t)n_cols), *tmp_node = malloc(4 * (size_t)n_cols);
        double *sel_dist = malloc(8 * (size_t)n_cols), *tmp_dist = malloc(8 * (size_t)n_cols);
#pragma omp for schedule(dynamic, 1)
        for (int64_t q = 0; q < nq; ++q) {
            int64_t n = orc_select(dist + q * n_cols, n_cols, n_cols, 0, 0, 0, col_node, 1, thr, baseobs, sel_row, sel_dist);
            run_observed(t, sc, sel_row, sel_dist, n, col_node, self_col ? self_col[q] : -1, method, criterion,
                         negative, &out[q], tmp_node, tmp_dist);
        }
        free(sel_row); free(tmp_node); free(sel_dist); free(tmp_dist);
        scratch_free(sc);
    }
    return 0;
}
