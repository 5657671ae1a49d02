// Backbone branch lengths on a fixed topology: what the reference gets from
// `FastTree -nosupport -nome -noml -intree` (apples/reestimateBackbone.py:82-84) -- balanced minimum-evolution
// lengths from log-corrected profile distances (algorithm restated in oracle/fasttree_me.py).
//
// Profiles are per-site and distances are sums over sites, so the alignment is walked in site chunks that keep
// (nodes + internal nodes) x (K+1) x chunk doubles in HBM whatever the tree size; every kernel streams whole
// planes (site-contiguous, coalesced) and is HBM-bound:
//   k_leaf     leaf byte -> weight plane + K frequency planes
//   k_mean     one tree level of "mean of two profiles" (bottom-up for subtree profiles, top-down for the
//              profiles of everything NOT below a node)
//   k_edge     one workgroup per branch: the six (leaf: three) pair sums of w1 w2 d and w1 w2, added to the
//              branch's accumulators
//   k_finish   log correction and the branch formula
#include <algorithm>
#include <cstring>
#include <vector>

#include "common.h"

namespace {

__constant__ double kB45[400] = {
#include "blosum45_table.inc"
};

struct CodeTab {
    int8_t code[256];
};

// plane layout of slot s: base = s * (K+1) * Lc; plane 0 = weight, plane 1+k = frequency of symbol k
template <int K>
__global__ void k_leaf(const uint8_t *__restrict__ rows, int64_t row_stride, const int32_t *__restrict__ leaf_slot,
                       const int32_t *__restrict__ leaf_row, int site0, int Lc, int n_sites, CodeTab tab,
                       double *__restrict__ prof) {
    int i = blockIdx.x;
    int t = blockIdx.y * blockDim.x + threadIdx.x;
    if (t >= Lc) return;
    double *p = prof + (int64_t)leaf_slot[i] * (K + 1) * Lc;
    int c = -1;
    if (t < n_sites) c = tab.code[rows[(int64_t)leaf_row[i] * row_stride + site0 + t]];
    p[t] = c >= 0 ? 1.0 : 0.0;
#pragma unroll
    for (int k = 0; k < K; k++) p[(int64_t)(k + 1) * Lc + t] = (c == k) ? 1.0 : 0.0;
}

template <int K>
__global__ void k_mean(const int4 *__restrict__ items, int Lc, double *__restrict__ prof) {
    int4 it = items[blockIdx.x];  // x = out slot, y / z = source slots
    int t = blockIdx.y * blockDim.x + threadIdx.x;
    if (t >= Lc) return;
    const double *a = prof + (int64_t)it.y * (K + 1) * Lc;
    const double *b = prof + (int64_t)it.z * (K + 1) * Lc;
    double *o = prof + (int64_t)it.x * (K + 1) * Lc;
    double wa = 0.5 * a[t], wb = 0.5 * b[t];
    double w = wa + wb;
    o[t] = w;
#pragma unroll
    for (int k = 0; k < K; k++) {
        double f = a[(int64_t)(k + 1) * Lc + t] * wa + b[(int64_t)(k + 1) * Lc + t] * wb;
        o[(int64_t)(k + 1) * Lc + t] = w > 0 ? f / w : f;
    }
}

template <int K>
__device__ inline void load_prof(const double *__restrict__ p, int Lc, int t, double &w, double (&f)[K]) {
    w = p[t];
#pragma unroll
    for (int k = 0; k < K; k++) f[k] = p[(int64_t)(k + 1) * Lc + t];
}

template <int K>
__device__ inline double dissim(const double (&f1)[K], const double (&f2)[K], const double (&Df2)[K]) {
    double s = 0;
    if (K == 4) {
#pragma unroll
        for (int k = 0; k < K; k++) s += f1[k] * f2[k];
        return 1.0 - s;
    }
#pragma unroll
    for (int k = 0; k < K; k++) s += f1[k] * Df2[k];
    return s;
}

template <int K>
__device__ inline void times_D(const double (&f)[K], double (&Df)[K]) {
    if (K == 4) return;
#pragma unroll
    for (int a = 0; a < K; a++) {
        double s = 0;
#pragma unroll
        for (int b = 0; b < K; b++) s += kB45[a * 20 + b] * f[b];
        Df[a] = s;
    }
}

// quartet of branch e: x = A1 (or the leaf A), y = A2 (-1 for a leaf), z = B, w = C (slots); x < 0: nothing to do.
// acc[e][12]: (num, den) of the pairs A1B A1C A2B A2C A1A2 BC (leaf: AB AC - - - BC)
template <int K>
__global__ __launch_bounds__(256) void k_edge(const int4 *__restrict__ quartets, int Lc, const double *__restrict__ prof,
                                              double *__restrict__ acc) {
    int e = blockIdx.x;
    int4 q = quartets[e];
    if (q.x < 0) return;
    const double *pa1 = prof + (int64_t)q.x * (K + 1) * Lc;
    const double *pa2 = q.y >= 0 ? prof + (int64_t)q.y * (K + 1) * Lc : nullptr;
    const double *pb = prof + (int64_t)q.z * (K + 1) * Lc;
    const double *pc = prof + (int64_t)q.w * (K + 1) * Lc;
    double s[12];
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = 0;
    for (int t = threadIdx.x; t < Lc; t += blockDim.x) {
        double wa1, wa2 = 0, wb, wc;
        double fa1[K], fa2[K], fb[K], fc[K], Db[K], Dc[K], Da2[K];
        load_prof<K>(pa1, Lc, t, wa1, fa1);
        load_prof<K>(pb, Lc, t, wb, fb);
        load_prof<K>(pc, Lc, t, wc, fc);
        times_D<K>(fb, Db);
        times_D<K>(fc, Dc);
        double ww;
        ww = wa1 * wb; s[0] += ww * dissim<K>(fa1, fb, Db); s[1] += ww;
        ww = wa1 * wc; s[2] += ww * dissim<K>(fa1, fc, Dc); s[3] += ww;
        ww = wb * wc;  s[10] += ww * dissim<K>(fb, fc, Dc); s[11] += ww;
        if (pa2) {
            load_prof<K>(pa2, Lc, t, wa2, fa2);
            times_D<K>(fa2, Da2);
            ww = wa2 * wb; s[4] += ww * dissim<K>(fa2, fb, Db); s[5] += ww;
            ww = wa2 * wc; s[6] += ww * dissim<K>(fa2, fc, Dc); s[7] += ww;
            ww = wa1 * wa2; s[8] += ww * dissim<K>(fa1, fa2, Da2); s[9] += ww;
        }
    }
    __shared__ double red[4][12];
#pragma unroll
    for (int i = 0; i < 12; i++) {
        double v = s[i];
        for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][i] = v;
    }
    __syncthreads();
    if (threadIdx.x < 12) {
        double v = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
        acc[(int64_t)e * 12 + threadIdx.x] += v;
    }
}

__device__ inline double corrected(double num, double den, int protein) {
    if (!(den > 0)) return 3.0;  // no site in common
    double d = num / den;
    // FastTree's LogCorrect: 3.0 once the raw distance reaches 0.74 (nt) / 0.99 (aa), never more than 3.0
    double c = protein ? (d < 0.99 ? -1.3 * log(1.0 - d) : 3.0) : (d < 0.74 ? -0.75 * log(1.0 - 4.0 * d / 3.0) : 3.0);
    return fmin(c, 3.0);
}

__global__ void k_finish(const int4 *__restrict__ quartets, const double *__restrict__ acc, int n, int protein,
                         double *__restrict__ out) {
    int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    int4 q = quartets[e];
    if (q.x < 0) {
        out[e] = 0.0;
        return;
    }
    const double *a = acc + (int64_t)e * 12;
    double d0 = corrected(a[0], a[1], protein), d1 = corrected(a[2], a[3], protein), dbc = corrected(a[10], a[11], protein);
    if (q.y < 0) {
        out[e] = (d0 + d1 - dbc) / 2;
        return;
    }
    double d2 = corrected(a[4], a[5], protein), d3 = corrected(a[6], a[7], protein), daa = corrected(a[8], a[9], protein);
    out[e] = (d0 + d1 + d2 + d3) / 4 - (daa + dbc) / 2;
}

struct Bufs {
    uint8_t *rows = nullptr;
    int32_t *leaf_slot = nullptr, *leaf_row = nullptr;
    int4 *items = nullptr, *quartets = nullptr;
    double *prof = nullptr, *acc = nullptr, *out = nullptr;
    ~Bufs() {
        (void)hipFree(rows); (void)hipFree(leaf_slot); (void)hipFree(leaf_row); (void)hipFree(items);
        (void)hipFree(quartets); (void)hipFree(prof); (void)hipFree(acc); (void)hipFree(out);
    }
};

int fail(const std::string &msg) {
    g_create_error = msg;
    return 1;
}

#define ME_TRY(call)                                                                   \
    do {                                                                               \
        hipError_t e__ = (call);                                                       \
        if (e__ != hipSuccess) return fail(std::string(#call) + ": " + hipGetErrorString(e__)); \
    } while (0)

template <int K>
int run(int n, const std::vector<int32_t> &leaf_slot, const std::vector<int32_t> &leaf_row_v, const std::vector<int4> &items,
        const std::vector<std::pair<int, int>> &level_ranges, const std::vector<int4> &quartets, int n_slots,
        const uint8_t *rows, int64_t n_rows, int L, int protein, int64_t site_chunk, double *out_len) {
    Bufs b;
    int n_leaves = (int)leaf_slot.size();
    size_t fr = 0, tot = 0;
    ME_TRY(hipMemGetInfo(&fr, &tot));
    int64_t fixed = n_rows * (int64_t)L + (int64_t)n * (12 * 8 + 8 + 16) + (int64_t)items.size() * 16 + (1 << 20);
    int64_t per_site = (int64_t)n_slots * (K + 1) * 8;
    int64_t Lc = site_chunk > 0 ? site_chunk : ((int64_t)(fr * 0.6) - fixed) / per_site;
    Lc = std::min<int64_t>(Lc / 64 * 64, ((int64_t)L + 63) / 64 * 64);
    if (Lc < 64) return fail("backbone lengths: not enough device memory for one 64-site chunk of profiles (" +
                             std::to_string(fr >> 20) + " MiB free)");
    ME_TRY(hipMalloc((void **)&b.rows, std::max<int64_t>(1, n_rows * (int64_t)L)));
    ME_TRY(hipMemcpy(b.rows, rows, n_rows * (int64_t)L, hipMemcpyHostToDevice));
    ME_TRY(hipMalloc((void **)&b.leaf_slot, n_leaves * 4));
    ME_TRY(hipMalloc((void **)&b.leaf_row, n_leaves * 4));
    ME_TRY(hipMemcpy(b.leaf_slot, leaf_slot.data(), n_leaves * 4, hipMemcpyHostToDevice));
    ME_TRY(hipMemcpy(b.leaf_row, leaf_row_v.data(), n_leaves * 4, hipMemcpyHostToDevice));
    ME_TRY(hipMalloc((void **)&b.items, std::max<size_t>(1, items.size()) * 16));
    ME_TRY(hipMemcpy(b.items, items.data(), items.size() * 16, hipMemcpyHostToDevice));
    ME_TRY(hipMalloc((void **)&b.quartets, (size_t)n * 16));
    ME_TRY(hipMemcpy(b.quartets, quartets.data(), (size_t)n * 16, hipMemcpyHostToDevice));
    ME_TRY(hipMalloc((void **)&b.acc, (size_t)n * 12 * 8));
    ME_TRY(hipMemset(b.acc, 0, (size_t)n * 12 * 8));
    ME_TRY(hipMalloc((void **)&b.out, (size_t)n * 8));
    ME_TRY(hipMalloc((void **)&b.prof, (size_t)(per_site * Lc)));
    CodeTab tab;
    memset(tab.code, -1, sizeof(tab.code));
    const char *alpha = protein ? "ARNDCQEGHILKMFPSTWYV" : "ACGT";
    for (int i = 0; alpha[i]; i++) {
        tab.code[(unsigned char)alpha[i]] = (int8_t)i;
        tab.code[(unsigned char)(alpha[i] | 0x20)] = (int8_t)i;
    }
    if (!protein) tab.code['U'] = tab.code['u'] = 3;
    for (int64_t s0 = 0; s0 < L; s0 += Lc) {
        int ns = (int)std::min<int64_t>(Lc, L - s0);
        int lc = (ns + 63) / 64 * 64;  // planes of the last chunk shrink with it
        dim3 blk(256);
        int gx = (lc + 255) / 256;
        k_leaf<K><<<dim3(n_leaves, gx), blk>>>(b.rows, L, b.leaf_slot, b.leaf_row, (int)s0, lc, ns, tab, b.prof);
        for (auto &r : level_ranges)
            if (r.second > r.first)
                k_mean<K><<<dim3(r.second - r.first, gx), blk>>>(b.items + r.first, lc, b.prof);
        k_edge<K><<<n, 256>>>(b.quartets, lc, b.prof, b.acc);
        ME_TRY(hipGetLastError());
    }
    k_finish<<<(n + 255) / 256, 256>>>(b.quartets, b.acc, n, protein, b.out);
    ME_TRY(hipGetLastError());
    ME_TRY(hipMemcpy(out_len, b.out, (size_t)n * 8, hipMemcpyDeviceToHost));
    return 0;
}

}  // namespace

extern "C" int apples_backbone_lengths(int device, int32_t n_nodes, const int32_t *parent, const int32_t *child_off,
                                       const int32_t *child_idx, const int32_t *leaf_row, const uint8_t *rows,
                                       int64_t n_rows, int32_t length, int protein, int64_t site_chunk, double *out_len) {
    g_create_error.clear();
    if (n_nodes < 4 || !parent || !child_off || !child_idx || !leaf_row || !rows || !out_len || length <= 0)
        return fail("backbone lengths: a tree of at least three leaves, its arrays, the alignment and an output buffer are required");
    if (site_chunk < 0 || site_chunk % 64) return fail("backbone lengths: site_chunk must be 0 (automatic) or a multiple of 64");
    int n = n_nodes, root = -1;
    for (int v = 0; v < n; v++)
        if (parent[v] < 0) {
            if (root >= 0) return fail("backbone lengths: more than one root");
            root = v;
        }
    if (root < 0) return fail("backbone lengths: no root");
    auto nk = [&](int v) { return child_off[v + 1] - child_off[v]; };
    auto kid = [&](int v, int i) { return child_idx[child_off[v] + i]; };
    for (int v = 0; v < n; v++) {
        int k = nk(v);
        if (v == root ? (k != 2 && k != 3) : (k != 0 && k != 2))
            return fail("backbone lengths: node " + std::to_string(v) + " has " + std::to_string(k) +
                        " children (resolve polytomies and suppress unifurcations first; the root may have two or three)");
        if (k == 0 && (leaf_row[v] < 0 || leaf_row[v] >= n_rows))
            return fail("backbone lengths: leaf node " + std::to_string(v) + " has no alignment row");
        for (int i = 0; i < k; i++)
            if (kid(v, i) < 0 || kid(v, i) >= n || parent[kid(v, i)] != v) return fail("backbone lengths: child lists and parents disagree");
    }
    // depth (parents first) and height (children first) orders
    std::vector<int> order;
    order.reserve(n);
    order.push_back(root);
    std::vector<int> depth(n, 0), height(n, 0);
    for (size_t i = 0; i < order.size(); i++) {
        int v = order[i];
        for (int j = 0; j < nk(v); j++) {
            depth[kid(v, j)] = depth[v] + 1;
            order.push_back(kid(v, j));
        }
    }
    if ((int)order.size() != n) return fail("backbone lengths: the tree is not connected");
    int max_h = 0, max_d = 0;
    for (int i = n - 1; i >= 0; i--) {
        int v = order[i];
        if (parent[v] >= 0) height[parent[v]] = std::max(height[parent[v]], height[v] + 1);
        max_h = std::max(max_h, height[v]);
        max_d = std::max(max_d, depth[v]);
    }
    bool binroot = nk(root) == 2;
    if (binroot && nk(kid(root, 0)) == 0 && nk(kid(root, 1)) == 0) return fail("backbone lengths: fewer than three leaves");
    // slots: subtree profile of node v = v; "everything else" profile of an internal node = n + its internal rank
    std::vector<int> up_slot(n, -1);
    int n_slots = n;
    for (int v = 0; v < n; v++)
        if (v != root && nk(v)) up_slot[v] = n_slots++;
    if (binroot) {  // the other side of the root edge is the other child's subtree
        up_slot[kid(root, 0)] = kid(root, 1);
        up_slot[kid(root, 1)] = kid(root, 0);
        n_slots = n;
        for (int v = 0; v < n; v++)
            if (v != root && nk(v) && parent[v] != root) up_slot[v] = n_slots++;
    }
    std::vector<int32_t> leaf_slot, leaf_row_v;
    for (int v = 0; v < n; v++)
        if (!nk(v)) {
            leaf_slot.push_back(v);
            leaf_row_v.push_back(leaf_row[v]);
        }
    // items by level: subtree profiles bottom-up (height 1..), then the complements top-down (depth 1..)
    std::vector<std::vector<int4>> down(max_h + 1), upl(max_d + 1);
    for (int v = 0; v < n; v++) {
        if (nk(v) == 2 && !(v == root)) down[height[v]].push_back(make_int4(v, kid(v, 0), kid(v, 1), 0));
        if (v == root || !nk(v)) continue;
        int p = parent[v];
        if (p == root) {
            if (binroot) continue;
            int o[2], c = 0;
            for (int j = 0; j < 3; j++)
                if (kid(root, j) != v) o[c++] = kid(root, j);
            upl[1].push_back(make_int4(up_slot[v], o[0], o[1], 0));
        } else {
            int sib = kid(p, 0) == v ? kid(p, 1) : kid(p, 0);
            upl[depth[v]].push_back(make_int4(up_slot[v], sib, up_slot[p], 0));
        }
    }
    std::vector<int4> items;
    std::vector<std::pair<int, int>> ranges;
    for (auto *lv : {&down, &upl})
        for (auto &l : *lv) {
            if (l.empty()) continue;
            ranges.emplace_back((int)items.size(), (int)(items.size() + l.size()));
            items.insert(items.end(), l.begin(), l.end());
        }
    // quartets
    std::vector<int4> quartets(n, make_int4(-1, -1, -1, -1));
    std::vector<int> copy_from(n, -1);
    for (int v = 0; v < n; v++) {
        if (v == root) continue;
        int p = parent[v], B, C;
        if (p == root) {
            if (binroot) {
                int s = kid(root, 0) == v ? kid(root, 1) : kid(root, 0);
                if (!nk(s)) {  // the sibling leaf's own entry is this edge
                    copy_from[v] = s;
                    continue;
                }
                B = kid(s, 0);
                C = kid(s, 1);
            } else {
                int o[2], c = 0;
                for (int j = 0; j < 3; j++)
                    if (kid(root, j) != v) o[c++] = kid(root, j);
                B = o[0];
                C = o[1];
            }
        } else {
            B = kid(p, 0) == v ? kid(p, 1) : kid(p, 0);
            C = up_slot[p];
        }
        quartets[v] = nk(v) ? make_int4(kid(v, 0), kid(v, 1), B, C) : make_int4(v, -1, B, C);
    }
    if (hipSetDevice(device) != hipSuccess) return fail("backbone lengths: hipSetDevice(" + std::to_string(device) + ") failed");
    int rc = protein ? run<20>(n, leaf_slot, leaf_row_v, items, ranges, quartets, n_slots, rows, n_rows, length, 1, site_chunk, out_len)
                     : run<4>(n, leaf_slot, leaf_row_v, items, ranges, quartets, n_slots, rows, n_rows, length, 0, site_chunk, out_len);
    if (rc) return rc;
    for (int v = 0; v < n; v++)
        if (copy_from[v] >= 0) out_len[v] = out_len[copy_from[v]];
    out_len[root] = 0.0;
    return 0;
}
