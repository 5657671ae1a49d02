// C ABI of libapples_hip.so: context, uploads, batch driver, timing.  See include/apples_hip.h.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>

#include "common.h"

thread_local std::string g_create_error;

namespace {

template <typename T>
int dev_alloc(apples_ctx *ctx, T **p, int64_t n) {
    *p = nullptr;
    if (n <= 0) n = 1;
    hipError_t e = hipMalloc((void **)p, (size_t)n * sizeof(T));
    if (e != hipSuccess && !ctx->blk_cache.empty()) {  // out of memory: give back the cached block buffers and try once more
        (void)hipGetLastError();
        for (auto &c : ctx->blk_cache) (void)hipFree(c.second);
        ctx->blk_cache.clear();
        e = hipMalloc((void **)p, (size_t)n * sizeof(T));
    }
    if (e != hipSuccess) {
        size_t fr = 0, tot = 0;
        (void)hipMemGetInfo(&fr, &tot);
        ctx->err = std::string("hipMalloc of ") + std::to_string((size_t)n * sizeof(T)) + " bytes: " + hipGetErrorString(e) +
                   " (" + std::to_string(fr >> 20) + " MiB free of " + std::to_string(tot >> 20) + ")";
        return 1;
    }
    return 0;
}

template <typename T>
int dev_upload(apples_ctx *ctx, T **p, const T *h, int64_t n) {
    if (dev_alloc(ctx, p, n)) return 1;
    if (n > 0) HIP_TRY(ctx, hipMemcpyAsync(*p, h, (size_t)n * sizeof(T), hipMemcpyHostToDevice, ctx->stream));
    return 0;
}

void dev_free(void *p) {
    if (p) (void)hipFree(p);
}

int64_t round_up(int64_t x, int64_t m) { return (x + m - 1) / m * m; }

// Buffers of query blocks come from a small cache that lives with the context: a host-buffer call
// (apples_place_from_sequences) would otherwise pay hipMalloc + hipFree of several hundred MB each time.
template <typename T>
int blk_alloc(apples_ctx *ctx, T **p, int64_t n) {
    const size_t bytes = (size_t)std::max<int64_t>(n, 1) * sizeof(T);
    size_t best = ctx->blk_cache.size();
    for (size_t i = 0; i < ctx->blk_cache.size(); ++i) {
        const size_t have = ctx->blk_cache[i].first;
        if (have >= bytes && have <= 2 * bytes + 4096 && (best == ctx->blk_cache.size() || have < ctx->blk_cache[best].first)) best = i;
    }
    if (best < ctx->blk_cache.size()) {
        *p = (T *)ctx->blk_cache[best].second;
        ctx->blk_size[*p] = ctx->blk_cache[best].first;
        ctx->blk_cache.erase(ctx->blk_cache.begin() + best);
        return 0;
    }
    if (dev_alloc(ctx, p, n)) {  // out of memory: drop the cache and try once more
        for (auto &c : ctx->blk_cache) dev_free(c.second);
        ctx->blk_cache.clear();
        if (dev_alloc(ctx, p, n)) return 1;
    }
    ctx->blk_size[*p] = bytes;
    return 0;
}

void blk_free(apples_ctx *ctx, void *p) {
    if (!p) return;
    auto it = ctx->blk_size.find(p);
    if (it == ctx->blk_size.end()) { dev_free(p); return; }
    const size_t bytes = it->second;
    ctx->blk_size.erase(it);
    // (the caller has drained every stream that used the buffer: the entry points synchronise before they free a block)
    ctx->blk_cache.emplace_back(bytes, p);
    size_t held = 0;
    for (auto &c : ctx->blk_cache) held += c.first;
    while (!ctx->blk_cache.empty() && (ctx->blk_cache.size() > 24 || held > ((size_t)6 << 30))) {  // bounded by bytes: the oldest go
        held -= ctx->blk_cache.front().first;
        dev_free(ctx->blk_cache.front().second);
        ctx->blk_cache.erase(ctx->blk_cache.begin());
    }
}

// BLOSUM45-derived dissimilarities (FastTree2's table, the data apples/distance.py:12-415 reads);
// supplied by the host in params? No: it is part of the algorithm, so it is compiled in via
// blosum45_table.inc, generated from apples_amd/data/blosum45_dist.txt by the build script.
const double kBlosum45[400] = {
#include "blosum45_table.inc"
};

int upload_tree(apples_ctx *ctx, const apples_tree *t) {
    DevTree &d = ctx->tree;
    d.dbg = ctx->dbg;
    d.force_poly = (int32_t)knob(ctx, "APPLES_LEAN_FORCE_POLY", 0);
    d.lean_small = ctx->params.criterion != APPLES_HYBRID || !(ctx->dbg & APPLES_DBG_HYBRID_RECORDS);
    d.n_nodes = t->n_nodes;
    int h = 0;
    for (int i = 0; i < t->n_nodes; ++i) h = std::max(h, t->level[i]);
    d.height = h;
    d.max_children = 0;
    d.poly_kids = 0;
    for (int i = 0; i < t->n_nodes; ++i) {
        const int nc = t->child_off[i + 1] - t->child_off[i];
        d.max_children = std::max(d.max_children, nc);
        if (nc > 2) d.poly_kids += nc;
    }
    if (dev_upload(ctx, &d.parent, t->parent, t->n_nodes)) return 1;
    if (dev_upload(ctx, &d.edge_len, t->edge_len, t->n_nodes)) return 1;
    if (dev_upload(ctx, &d.child_off, t->child_off, t->n_nodes + 1)) return 1;
    if (dev_upload(ctx, &d.child_idx, t->child_idx, std::max(t->n_nodes - 1, 1))) return 1;
    if (dev_upload(ctx, &d.level, t->level, t->n_nodes)) return 1;
    // level-ordered bit space: level by level, internal nodes then leaves, each block in node-id
    // order and aligned to a 64-bit word
    std::vector<std::vector<int32_t>> lv_int(h + 1), lv_leaf(h + 1);
    for (int i = 0; i < t->n_nodes; ++i)
        (t->child_off[i + 1] > t->child_off[i] ? lv_int : lv_leaf)[t->level[i]].push_back(i);
    std::vector<int32_t> lvlw(2 * (h + 1) + 1), lpos(t->n_nodes);
    int32_t words = 0;
    for (int l = 0; l <= h; ++l) {
        lvlw[2 * l] = words;
        for (size_t k = 0; k < lv_int[l].size(); ++k) lpos[lv_int[l][k]] = words * 64 + (int32_t)k;
        words += (int32_t)((lv_int[l].size() + 63) / 64);
        lvlw[2 * l + 1] = words;
        for (size_t k = 0; k < lv_leaf[l].size(); ++k) lpos[lv_leaf[l][k]] = words * 64 + (int32_t)k;
        words += (int32_t)((lv_leaf[l].size() + 63) / 64);
    }
    lvlw[2 * (h + 1)] = words;
    d.bm_words = words;
    std::vector<int32_t> lnode((size_t)std::max(words, 1) * 64, -1);
    for (int i = 0; i < t->n_nodes; ++i) lnode[lpos[i]] = i;
    if (dev_upload(ctx, &d.lvlw, lvlw.data(), (int64_t)lvlw.size())) return 1;
    if (dev_upload(ctx, &d.lnode, lnode.data(), (int64_t)lnode.size())) return 1;
    std::vector<NodeRec> rec(t->n_nodes);
    for (int i = 0; i < t->n_nodes; ++i) {
        NodeRec &r = rec[i];
        r.parent = t->parent[i];
        r.nchild = t->child_off[i + 1] - t->child_off[i];
        r.c0 = r.nchild > 0 ? t->child_idx[t->child_off[i]] : -1;
        r.c1 = r.nchild > 1 ? t->child_idx[t->child_off[i] + 1] : -1;
        r.lpos = lpos[i];
        r.ppos = t->parent[i] >= 0 ? lpos[t->parent[i]] : -1;
        r.e = t->edge_len[i];
        r.c0pos = r.c0 >= 0 ? lpos[r.c0] : -1;
        r.c1pos = r.c1 >= 0 ? lpos[r.c1] : -1;
        r.e0 = r.c0 >= 0 ? t->edge_len[r.c0] : 0.0;
        r.e1 = r.c1 >= 0 ? t->edge_len[r.c1] : 0.0;
        r.kleaf = 0;
        if (r.c0 >= 0 && t->child_off[r.c0 + 1] == t->child_off[r.c0]) r.kleaf |= 1u;
        if (r.c1 >= 0 && t->child_off[r.c1 + 1] == t->child_off[r.c1]) r.kleaf |= 2u;
        r.node = i;
    }
    {
        struct PE { int32_t parent, pad; double e; };
        static_assert(sizeof(PE) == 16, "pe record");
        std::vector<PE> pe(t->n_nodes);
        for (int i = 0; i < t->n_nodes; ++i) pe[i] = PE{t->parent[i], 0, t->edge_len[i]};
        PE *dpe = nullptr;
        if (dev_upload(ctx, &dpe, pe.data(), t->n_nodes)) return 1;
        d.pe = dpe;
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));  // (pe goes out of scope)
    }
    std::vector<int32_t> npos((size_t)t->n_nodes * 2);
    for (int i = 0; i < t->n_nodes; ++i) { npos[2 * (size_t)i] = rec[i].lpos; npos[2 * (size_t)i + 1] = rec[i].ppos; }
    if (dev_upload(ctx, &d.npos, npos.data(), (int64_t)npos.size())) return 1;
    if (dev_upload(ctx, &d.rec, rec.data(), t->n_nodes)) return 1;
    {
        std::vector<NodeRec> rec_l((size_t)std::max(words, 1) * 64);
        std::memset(rec_l.data(), 0xff, rec_l.size() * sizeof(NodeRec));
        for (int i = 0; i < t->n_nodes; ++i) rec_l[lpos[i]] = rec[i];
        if (dev_upload(ctx, &d.rec_l, rec_l.data(), (int64_t)rec_l.size())) return 1;
    }
    // Scan formulation of the sweep (sweep_scan.hip), an alternative kept behind APPLES_SWEEP_SCAN=1: same
    // bytes out, no atomics and no node map, but slower than the level loop on MI355X as measured.
    // Per leaf, by level, the ancestor's edge length and node id, and an Euler-tour range-minimum table for
    // lowest common ancestors.  Needs post-order node ids (a subtree = a contiguous id range ending at its
    // root), at most 254 levels and tables of a sensible size.
    d.scan = false;
    {
        // left-to-right post-order ids (a subtree = a contiguous id range ending at its root, children tiling it from the
        // left in file order): what the merged level lists of the sweep rely on.  apples_amd/tree.py numbers that way
        // (apples/util.py:57-69); a C-ABI caller with another numbering gets the node map / node bits instead.
        const int n = t->n_nodes;
        std::vector<int64_t> size(n, 1);
        bool postorder = true;
        for (int i = 0; i < n - 1 && postorder; ++i) {
            const int p = t->parent[i];
            if (p <= i || p >= n) postorder = false; else size[p] += size[i];
        }
        if (postorder && (t->parent[n - 1] != -1 || size[n - 1] != n)) postorder = false;
        for (int i = 0; i < n && postorder; ++i) {
            int64_t at = i - size[i] + 1;
            for (int c = t->child_off[i]; c < t->child_off[i + 1]; ++c) {
                const int k = t->child_idx[c];
                if (k < 0 || k >= n || k - size[k] + 1 != at) { postorder = false; break; }
                at = k + 1;
            }
            if (postorder && t->child_off[i + 1] > t->child_off[i] && at != i) postorder = false;
        }
        d.merge_ok = postorder;
    }
    if ((ctx->dbg & APPLES_DBG_SWEEP_SCAN) && h <= 254) {
        const int n = t->n_nodes;
        std::vector<int64_t> size(n, 1);
        bool postorder = true;
        for (int i = 0; i < n - 1 && postorder; ++i) {
            const int p = t->parent[i];
            if (p <= i || p >= n) postorder = false; else size[p] += size[i];
        }
        if (postorder && (t->parent[n - 1] != -1 || size[n - 1] != n)) postorder = false;
        for (int i = 0; i < n && postorder; ++i) {  // children tile the parent's range from the left, in file order
            int64_t at = i - size[i] + 1;
            for (int c = t->child_off[i]; c < t->child_off[i + 1]; ++c) {
                const int k = t->child_idx[c];
                if (k - size[k] + 1 != at) { postorder = false; break; }
                at = k + 1;
            }
            if (postorder && t->child_off[i + 1] > t->child_off[i] && at != i) postorder = false;
        }
        int64_t total = 0;
        for (int i = 0; i < n; ++i)
            if (t->child_off[i + 1] == t->child_off[i]) total += t->level[i] + 1;
        const int64_t max_entries = (int64_t)knob(ctx, "APPLES_SCAN_TABLE_MAX", ((int64_t)200 << 20));
        if (postorder && total <= max_entries) {
            std::vector<int4> info(n);
            std::vector<AncRec> anc((size_t)total);
            int64_t at = 0;
            for (int i = 0; i < n; ++i) {
                info[i] = make_int4(-1, t->level[i], -1, 0);
                if (t->child_off[i + 1] != t->child_off[i]) continue;
                info[i].x = (int32_t)at;
                int v = i;
                for (int m = t->level[i]; m >= 0; --m) {
                    if (v < 0 || t->level[v] != m) { postorder = false; break; }  // (level must be the depth)
                    anc[(size_t)at + m].e = t->edge_len[v];
                    anc[(size_t)at + m].node = v;
                    anc[(size_t)at + m].pad = 0;
                    v = t->parent[v];
                }
                at += t->level[i] + 1;
            }
            // Euler tour of levels (a node's level on entry and again after each child) and its sparse table
            std::vector<uint8_t> euler;
            euler.reserve((size_t)2 * n);
            {
                std::vector<std::pair<int, int>> stack;  // (node, next child)
                stack.emplace_back(n - 1, 0);
                while (!stack.empty() && postorder) {
                    auto &top = stack.back();
                    const int v = top.first;
                    if (top.second == 0 && t->child_off[v + 1] == t->child_off[v]) info[v].z = (int32_t)euler.size();
                    euler.push_back((uint8_t)t->level[v]);
                    const int c = t->child_off[v] + top.second;
                    if (c < t->child_off[v + 1]) { ++top.second; stack.emplace_back(t->child_idx[c], 0); }
                    else {
                        stack.pop_back();
                        // (the parent's level is appended when control returns to it: its loop iteration pushes it)
                    }
                }
            }
            const int64_t elen = (int64_t)euler.size();
            int K = 1;
            while (((int64_t)1 << K) <= elen) ++K;
            if (postorder && elen * K <= ((int64_t)1 << 31)) {
                std::vector<uint8_t> rmq((size_t)elen * K);
                std::copy(euler.begin(), euler.end(), rmq.begin());
                for (int k = 1; k < K; ++k) {
                    const uint8_t *prev = rmq.data() + (size_t)(k - 1) * elen;
                    uint8_t *cur = rmq.data() + (size_t)k * elen;
                    const int64_t half = (int64_t)1 << (k - 1);
                    for (int64_t x = 0; x < elen; ++x) cur[x] = std::min(prev[x], prev[std::min(x + half, elen - 1)]);
                }
                if (dev_upload(ctx, &d.leaf_info, info.data(), n)) return 1;
                if (dev_upload(ctx, &d.anc, anc.data(), total)) return 1;
                if (dev_upload(ctx, &d.rmq, rmq.data(), (int64_t)rmq.size())) return 1;
                HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
                d.euler_len = (int32_t)elen;
                d.rmq_k = K;
                d.scan = true;
            }
        }
    }
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

// sort keys so that deeper leaves come first; rows that are not tree leaves go last
// (scan formulation of the sweep: by node id alone -- the sweep wants its leaves in post-order)
std::vector<int32_t> level_order(const int32_t *node, int64_t n, const std::vector<int32_t> &level, bool by_id = false) {
    std::vector<int32_t> idx(n);
    std::iota(idx.begin(), idx.end(), 0);
    if (by_id) {
        std::stable_sort(idx.begin(), idx.end(), [&](int32_t a, int32_t b) {
            const int64_t ka = node[a] >= 0 ? node[a] : ((int64_t)1 << 40), kb = node[b] >= 0 ? node[b] : ((int64_t)1 << 40);
            return ka < kb;
        });
        return idx;
    }
    std::stable_sort(idx.begin(), idx.end(), [&](int32_t a, int32_t b) {
        int la = node[a] >= 0 ? level[node[a]] : -1;
        int lb = node[b] >= 0 ? level[node[b]] : -1;
        if (la != lb) return la > lb;
        // within a level by node id (= left to right in the tree): the sweep's per-level lists then
        // come out in tree order and parents read their children's records nearly sequentially
        return la >= 0 && node[a] < node[b];
    });
    return idx;
}

// Clade blocks of a clustered reference (DevAlign::blk_*): whole subtrees of the backbone whose leaves are all members of one
// cluster.  A query that accepts the cluster observes every leaf of the block (apples/Reference.py:146-152), so inside the block
// the induced subtree (apples/Subtree.py:23-43) is the block itself, the same for every such query: its sweep runs on a static
// schedule (sweep_lean.hip: k_blocks_up / k_blocks_down) and the per-query merged lists see the block's root as one leaf.
// Needs node ids in post-order (children before parents, a subtree = a contiguous id range).  Inside a block a node may have up
// to BLK_MAX_DEG children (round 6; binary until then: a polytomy cut its block into its children's).  A node of m > 2 children is a
// chain of m - 1 records, each right behind the subtree of its right operand: (c0, c1), then (the chain so far, c_j) with no edge on
// the left -- bottom-up that is the reference's sum in file order, ((u0 + u1) + u2) + ..., on the binary walk as it is (lifting the
// partial sum over a zero edge and adding it to 0 leaves its bits: the sums start at +0.0 and are never -0.0); the chain's last
// record is the node itself and also lists ALL its children (blk_pk_*) for the top-down walk, where every child's R is the sum
// over all its siblings in file order (k_blocks_down's polytomy step), the chain's other records being skipped there.
int build_blocks(apples_ctx *ctx, const apples_tree *t, const std::vector<int32_t> &level, const std::vector<int32_t> &slot_node,
                 const std::vector<int32_t> &slot_rep, const std::vector<int32_t> &slot_mpos, const std::vector<int32_t> &rep_moff) {
    DevAlign &a = ctx->aln;
    a.n_blocks = 0;
    a.n_e = 0;
    if (a.all_singleton || !ctx->tree.merge_ok || ctx->tree.scan || (ctx->dbg & APPLES_DBG_NO_BLOCKS)) return 0;
    const int n = t->n_nodes;
    std::vector<int32_t> node_slot(n, -1), pure(n, -1), nleaf(n, 0), first(n, 0), nn(n, 1);
    for (int64_t s = 0; s < a.n_refs; ++s)
        if (slot_node[s] >= 0) node_slot[slot_node[s]] = (int32_t)s;
    for (int v = 0; v < n; ++v) {
        const int c0 = t->child_off[v], c1 = t->child_off[v + 1];
        first[v] = v;
        if (c0 == c1) {
            pure[v] = node_slot[v] >= 0 ? slot_rep[node_slot[v]] : -1;
            nleaf[v] = 1;
            continue;
        }
        int p = -2, nl = 0, nv = 1;
        for (int k = c0; k < c1; ++k) {
            const int c = t->child_idx[k];
            if (c >= v) return 0;  // not a post-order numbering: no blocks
            first[v] = std::min(first[v], first[c]);
            nl += nleaf[c];
            nv += nn[c];
            if (pure[c] < 0) p = -1;
            else if (p == -2) p = pure[c];
            else if (p != pure[c]) p = -1;
        }
        nleaf[v] = nl;
        nn[v] = nv;
        pure[v] = (c1 - c0 >= 2 && c1 - c0 <= BLK_MAX_DEG && p >= 0 && v - first[v] + 1 == nv) ? p : -1;
    }
    std::vector<std::vector<int32_t>> by_rep((size_t)a.n_reps);
    int n_blocks = 0;
    for (int v = 0; v < n; ++v)
        if (pure[v] >= 0 && nleaf[v] >= 2 && (t->parent[v] < 0 || pure[t->parent[v]] < 0)) { by_rep[pure[v]].push_back(v); ++n_blocks; }
    // (the selection's bitmap over emission indices: 4 096 words of LDS at most, k_select_clusters)
    int64_t n_leaf_slots = 0;
    for (int64_t s = 0; s < a.n_refs; ++s) n_leaf_slots += slot_node[s] >= 0 ? 1 : 0;
    if (n_blocks == 0 || n_leaf_slots + n_blocks > SELECT_CLUSTERS_MAX_SLOTS || n >= BLK_NODE_MASK) return 0;
    std::vector<int4> rec_i;
    std::vector<double2> rec_e, rec_c;  // rec_c: BME's coefficients of the two operands (1 / children; 1 for a chain's partial sum)
    std::vector<int2> rec_p;           // per record: (first entry of pk_*, children) for a polytomy's last record, else (0, 0)
    std::vector<int2> pk_i;            // a polytomy's children in file order: (slot or -(member position + 1), node id) ...
    std::vector<double> pk_e;          // ... and their edge lengths
    std::vector<double> stat_plain, stat_bme;  // per record: the first three components of the node's S tuple (OLS / BE / FM; BME)
    std::vector<int32_t> rep_soff((size_t)a.n_reps + 1, 0), mem_block((size_t)rep_moff[a.n_reps], -1), blk_root, blk_rslot, blk_nodes, slot_of(n, -1), chain_slot(n, -1);
    for (int64_t c = 0; c < a.n_reps; ++c) {
        rep_soff[c] = (int32_t)rec_i.size();
        const int64_t rc = c;
        int sc = 0;
        for (int u : by_rep[c]) {
            const int b = (int)blk_root.size();
            bool first_leaf = true;
            // one record: (left operand, right operand) -> slot sc; `lpart`: the left operand is the chain's partial sum (no edge, no node)
            auto emit = [&](int l, bool lpart, int lslot, int r, int m, bool is_root, bool is_top, int self_node) {
                auto ref = [&](int k) { return slot_of[k] >= 0 ? slot_of[k] : -(slot_mpos[node_slot[k]] + 1); };
                const int here = sc++;
                const int lref = lpart ? lslot : ref(l);
                int w = r;
                if (is_root) w |= BLK_F_ROOT;
                if (is_top && m > 2) w |= BLK_F_POLY;
                if (!is_top) w |= BLK_F_PART;  // (a chain's inner record: not a node)
                rec_i.push_back(make_int4(lref, ref(r), lpart ? (int)BLK_F_NOEDGE : l, w));
                rec_e.push_back(make_double2(lpart ? 0.0 : t->edge_len[l], t->edge_len[r]));
                rec_c.push_back(make_double2(lpart ? 1.0 : 1.0 / (double)m, 1.0 / (double)m));
                rec_p.push_back(make_int2(0, 0));
                // the query-independent components (S, Sd, Sd2 of OLS / BME; sweep_math.h) with node_S's operations in node_S's order:
                // lift, then 0 + left + right; BME: each share times its coefficient.  (This file is compiled with -ffp-contract=off.)
                for (int bme = 0; bme < 2; ++bme) {
                    std::vector<double> &st = bme ? stat_bme : stat_plain;
                    double acc[3] = {0, 0, 0};
                    for (int k = 0; k < 2; ++k) {
                        const bool part = k == 0 && lpart;
                        const int cs = k == 0 ? (lpart ? lslot : slot_of[l]) : slot_of[r];
                        double s0 = 1, s1 = 0, s2 = 0;  // a leaf's tuple starts (1, 0, 0, ...) for every method
                        if (cs >= 0) { const double *q = &st[(size_t)(rep_soff[rc] + cs) * 3]; s0 = q[0]; s1 = q[1]; s2 = q[2]; }
                        const double e = part ? 0.0 : t->edge_len[k == 0 ? l : r];
                        double u[3];
                        u[0] = s0;
                        u[1] = s0 * e + s1;
                        u[2] = s0 * e * e + s2 + 2 * e * s1;
                        const double coef = part ? 1.0 : 1.0 / (double)m;
                        for (int x = 0; x < 3; ++x) acc[x] += bme ? coef * u[x] : u[x];
                    }
                    st.push_back(acc[0]); st.push_back(acc[1]); st.push_back(acc[2]);
                }
                (void)self_node;
                return here;
            };
            for (int v = first[u]; v <= u; ++v) {
                if (t->child_off[v] == t->child_off[v + 1]) {
                    mem_block[rep_moff[c] + slot_mpos[node_slot[v]]] = (b << 1) | (first_leaf ? 1 : 0);  // (one member speaks for the block where it is counted)
                    first_leaf = false;
                }
                // v (a leaf, or an internal node whose own last record went out when its last child was done) is complete: if it is
                // the j-th child (j >= 1) of its parent, the parent's record over (what came before, v) follows right here
                if (v == u) break;
                const int par = t->parent[v];
                const int pc0 = t->child_off[par], m = t->child_off[par + 1] - pc0;
                int j = 0;
                while (t->child_idx[pc0 + j] != v) ++j;
                if (j == 0) continue;
                const bool top = j == m - 1;
                const int here = emit(t->child_idx[pc0], j > 1, chain_slot[par], v, m, top && par == u, top, par);
                chain_slot[par] = here;
                if (top) {
                    slot_of[par] = here;
                    if (m > 2) {  // every child of the polytomy, for the top-down walk
                        rec_p.back() = make_int2((int)pk_i.size(), m);
                        for (int k = 0; k < m; ++k) {
                            const int ck = t->child_idx[pc0 + k];
                            pk_i.push_back(make_int2(slot_of[ck] >= 0 ? slot_of[ck] : -(slot_mpos[node_slot[ck]] + 1), ck));
                            pk_e.push_back(t->edge_len[ck]);
                        }
                    }
                }
            }
            blk_root.push_back(u);
            blk_rslot.push_back(slot_of[u]);
            blk_nodes.push_back(nn[u] - 1);
        }
    }
    rep_soff[a.n_reps] = (int32_t)rec_i.size();
    // per cluster: its blocks (numbered cluster by cluster: [rep_boff[c], rep_boff[c + 1])) and its members outside every block
    // (positions in the cluster's member list), for the selection's short form of the last phase (select.hip)
    std::vector<int32_t> rep_boff((size_t)a.n_reps + 1, 0), rep_loff((size_t)a.n_reps + 1, 0), loose_mp;
    for (int64_t c = 0; c < a.n_reps; ++c) {
        rep_boff[c + 1] = rep_boff[c] + (int32_t)by_rep[c].size();
        for (int m = rep_moff[c]; m < rep_moff[c + 1]; ++m)
            if (mem_block[m] < 0) loose_mp.push_back(m - rep_moff[c]);
        rep_loff[c + 1] = (int32_t)loose_mp.size();
    }
    if (loose_mp.empty()) loose_mp.push_back(0);
    // emission order: tree-leaf slots and block roots by (level, deepest first; node id)
    struct Ent { int32_t lvl, node, kind, idx; };
    std::vector<Ent> ents;
    for (int64_t s = 0; s < a.n_refs; ++s)
        if (slot_node[s] >= 0) ents.push_back({level[slot_node[s]], slot_node[s], 0, (int32_t)s});
    for (int b = 0; b < n_blocks; ++b) ents.push_back({level[blk_root[b]], blk_root[b], 1, b});
    std::sort(ents.begin(), ents.end(), [](const Ent &x, const Ent &y) { return x.lvl != y.lvl ? x.lvl > y.lvl : x.node < y.node; });
    const int H = ctx->tree.height;
    std::vector<int32_t> e_of_slot((size_t)a.n_refs, -1), e_of_blk((size_t)n_blocks, -1), e_node(ents.size()), lvl_e((size_t)H + 2, 0);
    for (size_t e = 0; e < ents.size(); ++e) {
        if (ents[e].lvl > H || ents[e].lvl < 0) return 0;
        (ents[e].kind ? e_of_blk[ents[e].idx] : e_of_slot[ents[e].idx]) = (int32_t)e;
        e_node[e] = ents[e].node;
        ++lvl_e[ents[e].lvl];  // for now: entries AT level l
    }
    for (int l = H, above = 0; l >= -1; --l) {  // entry l + 1 = entries above level l
        const int at = l >= 0 ? lvl_e[l] : 0;
        lvl_e[l + 1] = above;
        above += at;
    }
    if (dev_upload(ctx, &a.blk_rec_i, rec_i.data(), (int64_t)rec_i.size())) return 1;
    if (dev_upload(ctx, &a.blk_rec_e, rec_e.data(), (int64_t)rec_e.size())) return 1;
    if (dev_upload(ctx, &a.blk_rec_c, rec_c.data(), (int64_t)rec_c.size())) return 1;
    if (dev_upload(ctx, &a.blk_rec_p, rec_p.data(), (int64_t)rec_p.size())) return 1;
    if (pk_i.empty()) { pk_i.push_back(make_int2(0, 0)); pk_e.push_back(0.0); }
    if (dev_upload(ctx, &a.blk_pk_i, pk_i.data(), (int64_t)pk_i.size())) return 1;
    if (dev_upload(ctx, &a.blk_pk_e, pk_e.data(), (int64_t)pk_e.size())) return 1;
    if (dev_upload(ctx, &a.blk_stat[0], stat_plain.data(), (int64_t)stat_plain.size())) return 1;
    if (dev_upload(ctx, &a.blk_stat[1], stat_bme.data(), (int64_t)stat_bme.size())) return 1;
    if (dev_upload(ctx, &a.rep_soff, rep_soff.data(), (int64_t)rep_soff.size())) return 1;
    std::vector<int32_t> cl_order((size_t)a.n_reps);
    std::iota(cl_order.begin(), cl_order.end(), 0);
    std::stable_sort(cl_order.begin(), cl_order.end(), [&](int32_t x, int32_t y) { return rep_soff[x + 1] - rep_soff[x] > rep_soff[y + 1] - rep_soff[y]; });
    if (dev_upload(ctx, &a.cl_order, cl_order.data(), a.n_reps)) return 1;
    if (dev_upload(ctx, &a.mem_block, mem_block.data(), (int64_t)mem_block.size())) return 1;
    if (dev_upload(ctx, &a.rep_boff, rep_boff.data(), (int64_t)rep_boff.size())) return 1;
    if (dev_upload(ctx, &a.rep_loff, rep_loff.data(), (int64_t)rep_loff.size())) return 1;
    if (dev_upload(ctx, &a.loose_mp, loose_mp.data(), (int64_t)loose_mp.size())) return 1;
    if (dev_upload(ctx, &a.blk_root, blk_root.data(), n_blocks)) return 1;
    if (dev_upload(ctx, &a.blk_rslot, blk_rslot.data(), n_blocks)) return 1;
    if (dev_upload(ctx, &a.blk_nodes, blk_nodes.data(), n_blocks)) return 1;
    if (dev_upload(ctx, &a.e_of_slot, e_of_slot.data(), a.n_refs)) return 1;
    if (dev_upload(ctx, &a.e_of_blk, e_of_blk.data(), n_blocks)) return 1;
    if (dev_upload(ctx, &a.e_node, e_node.data(), (int64_t)e_node.size())) return 1;
    if (dev_upload(ctx, &a.lvl_e, lvl_e.data(), (int64_t)lvl_e.size())) return 1;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));  // (the vectors are locals)
    a.n_blocks = n_blocks;
    a.n_e = (int64_t)ents.size();
    return 0;
}

int ensure_packed8(apples_ctx *ctx);

int setup_alignment(apples_ctx *ctx, const apples_tree *t, const apples_alignment *al) {
    DevAlign &a = ctx->aln;
    a.n_rows = al->n_rows;
    a.n_refs = al->n_refs;
    a.L = al->length;
    a.W = (a.L + 31) / 32;
    a.G = (a.W + 3) / 4;
    a.slots_pad = round_up(std::max<int64_t>(a.n_rows, 1), APPLES_TPB);
    if (a.L <= 0 || a.L > 65535) { ctx->err = "alignment length must be in [1, 65535]"; return 1; }
    if (al->n_refs > al->n_rows) { ctx->err = "n_refs > n_rows"; return 1; }
    std::vector<int32_t> level(t->level, t->level + t->n_nodes);
    for (int64_t r = 0; r < a.n_refs; ++r)
        if (al->row_node[r] >= t->n_nodes) { ctx->err = "row_node out of range"; return 1; }
    {  // two reference rows on one tree leaf: the observed leaves of a query would not be distinct nodes
        std::vector<uint8_t> seen(t->n_nodes, 0);
        for (int64_t r = 0; r < a.n_refs; ++r) {
            const int nd = al->row_node[r];
            if (nd < 0) continue;
            if (seen[nd]) { ctx->tree.merge_ok = false; break; }
            seen[nd] = 1;
        }
    }
    std::vector<int32_t> ord = level_order(al->row_node, a.n_refs, level, ctx->tree.scan);
    a.slot_row.assign(a.n_rows, 0);
    a.row_slot.assign(a.n_rows, 0);
    for (int64_t s = 0; s < a.n_refs; ++s) { a.slot_row[s] = ord[s]; a.row_slot[ord[s]] = (int32_t)s; }
    for (int64_t r = a.n_refs; r < a.n_rows; ++r) { a.slot_row[r] = (int32_t)r; a.row_slot[r] = (int32_t)r; }
    std::vector<int32_t> slot_node(a.n_refs), slot_level(a.n_refs);
    for (int64_t s = 0; s < a.n_refs; ++s) {
        int nd = al->row_node[a.slot_row[s]];
        slot_node[s] = nd;
        slot_level[s] = nd >= 0 ? level[nd] : -1;
    }
    // clusters
    std::vector<int32_t> rep_slot, rep_moff, mem_slot, slot_rep(a.n_refs, -1), slot_mpos(a.n_refs, 0);
    if (al->rep_row == nullptr) {
        a.n_reps = a.n_refs;
        a.all_singleton = true;
        rep_slot.resize(a.n_reps); rep_moff.resize(a.n_reps + 1); mem_slot.resize(a.n_reps);
        for (int64_t j = 0; j < a.n_reps; ++j) {
            rep_slot[j] = a.row_slot[j]; rep_moff[j] = (int32_t)j; mem_slot[j] = a.row_slot[j];
            slot_rep[a.row_slot[j]] = (int32_t)j;
        }
        rep_moff[a.n_reps] = (int32_t)a.n_reps;
    } else {
        a.n_reps = al->n_reps;
        a.all_singleton = true;
        rep_slot.resize(a.n_reps); rep_moff.assign(al->member_off, al->member_off + a.n_reps + 1);
        mem_slot.resize(rep_moff[a.n_reps]);
        for (int64_t j = 0; j < a.n_reps; ++j) {
            if (al->rep_row[j] < 0 || al->rep_row[j] >= a.n_rows) { ctx->err = "rep_row out of range"; return 1; }
            rep_slot[j] = a.row_slot[al->rep_row[j]];
            int cnt = rep_moff[j + 1] - rep_moff[j];
            if (cnt != 1 || al->member_row[rep_moff[j]] != al->rep_row[j]) a.all_singleton = false;
            for (int m = rep_moff[j]; m < rep_moff[j + 1]; ++m) {
                int r = al->member_row[m];
                if (r < 0 || r >= a.n_refs) { ctx->err = "member_row out of range"; return 1; }
                mem_slot[m] = a.row_slot[r];
                slot_rep[a.row_slot[r]] = (int32_t)j;
                slot_mpos[a.row_slot[r]] = m - rep_moff[j];
            }
        }
        for (int64_t s = 0; s < a.n_refs; ++s)
            if (slot_rep[s] < 0) { ctx->err = "every reference row must belong to exactly one cluster"; return 1; }
        if (a.all_singleton && a.n_reps != a.n_refs) a.all_singleton = false;
    }
    if (dev_upload(ctx, &a.slot_node, slot_node.data(), a.n_refs)) return 1;
    if (dev_upload(ctx, &a.slot_level, slot_level.data(), a.n_refs)) return 1;
    dev_free(a.lvl_slots); a.lvl_slots = nullptr;
    if (!ctx->tree.scan) {  // slots with a level above l, for l = -1 .. height (k_select_clusters ranks these in its bitmap)
        const int H = ctx->tree.height;
        std::vector<int32_t> lvl_slots(H + 2, 0);
        bool sorted = true;
        for (int64_t s = 0; s < a.n_refs; ++s) {
            if (s > 0 && slot_level[s] > slot_level[s - 1]) sorted = false;
            if (slot_level[s] > H) sorted = false;
            else if (slot_level[s] >= 0) ++lvl_slots[slot_level[s]];  // for now: slots AT level l in entry l
        }
        for (int l = H, above = 0; l >= -1; --l) {  // entry l + 1 = slots above level l
            const int at = l >= 0 ? lvl_slots[l] : 0;
            lvl_slots[l + 1] = above;
            above += at;
        }
        if (sorted && dev_upload(ctx, &a.lvl_slots, lvl_slots.data(), (int64_t)lvl_slots.size())) return 1;
    }
    if (dev_upload(ctx, &a.slot_rep, slot_rep.data(), a.n_refs)) return 1;
    if (dev_upload(ctx, &a.slot_mpos, slot_mpos.data(), a.n_refs)) return 1;
    if (dev_upload(ctx, &a.rep_slot, rep_slot.data(), a.n_reps)) return 1;
    if (dev_upload(ctx, &a.rep_moff, rep_moff.data(), a.n_reps + 1)) return 1;
    if (dev_upload(ctx, &a.mem_slot, mem_slot.data(), (int64_t)mem_slot.size())) return 1;
    if (build_blocks(ctx, t, level, slot_node, slot_rep, slot_mpos, rep_moff)) return 1;

    // the rows go to the device as the caller holds them (one copy, no re-ordered staging buffer on the
    // host); the packing kernels gather them into slot order through d_slot_row
    if (dev_upload(ctx, &a.raw, al->rows, a.n_rows * (int64_t)a.L)) return 1;
    if (dev_upload(ctx, &a.d_slot_row, a.slot_row.data(), a.n_rows)) return 1;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));

    if (ctx->params.model == APPLES_SCOREDIST) {
        int Lpad = (a.L + 15) / 16 * 16;
        if (dev_alloc(ctx, &a.aa_idx, (int64_t)(Lpad / 16) * a.slots_pad * 16)) return 1;
        HIP_TRY(ctx, hipMemsetAsync(a.aa_idx, 160, (size_t)(Lpad / 16) * a.slots_pad * 16, ctx->stream));
        if (dev_alloc(ctx, &a.aa_mask, (int64_t)(Lpad / 16) * a.slots_pad)) return 1;
        HIP_TRY(ctx, hipMemsetAsync(a.aa_mask, 0, (size_t)(Lpad / 16) * a.slots_pad * 2, ctx->stream));
        if (launch_pack_aa(ctx, a.raw, a.n_rows, a.L, a.aa_idx, a.aa_mask, a.slots_pad, false, nullptr, a.d_slot_row)) return 1;
        a.planes = 0;
        double tab[21 * 21];
        for (int i = 0; i < 21; ++i)
            for (int j = 0; j < 21; ++j) tab[i * 21 + j] = (i < 20 && j < 20) ? kBlosum45[i * 20 + j] : 0.0;
        if (dev_upload(ctx, &ctx->blosum, tab, 21 * 21)) return 1;
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        // one-hot operand image of the reference rows for the matrix-core filter of the fused pass (dist_sd.hip): 10 bytes per
        // site and slot (253 MB at 50 000 x 500)
        if (a.all_singleton && !(ctx->dbg & APPLES_DBG_NO_SD_GEMM) && sd_steps(a.L) >= 2 && a.L <= 4096 &&
            a.slots_pad <= 458752) {  // (the evaluating kernels stage a query row and the segment counts' prefix in LDS: 64 KB per workgroup)
            a.sd_fp6 = (ctx->dbg & APPLES_DBG_SD_FP6) != 0;  // (off by default: DESIGN.md section 5)
            uint8_t codes[400];
            sd_table_codes(kBlosum45, codes, a.sd_fp6);
            if (dev_upload(ctx, &ctx->sd_tq4, codes, 400)) return 1;
            if (dev_alloc(ctx, &a.sd_ref4, a.slots_pad * (int64_t)sd_steps(a.L) * 64)) return 1;
            if (dev_alloc(ctx, &a.sd_nvr, a.slots_pad)) return 1;
            if (launch_sd_expand(ctx, a.raw, a.n_rows, a.sd_ref4, a.slots_pad, ctx->stream, a.d_slot_row, 0, false, a.sd_nvr, nullptr,
                                 a.aa_mask, a.slots_pad)) return 1;
            a.aa_Lrow = (int32_t)round_up(a.L, 64);
            if (dev_alloc(ctx, &a.aa_rows, a.slots_pad * (int64_t)a.aa_Lrow)) return 1;
            if (dev_alloc(ctx, &a.aa_mrows, a.slots_pad * (int64_t)(a.aa_Lrow / 16))) return 1;
            if (launch_sd_rows(ctx)) return 1;
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));  // (codes is on the stack)
        }
        // clustered reference (the default route of -p): the representatives' rows in representative order for the fused selection
        // by representatives (select.hip: k_select_clusters around k_cluster_dist_sd)
        if (!a.all_singleton && a.n_refs <= SELECT_CLUSTERS_MAX_SLOTS && !(ctx->dbg & APPLES_DBG_NO_FUSE))
            if (launch_build_cluster_panels_aa(ctx)) return 1;
    } else {
        int *d_exotic = nullptr;
        if (dev_alloc(ctx, &d_exotic, 1)) return 1;
        HIP_TRY(ctx, hipMemsetAsync(d_exotic, 0, sizeof(int), ctx->stream));
        a.planes = 2;
        int64_t words = (int64_t)a.G * 3 * a.slots_pad;
        if (dev_alloc(ctx, &a.packed, words)) return 1;
        HIP_TRY(ctx, hipMemsetAsync(a.packed, 0, (size_t)words * sizeof(uint4), ctx->stream));
        // bytes beyond ACGT- in a singleton context whose fused pass runs on the matrix cores: the 2-plane rows and the fp4 images
        // keep them as gaps, the 8-plane form comes beside them (DevAlign::ex_ok)
        a.ex_ok = a.all_singleton && dist_mfma_enabled(ctx) && a.L < 8192;
        int32_t *d_row_bad = nullptr;
        if (a.ex_ok) {
            if (dev_alloc(ctx, &d_row_bad, a.slots_pad)) return 1;
            HIP_TRY(ctx, hipMemsetAsync(d_row_bad, 0, (size_t)a.slots_pad * sizeof(int32_t), ctx->stream));
        }
        if (launch_pack_rows(ctx, a.raw, a.n_rows, a.L, 2, a.packed, a.slots_pad, false, d_exotic, nullptr, a.d_slot_row, d_row_bad)) return 1;
        int exotic = 0;
        HIP_TRY(ctx, hipMemcpyAsync(&exotic, d_exotic, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        if (exotic && a.ex_ok) {
            // which slots, which sites (from the caller's rows: only the flagged rows are read again), and the 8-plane form
            std::vector<int32_t> bad((size_t)a.n_rows), off((size_t)a.n_rows + 1, 0);
            HIP_TRY(ctx, hipMemcpy(bad.data(), d_row_bad, (size_t)a.n_rows * sizeof(int32_t), hipMemcpyDeviceToHost));
            std::vector<uint16_t> sites;
            a.ex_max = 0;
            for (int64_t s = 0; s < a.n_rows; ++s) {
                if (bad[s]) {
                    const uint8_t *row = al->rows + (int64_t)a.slot_row[s] * a.L;
                    for (int i = 0; i < a.L; ++i) {
                        const uint8_t b = row[i];
                        if (b != '-' && b != 'A' && b != 'C' && b != 'G' && b != 'T') sites.push_back((uint16_t)i);
                    }
                }
                off[s + 1] = (int32_t)sites.size();
                a.ex_max = std::max(a.ex_max, off[s + 1] - off[s]);
            }
            if (sites.empty()) sites.push_back(0);
            if (dev_upload(ctx, &a.ex_off, off.data(), (int64_t)off.size())) return 1;
            if (dev_upload(ctx, &a.ex_site, sites.data(), (int64_t)sites.size())) return 1;
            if (ensure_packed8(ctx)) return 1;
            exotic = 0;
        }
        dev_free(d_row_bad);
        if (exotic) {  // symbols beyond ACGT-: keep the raw byte in 8 planes
            dev_free(a.packed);
            a.planes = 8;
            words = (int64_t)a.G * 9 * a.slots_pad;
            if (dev_alloc(ctx, &a.packed, words)) return 1;
            HIP_TRY(ctx, hipMemsetAsync(a.packed, 0, (size_t)words * sizeof(uint4), ctx->stream));
            if (launch_pack_rows(ctx, a.raw, a.n_rows, a.L, 8, a.packed, a.slots_pad, false, d_exotic, nullptr, a.d_slot_row)) return 1;
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        }
        dev_free(d_exotic);
        // pre-expanded reference image for the GEMM form of the fused pass (dist_gemm.hip): 2 bytes per site
        if (a.all_singleton && a.planes == 2 && a.L <= GEMM_MAX_L && dist_mfma_enabled(ctx) && !(ctx->dbg & APPLES_DBG_NO_DIST_GEMM)) {
            // (the image's own allocation marks the context's fp4 images as compact: 64 bytes per 64-site block)
            if (dev_alloc(ctx, &a.ref_f4, a.slots_pad * (int64_t)a.G * 128)) return 1;
            if (launch_expand_queries_f4(ctx, a.raw, a.n_rows, a.ref_f4, a.slots_pad, ctx->stream, a.d_slot_row)) return 1;
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        }
        // clustered reference on the ACGT- fast path: panels for the fused selection by representatives
        if (!a.all_singleton && a.planes == 2 && a.n_refs <= SELECT_CLUSTERS_MAX_SLOTS && a.G <= 64 &&
            !(ctx->dbg & APPLES_DBG_NO_FUSE) && dist_mfma_enabled(ctx))
            if (launch_build_cluster_panels(ctx)) return 1;
    }
    return 0;
}

// the reference rows once more in the 8-plane form, beside the 2-plane rows of a context that keeps bytes beyond ACGT- as gaps there
// (DevAlign::ex_ok): built with the first such byte, in the reference or in a query block
int ensure_packed8(apples_ctx *ctx) {
    DevAlign &a = ctx->aln;
    if (a.packed8 || a.planes == 8) return 0;
    const int64_t words = (int64_t)a.G * 9 * a.slots_pad;
    if (dev_alloc(ctx, &a.packed8, words)) return 1;
    HIP_TRY(ctx, hipMemsetAsync(a.packed8, 0, (size_t)words * sizeof(uint4), ctx->stream));
    if (launch_pack_rows(ctx, a.raw, a.n_rows, a.L, 8, a.packed8, a.slots_pad, false, nullptr, nullptr, a.d_slot_row)) return 1;
    // query blocks that exist already: their 8-plane form as well (what exact8_rows asks for)
    for (auto &qb : ctx->blocks) {
        if (!qb.live || qb.table || qb.planes != 2 || qb.packed8 || !qb.raw) continue;
        const int64_t w = qb.n_pad * a.G * 9;
        if (blk_alloc(ctx, &qb.packed8, w)) return 1;
        HIP_TRY(ctx, hipMemsetAsync(qb.packed8, 0, (size_t)w * sizeof(uint4), ctx->stream));
        if (launch_pack_rows(ctx, qb.raw, qb.n, a.L, 8, qb.packed8, 0, true, nullptr)) return 1;
    }
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

int repack_to_bytes(apples_ctx *ctx) {  // a query block carries symbols beyond ACGT-: widen the reference
    DevAlign &a = ctx->aln;
    if (a.planes == 8) return 0;
    int *d_exotic = nullptr;
    if (dev_alloc(ctx, &d_exotic, 1)) return 1;
    dev_free(a.rep_packed); dev_free(a.packed_rm); dev_free(a.ref_f4);  // the fast paths are for ACGT- contexts only
    a.rep_packed = a.packed_rm = nullptr;
    a.ref_f4 = nullptr;
    dev_free(a.packed);
    a.planes = 8;
    int64_t words = (int64_t)a.G * 9 * a.slots_pad;
    if (dev_alloc(ctx, &a.packed, words)) return 1;
    HIP_TRY(ctx, hipMemsetAsync(a.packed, 0, (size_t)words * sizeof(uint4), ctx->stream));
    if (launch_pack_rows(ctx, a.raw, a.n_rows, a.L, 8, a.packed, a.slots_pad, false, d_exotic, nullptr, a.d_slot_row)) return 1;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    dev_free(d_exotic);
    // existing query blocks were packed with 2 planes: repack them
    for (auto &qb : ctx->blocks) {
        if (!qb.live || qb.planes == 8) continue;
        blk_free(ctx, qb.packed);
        blk_free(ctx, qb.qf4);  // the fp4 image only serves the ACGT- fast path
        qb.qf4 = nullptr;
        int64_t w = qb.n_pad * a.G * 9;
        if (blk_alloc(ctx, &qb.packed, w)) return 1;
        HIP_TRY(ctx, hipMemsetAsync(qb.packed, 0, (size_t)w * sizeof(uint4), ctx->stream));
        int *d_ex2 = nullptr;
        if (dev_alloc(ctx, &d_ex2, 1)) return 1;
        if (launch_pack_rows(ctx, qb.raw, qb.n, a.L, 8, qb.packed, 0, true, d_ex2)) return 1;
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        dev_free(d_ex2);
        qb.planes = 8;
    }
    return 0;
}

// HYBRID (apples/Algorithm.py:76-82) needs every edge's solution once the sweep is through: the lean sweep keeps them in its
// entries and ranks them itself (sweep_lean.hip:lean_hybrid_pick); the level loop and the scan sweep write per-edge records
// (Workspace::Sweep::xe).  True = this context's HYBRID passes take the records.  APPLES_DBG_HYBRID_RECORDS: on every tree.
bool hybrid_records(const apples_ctx *ctx) {
    if (ctx->params.criterion != APPLES_HYBRID) return false;
    return (ctx->dbg & APPLES_DBG_HYBRID_RECORDS) || !sweep_lean_layout(ctx->tree, false);
}

// observed-leaf count above which a query goes straight to a workgroup-sized sweep team
int big_threshold(const apples_ctx *ctx) {
    const int env = (int)knob(ctx, "APPLES_BIG_THRESHOLD", 0);
    // the lean sweep's wavefront-sized teams are the efficient ones and take their queue largest first, finely graded
    // (C3 sweep 15.8 / 15.8 / 16.7 / 18.7 ms at 4 096 / 8 192 / 12 288 / 16 384, the clustered route's 38.4 / 35.5 / 35.1 / 36.0);
    // the level loop's cut was measured at 4 096
    const int v = env > 0 ? env : (sweep_lean_layout(ctx->tree, hybrid_records(ctx)) ? LEAN_BIG_THRESHOLD : 4096);
    // scan sweep: a wavefront-sized team keeps the per-leaf state of at most SCAN_LDS_LEAVES_SMALL leaves in LDS
    return ctx->tree.scan ? std::min(v, SCAN_LDS_LEAVES_SMALL) : v;
}

// ... for the device batch under way: a small batch (a shard of a multi-GPU job, a -d block) routes from half the count.  Its
// wavefront-sized teams have three or four queries each, so their kernels' time is the longest query's, and a query of 8 000
// observed leaves takes a wavefront about four times as long as a workgroup-sized team (measured at config 3, host -> host:
// 12 500 queries 7.15 -> 6.80 ms, sweep 2.40 -> 2.11; the full 100 000, in batches of 25 000, 48.7 -> 49.1: the cut stays where
// it was there).  Never above the count the workspace was sized with.
int route_threshold(const apples_ctx *ctx) {
    int v = big_threshold(ctx);
    const bool fixed = knob_on(ctx, "APPLES_BIG_THRESHOLD");  // (the knob fixes the cut for every batch size)
    // (a workspace regrown with per-edge records -- a HYBRID pass, apples_sweep_edges -- stays with the level loop for later
    // MLSE / ME passes too: its cut, not the lean sweep's the tree would be eligible for)
    if (!fixed && !ctx->tree.scan && ctx->ws.batch > 0 && !ctx->ws.small.lean) v = std::min(v, 4096);
    if (fixed || ctx->tree.scan || v != LEAN_BIG_THRESHOLD) return v;
    // (13 312: the clustered route's batches of 14 300 queries with 3 100 observed leaves each lose by the lower cut -- sweep 35.6 ->
    // 37.8 ms per pass, the workgroup-sized teams flooded -- and stay with the higher one)
    return ctx->cur_batch_queries > 0 && ctx->cur_batch_queries <= LEAN_SMALL_BATCH ? v / 2 : v;
}

// ints per query / team of the lean sweep's group offsets: height + 4, and once more for the child records' offsets on a tree with polytomies
static int64_t lean_grp_stride(const DevTree &t) { return (int64_t)(t.height + 4) * ((t.max_children > 2 || t.force_poly) ? 2 : 1); }

void free_sweep(Workspace::Sweep &sw) {
    dev_free(sw.map); dev_free(sw.ver); dev_free(sw.order); dev_free(sw.ent); dev_free(sw.grp_off); dev_free(sw.A); dev_free(sw.B); dev_free(sw.xe);
    dev_free(sw.ent_f); dev_free(sw.ent_i); dev_free(sw.leaf_g); dev_free(sw.meta); dev_free(sw.lean); dev_free(sw.lean_leaf); dev_free(sw.lean_meta);
    sw = Workspace::Sweep();
}

void swap_bufs(Workspace &w) {
    BatchBuf &a = w.alt;
    std::swap(w.dist, a.dist); std::swap(w.counts, a.counts); std::swap(w.obs_node, a.obs_node);
    std::swap(w.obs_dist, a.obs_dist); std::swap(w.cnt_gt, a.cnt_gt); std::swap(w.n_obs, a.n_obs);
    std::swap(w.seg_slot, a.seg_slot); std::swap(w.seg_cnt, a.seg_cnt); std::swap(w.dist_slow, a.dist_slow);
    std::swap(w.slow_list, a.slow_list); std::swap(w.slow_count, a.slow_count); std::swap(w.route_list, a.route_list);
    std::swap(w.route_count, a.route_count); std::swap(w.overflow_list, a.overflow_list);
    std::swap(w.overflow_count, a.overflow_count);
    std::swap(w.cls_list, a.cls_list); std::swap(w.cls_count, a.cls_count);
}

void free_workspace(Workspace &w) {
    {
        BatchBuf &a = w.alt;
        dev_free(a.dist); dev_free(a.counts); dev_free(a.obs_node); dev_free(a.obs_dist); dev_free(a.cnt_gt);
        dev_free(a.n_obs); dev_free(a.seg_slot); dev_free(a.seg_cnt); dev_free(a.dist_slow); dev_free(a.slow_list);
        dev_free(a.route_list); dev_free(a.overflow_list);
        dev_free(a.cls_list); dev_free(a.cls_count);  // (the other counters live inside cls_count's block)
    }
    dev_free(w.dist); dev_free(w.counts); dev_free(w.obs_node); dev_free(w.obs_dist); dev_free(w.cnt_gt);
    dev_free(w.n_obs); dev_free(w.overflow_list); dev_free(w.route_list);
    dev_free(w.cls_list); dev_free(w.cls_count); dev_free(w.seg_slot); dev_free(w.seg_cnt);
    dev_free(w.dist_slow); dev_free(w.slow_list); dev_free(w.row_off);
    free_sweep(w.small);
    free_sweep(w.big);
    w = Workspace();
}

int alloc_sweep(apples_ctx *ctx, Workspace::Sweep &sw, int wgs, int teams_per_wg, int64_t cap, int64_t leaf_cap, bool xe, int64_t batch) {
    const DevTree &t = ctx->tree;
    sw.wgs = wgs;
    sw.teams = (int64_t)wgs * teams_per_wg;
    sw.cap = cap;
    sw.leaf_cap = leaf_cap;
    if (t.scan) {  // scan formulation: component arrays of `cap` entries per team (16-byte aligned pairs)
        sw.cap = cap = round_up(cap, 64);
        if (dev_alloc(ctx, &sw.ent_f, sw.teams * 13 * cap)) return 1;
        if (dev_alloc(ctx, &sw.ent_i, sw.teams * 5 * cap)) return 1;
        if (dev_alloc(ctx, &sw.meta, sw.teams * 4)) return 1;
        const int lds_leaves = teams_per_wg == 1 ? SCAN_LDS_LEAVES_BIG : SCAN_LDS_LEAVES_SMALL;
        if (leaf_cap > lds_leaves) {
            sw.leaf_cap = leaf_cap = round_up(leaf_cap, 2);
            if (dev_alloc(ctx, &sw.leaf_g, sw.teams * leaf_cap * 3)) return 1;  // 6 bytes per leaf
        } else {
            sw.leaf_cap = 0;
        }
        if (xe)
            if (dev_alloc(ctx, &sw.xe, sw.teams * cap * 18)) return 1;
        return 0;
    }
    if (sweep_lean_layout(t, xe) && teams_per_wg == 4) {
        // sweep_lean.hip, wavefront-sized teams: a pool of entries for the whole batch (bottom-up and top-down are two
        // kernels; a query's entries live from one to the other), per-query level offsets and hand-over records, and
        // per-leaf scratch for the bottom-up teams.  `cap` = pool entries here.
        sw.lean_cap1 = cap / 4 * 4;
        sw.lean_leaf1 = round_up(std::max<int64_t>(std::min<int64_t>(leaf_cap, big_threshold(ctx)), 4), 4);
        sw.teams = sweep_lean_up_teams(ctx);
        char *p = nullptr;
        if (dev_alloc(ctx, &p, sw.lean_cap1 * LEAN_BYTES_PER_NODE)) return 1;
        sw.lean = p;
        if (dev_alloc(ctx, &p, 3 * sw.teams * sw.lean_leaf1 * LEAN_BYTES_PER_LEAF)) return 1;  // (x 3: run_sweep_second's teams, the second half's of a small batch)
        sw.lean_leaf = p;
        if (dev_alloc(ctx, &sw.grp_off, batch * lean_grp_stride(t))) return 1;
        if (dev_alloc(ctx, &sw.lean_meta, batch)) return 1;
        return 0;
    }
    if (sweep_lean_layout(t, xe)) {  // sweep_lean.hip, workgroup-sized teams: field arrays in place of ent and A
        // (a tree with polytomies: a team's child records live in its entry slots from the top down -- at most one per child of a
        // polytomy, DevTree::poly_kids)
        sw.lean_cap1 = round_up(cap + 1 + t.poly_kids, 4);
        sw.lean_leaf1 = round_up(std::max<int64_t>(leaf_cap, 4), 4);
        char *p = nullptr;
        if (dev_alloc(ctx, &p, sw.teams * (sw.lean_cap1 * LEAN_BYTES_PER_NODE + sw.lean_leaf1 * LEAN_BYTES_PER_LEAF))) return 1;
        sw.lean = p;
        if (dev_alloc(ctx, &sw.grp_off, sw.teams * lean_grp_stride(t))) return 1;
        return 0;
    }
    if (sweep_merge_lists(t)) {
        if (dev_alloc(ctx, &sw.ent, sw.teams * (cap + 1))) return 1;
    } else if (!sweep_bits_in_lds(t)) {
        if (dev_alloc(ctx, &sw.map, sw.teams * (int64_t)t.n_nodes)) return 1;
        HIP_TRY(ctx, hipMemsetAsync(sw.map, 0, (size_t)sw.teams * t.n_nodes * 4, ctx->stream));
        if (dev_alloc(ctx, &sw.ver, sw.teams)) return 1;
        HIP_TRY(ctx, hipMemsetAsync(sw.ver, 0, (size_t)sw.teams * 4, ctx->stream));
        if (dev_alloc(ctx, &sw.order, sw.teams * (cap + 1))) return 1;
    }
    if (dev_alloc(ctx, &sw.grp_off, sw.teams * (int64_t)(t.height + 4))) return 1;
    HIP_TRY(ctx, hipMalloc(&sw.A, (size_t)sw.teams * (cap + 1) * 64));
    if (t.max_children > 2) HIP_TRY(ctx, hipMalloc(&sw.B, (size_t)sw.teams * (cap + 1) * 48));
    if (xe)
        if (dev_alloc(ctx, &sw.xe, sw.teams * (cap + leaf_cap) * 18)) return 1;
    return 0;
}

// workspaces for `members` rows/columns per query
// `slim`: the fused matrix-core path writes no full distance rows for the batch -- only the queries on the top-up
// list get them -- so dist and dist_slow hold a slice of the batch (an eighth) and the list is walked in slices:
// a query then costs 4 + 2 bytes per reference slot instead of 20, and a C3 pass needs 4 device batches, not 7.
// `seg_stride` > 0 (the clustered fused route): its survivors are representatives, rows of seg_stride = reps_pad entries, and
// only what its top-up phase forwards takes a full row -- seg_slot / seg_cnt are that narrow and dist_slow holds a slice; dist
// (the queries' rows of member distances) stays whole.  6.4 -> 4.2 MB per query at 200 000 references: five device batches
// for config 3's 100 000 queries, not seven (every kernel of the route has a tail of a few hundred microseconds per batch).
int ensure_workspace(apples_ctx *ctx, int64_t members, int64_t stride, int64_t want_batch, bool need_dist,
                     bool need_counts, bool need_xe, bool need_fused = false, bool need_alt = false, bool slim = false,
                     int64_t seg_stride = 0, bool want_ragged = false) {
    Workspace &w = ctx->ws;
    const DevTree &t = ctx->tree;
    int64_t batch = want_batch;
    if (ctx->params.max_batch > 0) batch = std::min(batch, (int64_t)ctx->params.max_batch);
    const bool no_slim = knob_on(ctx, "APPLES_NO_SLIM_BATCH");  // diagnostic knob
    slim = slim && need_fused && !need_counts && !need_alt && !no_slim;
    const bool cslim = seg_stride > 0 && seg_stride < stride && need_fused && !need_counts && !need_alt && !slim && !no_slim;
    int64_t per_q = stride * 8 + members * 12 + (int64_t)(t.height + 2) * 4 + (need_counts ? stride * 4 : 0) +
                    (need_fused ? stride * 12 + stride / 16 : 0);
    if (slim) per_q = stride * 4 + stride / 16 + stride * 2 + members * 12 + (int64_t)(t.height + 2) * 4;
    if (cslim) per_q = stride * 8 + stride + seg_stride * 4 + seg_stride / 16 + members * 12 + (int64_t)(t.height + 2) * 4;
    // ragged rows (common.h, Workspace::ragged): the clustered fused route on the lean sweep, a reference of at least four small rows
    const int64_t row_small = knob(ctx, "APPLES_RAGGED_SMALL", 16384);  // (test knob: small rows that send many queries to the big ones)
    const bool ragged = want_ragged && cslim && !need_xe && !w.big.xe && sweep_lean_layout(t, false) && stride >= 4 * row_small && row_small >= 16 &&
                        !ctx->no_ragged && !knob_on(ctx, "APPLES_NO_RAGGED");
    int64_t row_big = 0;
    if (ragged) {
        // big rows: for a sixteenth of the wanted batch, 1 024 to 8 192 of them (APPLES_RAGGED_BIG: test knob -- a block that runs out)
        row_big = std::min<int64_t>(8192, std::max<int64_t>(1024, want_batch / 16));
        if (knob_on(ctx, "APPLES_RAGGED_BIG")) row_big = std::max<int64_t>(1, knob(ctx, "APPLES_RAGGED_BIG", 0));
        per_q = row_small * 20 + stride + seg_stride * 4 + seg_stride / 16 + (int64_t)(t.height + 2) * 4 + 8;
    }
    // batch buffers: up to 144 GiB, at most half of what is free on the card (288 GB HBM3E; the clade blocks' pool, sized afterwards
    // from what is then free -- a third of it, 16 GiB at most -- and the scoredist rows of representatives come on top; bigger
    // batches amortise the sweep's tail: at 200 k leaves 16 k-query batches are 13 % faster than 5 k).
    // Allocating them is not free: with 160 GiB a resident C3 pass is another 4 % faster, but a one-shot
    // command-line run of the same size pays 1.1 to 3.4 s more for the allocation.
    int64_t capq = 0;
    auto size_batch = [&]() {
        // (half of what is free, 144 GiB at most: on an empty 288-GB device the clustered route's 100 000 queries at 200 000
        // references are then four device batches, not five -- 36.0 -> 34.4 ms; the other workloads do not move:
        // profiles/r05_batch_gib_exp.txt)
        int64_t budget_gib = 96;
        size_t fr = 0, tot = 0;
        if (hipMemGetInfo(&fr, &tot) == hipSuccess) budget_gib = std::max<int64_t>(8, std::min<int64_t>(144, (int64_t)(fr >> 30) / 2));
        if (ctx->params.batch_gib > 0) budget_gib = std::min<int64_t>(budget_gib, ctx->params.batch_gib);  // the caller's cap only lowers it
        if (knob_on(ctx, "APPLES_BATCH_GIB")) budget_gib = std::max<int64_t>(1, knob(ctx, "APPLES_BATCH_GIB", 0));  // tuning knob (experiments: replaces both)
        // (ragged rows: the big rows come out of the same budget, a quarter of it at most -- a one-shot run's 24 GiB: 1 500 of them
        // at 200 000 leaves)
        if (ragged && !knob_on(ctx, "APPLES_RAGGED_BIG"))
            row_big = std::max<int64_t>(64, std::min<int64_t>(row_big, ((budget_gib << 30) / 4) / (stride * 20) - 1));
        const int64_t fixed = ragged ? (row_big + 1) * stride * 20 : 0;
        capq = std::max<int64_t>(32, (((need_alt ? budget_gib / 2 : budget_gib) << 30) - fixed) / std::max<int64_t>(per_q, 1));
        int64_t b = want_batch;
        if (ctx->params.max_batch > 0) b = std::min(b, (int64_t)ctx->params.max_batch);
        b = std::min(b, capq);
        // (ragged rows: config 3 through clusters at device batches of 100 000 / 50 016 / 33 344 / 25 024 queries: 28.75 / 27.74 /
        // 28.35 / 28.95 ms per pass, 26.65 / 26.62 / 27.46 / 28.28 of it on the device -- beyond 50 000 the kernels gain nothing more
        // and a host buffer's first chunk travels with nothing to hide behind; APPLES_RAGGED_BATCH: tuning knob)
        if (ragged) b = std::min<int64_t>(b, std::max<int64_t>(32, knob(ctx, "APPLES_RAGGED_BATCH", 50016)));
        if (sweep_lean_layout(t, need_xe || w.big.xe != nullptr)) {
            // sweep_lean.hip's pool cursor is a 32-bit counter that every query of the batch adds its share to, whether the pool
            // still has room or not: the batch's requests together must stay below 2^32 (the largest share: lean_query_cap of
            // the routing threshold, 3 n + 1 027 entries; 167 000 queries at the default threshold)
            const int64_t share = 3 * std::min<int64_t>(std::max<int64_t>(members, w.obs_cap), big_threshold(ctx)) + 1028;
            b = std::min<int64_t>(b, (int64_t)(0xffffffffll / share) / 32 * 32 - 32);
        }
        return round_up(std::max<int64_t>(b, 1), 32);
    };
    batch = size_batch();
    bool regrow = batch > w.batch || members > w.obs_cap || stride > w.stride || (need_counts && !w.counts) ||
                  (need_xe && !w.big.xe) || (need_dist && !w.dist) || (need_fused && !w.seg_slot) || (need_alt && !w.has_alt) ||
                  // full rows wanted where only a slice exists (run_block steps by w.batch) -- unless the whole block fits the slice
                  // (a slim workspace serves the few queries of an exact8 block as it is: no regrowing back and forth)
                  (!slim && !cslim && w.dist_rows < std::min(w.batch, round_up(std::max<int64_t>(want_batch, 1), 32))) ||
                  // rows of another shape (every other route indexes q x obs_cap / q x stride)
                  ragged != w.ragged || (ragged && (row_small != w.row_small || row_big > w.row_big));
    if (!regrow) return 0;
    if (!ctx->blk_cache.empty()) {  // cached block buffers count as used in hipMemGetInfo: give them back, then size the batch
        for (auto &c : ctx->blk_cache) dev_free(c.second);
        ctx->blk_cache.clear();
        batch = size_batch();
    }
    if ((slim == w.slim || !w.slim) && (cslim == w.cslim || !w.cslim) && ragged == w.ragged) batch = std::max(batch, w.batch);
    if (!slim) batch = std::min(batch, std::max<int64_t>(capq, 32));  // (a slim workspace's batch would not fit with full rows)
    batch = round_up(std::max<int64_t>(batch, 1), 32);
    const int64_t drows = (slim || cslim) ? std::min(batch, std::max<int64_t>(2048, batch / 8)) : batch;
    const int64_t seg_w = cslim ? seg_stride : std::max<int64_t>(stride, 1);  // entries of a row of seg_slot
    int64_t obs_cap = std::max(members, w.obs_cap);
    stride = std::max(stride, w.stride);
    bool xe = need_xe || w.big.xe != nullptr, had_counts = w.counts != nullptr;
    bool fused = need_fused || w.seg_slot != nullptr;
    bool alt = need_alt || w.has_alt;
    free_workspace(w);
    w.batch = batch;
    w.obs_cap = obs_cap;
    w.stride = stride;
    w.slim = slim;
    w.cslim = cslim;
    w.dist_rows = drows;
    w.ragged = ragged; w.row_small = ragged ? row_small : 0; w.row_big = row_big;
    const int64_t rag_entries = ragged ? batch * row_small + (row_big + 1) * stride : 0;  // (+ 1: the row the queries share when the big ones run out)
    if (ragged) {
        if (dev_alloc(ctx, &w.row_off, batch)) return 1;
        if (!ctx->d_rag && dev_alloc(ctx, &ctx->d_rag, 2)) return 1;
        w.dist_rows = std::min<int64_t>(w.dist_rows, rag_entries / std::max<int64_t>(stride, 1));  // (whole rows of `dist`, for whoever counts in them)
    }
    for (int set = 0; set < (alt ? 2 : 1); ++set) {
        if (dev_alloc(ctx, &w.dist, ragged ? rag_entries : (cslim ? batch : drows) * std::max<int64_t>(stride, 1))) return 1;
        if (need_counts || had_counts)
            if (dev_alloc(ctx, &w.counts, batch * std::max<int64_t>(stride, 1))) return 1;
        if (dev_alloc(ctx, &w.obs_node, ragged ? rag_entries : batch * obs_cap)) return 1;
        if (dev_alloc(ctx, &w.obs_dist, ragged ? rag_entries : batch * obs_cap)) return 1;
        if (dev_alloc(ctx, &w.cnt_gt, batch * (int64_t)(t.height + 2))) return 1;
        if (dev_alloc(ctx, &w.n_obs, batch)) return 1;
        if (fused) {
            if (dev_alloc(ctx, &w.seg_slot, batch * seg_w)) return 1;
            if (dev_alloc(ctx, &w.seg_cnt, batch * std::max<int64_t>(seg_w / 64, 1))) return 1;
            if (dev_alloc(ctx, &w.dist_slow, drows * std::max<int64_t>(stride, 1))) return 1;
            if (dev_alloc(ctx, &w.slow_list, 3 * batch)) return 1;  // the list, what is known about its entries (SelectArgs.slow_hint), and the
                                                                    // list of what the clustered route's phase 4 forwards to the general selection
        }
        if (dev_alloc(ctx, &w.route_list, 3 * batch)) return 1;
        if (dev_alloc(ctx, &w.overflow_list, batch)) return 1;
        if (dev_alloc(ctx, &w.cls_list, 16 * batch)) return 1;  // four size classes, then the largest class once more in four; all of it
                                                                // a second time for the overlapped top-up chain (run_sweep_second)
        // one block for every per-batch counter, cleared by one memset: [0..3] size-class counts,
        // [4..6] the sweep launches' work cursors, [7] the lean sweep's pool cursor, [8] routed, [9] top-up list, [10] overflow,
        // [11] the lean top-down kernel's cursor, [12..14] routed queries by size class, [16..19] the largest size class of the
        // small teams split four ways (sweep_lean.hip takes the longest jobs first), [20] what the clustered route's top-up phase forwards,
        // [21] k_select_stream's row cursor
        // [32..63] the same for the second set of queues
        if (dev_alloc(ctx, &w.cls_count, 64)) return 1;
        w.route_count = w.cls_count + 8;
        w.slow_count = w.cls_count + 9;
        w.overflow_count = w.cls_count + 10;
        if (alt && set == 0) swap_bufs(w);
    }
    w.has_alt = alt;
    // small teams: one wavefront per query, up to 8 workgroups (32 waves) per CU on 256 CUs;
    // map is n_nodes ints per team (<= ~8 GiB in total), order/S/R share ~16 GiB
    int64_t nn = t.n_nodes;
    // big trees keep a node map of n_nodes ints per team (<= ~8 GiB in total)
    // (2 048 teams are resident at two wavefronts per SIMD; 3 072 measured best at both 10 k and 200 k leaves)
    int64_t teams = (t.scan || sweep_bits_in_lds(t) || sweep_merge_lists(t)) ? 3072 : std::min<int64_t>(3072, std::max<int64_t>(64, ((int64_t)8 << 30) / (4 * nn)));
    teams = std::min<int64_t>(teams, round_up(batch, 4));
    if (knob_on(ctx, "APPLES_SWEEP_TEAMS")) teams = std::max<int64_t>(4, knob(ctx, "APPLES_SWEEP_TEAMS", 0));  // tuning knob
    int wgs_small = (int)std::max<int64_t>(1, teams / 4);
    int64_t per_node = t.scan ? 120 + (xe ? 144 : 0) : 68 + (sweep_merge_lists(t) ? 12 : 0) + (t.max_children > 2 ? 48 : 0) + (xe ? 2 * 144 : 0);
    if (sweep_lean_layout(t, xe)) per_node = LEAN_BYTES_PER_NODE + 4;
    const bool lean_pool = sweep_lean_layout(t, xe);
    int64_t cap = std::min<int64_t>(nn, std::max<int64_t>(1024, ((int64_t)16 << 30) / ((int64_t)wgs_small * 4 * per_node)));
    if (knob_on(ctx, "APPLES_SWEEP_CAP")) cap = std::min<int64_t>(cap, std::max<int64_t>(64, knob(ctx, "APPLES_SWEEP_CAP", 0)));  // test knob: small teams overflow early
    if (lean_pool) {
        // the batch's pool: what its queries ask for at 3 entries per observed leaf, within 12 GiB (queries beyond the
        // pool go to the workgroup-sized teams); APPLES_LEAN_POOL_MB: test knob
        int64_t want = batch * (3 * std::min<int64_t>(members, big_threshold(ctx)) + 1028);  // (sweep_lean.hip:lean_query_cap)
        // (ragged rows: batches three times as long -- 1 300 entries for a query of the clustered route's hundred-odd entries)
        int64_t pool = std::min<int64_t>(want, ((int64_t)(ragged ? 32 : 12) << 30) / LEAN_BYTES_PER_NODE);
        if (knob_on(ctx, "APPLES_LEAN_POOL_MB")) pool = std::max<int64_t>(1024, ((int64_t)knob(ctx, "APPLES_LEAN_POOL_MB", 0) << 20) / LEAN_BYTES_PER_NODE);
        cap = std::min<int64_t>(pool, 0x7ffffff0ll);
    }
    if (alloc_sweep(ctx, w.small, wgs_small, 4, cap, t.scan ? 0 : std::min<int64_t>(members, std::max<int64_t>(cap, big_threshold(ctx))), xe, batch)) return 1;
    // big teams: one workgroup per query with full-size scratch (~24 GiB in total)
    int64_t per_wg = nn * (4 + per_node) + (t.height + 4) * 8 + (sweep_lean_layout(t, xe) ? members * LEAN_BYTES_PER_LEAF + t.poly_kids * LEAN_BYTES_PER_NODE : 0);
    int64_t big_max = knob(ctx, "APPLES_SWEEP_BIG_WGS", 512);
    int wgs_big = (int)std::min<int64_t>(big_max, std::max<int64_t>(4, ((int64_t)24 << 30) / std::max<int64_t>(per_wg, 1)));
    wgs_big = (int)std::min<int64_t>(wgs_big, batch);
    if (alloc_sweep(ctx, w.big, wgs_big, 1, nn, members, xe, batch)) return 1;
    return 0;
}

void free_block(apples_ctx *ctx, QueryBlock *qb) {
    blk_free(ctx, qb->table); blk_free(ctx, qb->raw); blk_free(ctx, qb->packed); blk_free(ctx, qb->qf4);
    blk_free(ctx, qb->aa_idx); blk_free(ctx, qb->aa_mask); blk_free(ctx, qb->self_slot); blk_free(ctx, qb->out);
    blk_free(ctx, qb->sd_q4); blk_free(ctx, qb->sd_nvq);
    blk_free(ctx, qb->packed8); blk_free(ctx, qb->q_ex); blk_free(ctx, qb->ex_idx);
    if (qb->ex_block) { free_block(ctx, qb->ex_block); delete qb->ex_block; }
    *qb = QueryBlock();
}

// Device buffers of a block of n queries (nothing uploaded yet); padding rows are preset on `st`.
int alloc_block(apples_ctx *ctx, int64_t n, const int32_t *self_row, int planes, QueryBlock *qb, hipStream_t st) {
    DevAlign &a = ctx->aln;
    qb->n = n;
    qb->n_pad = round_up(std::max<int64_t>(n, 1), 32);
    if (blk_alloc(ctx, &qb->raw, n * a.L)) return 1;
    std::vector<int32_t> self(std::max<int64_t>(n, 1), -1);
    if (self_row)
        for (int64_t i = 0; i < n; ++i) {
            if (self_row[i] >= a.n_refs) { ctx->err = "self_row out of range"; return 1; }
            self[i] = self_row[i] >= 0 ? a.row_slot[self_row[i]] : -1;
        }
    if (blk_alloc(ctx, &qb->self_slot, (int64_t)self.size())) return 1;
    HIP_TRY(ctx, hipMemcpyAsync(qb->self_slot, self.data(), self.size() * sizeof(int32_t), hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));  // `self` goes out of scope
    if (blk_alloc(ctx, &qb->out, std::max<int64_t>(n, 1))) return 1;
    if (ctx->params.model == APPLES_SCOREDIST) {
        int Lpad = (a.L + 15) / 16 * 16;
        if (blk_alloc(ctx, &qb->aa_idx, qb->n_pad * Lpad)) return 1;
        HIP_TRY(ctx, hipMemsetAsync(qb->aa_idx, 20, (size_t)qb->n_pad * Lpad, st));
        if (blk_alloc(ctx, &qb->aa_mask, qb->n_pad * (Lpad / 16))) return 1;
        HIP_TRY(ctx, hipMemsetAsync(qb->aa_mask, 0, (size_t)qb->n_pad * (Lpad / 16) * 2, st));
        if (a.sd_ref4) {  // operand image for the matrix-core filter (dist_sd.hip), tiled like the reference's
            const int64_t n_img = round_up(qb->n_pad, 256) + 256;  // a sub-batch may start at any multiple of 32
            if (blk_alloc(ctx, &qb->sd_q4, sd_query_image_bytes(ctx, n_img))) return 1;
            if (blk_alloc(ctx, &qb->sd_nvq, n_img)) return 1;
        }
    } else {
        qb->planes = planes;
        int64_t w = qb->n_pad * a.G * (planes + 1);
        if (blk_alloc(ctx, &qb->packed, w)) return 1;
        HIP_TRY(ctx, hipMemsetAsync(qb->packed, 0, (size_t)w * sizeof(uint4), st));
        if (planes == 2 && a.ex_ok) {  // which queries carry a byte beyond ACGT- (k_pack_rows<2>); the 8-plane form where the context keeps one
            if (blk_alloc(ctx, &qb->q_ex, qb->n_pad)) return 1;
            HIP_TRY(ctx, hipMemsetAsync(qb->q_ex, 0, (size_t)qb->n_pad * sizeof(int32_t), st));
            if (a.packed8) {
                const int64_t w8 = qb->n_pad * a.G * 9;
                if (blk_alloc(ctx, &qb->packed8, w8)) return 1;
                HIP_TRY(ctx, hipMemsetAsync(qb->packed8, 0, (size_t)w8 * sizeof(uint4), st));
            }
        }
        if (planes == 2 && dist_mfma_enabled(ctx)) {  // fp4 operand image for the matrix-core distance kernel
            const int64_t n128 = round_up(qb->n_pad, 256) + 256;  // a sub-batch may start at any multiple of 32
            if (blk_alloc(ctx, &qb->qf4, n128 * a.G * 256)) return 1;
        }
    }
    return 0;
}

// Upload queries [q0, q0 + nq) of the block from the caller's buffer and bring them into the device
// layouts, all on `st`.  q0 is a multiple of 32.  Symbols beyond ACGT- raise the context's d_exotic flag
// (2-plane packing only); the caller reads it.
int fill_block(apples_ctx *ctx, QueryBlock *qb, const uint8_t *queries, int64_t q0, int64_t nq, hipStream_t st) {
    DevAlign &a = ctx->aln;
    if (nq <= 0) return 0;
    HIP_TRY(ctx, hipMemcpyAsync(qb->raw + q0 * a.L, queries + q0 * a.L, (size_t)nq * a.L, hipMemcpyHostToDevice, st));
    if (ctx->params.model == APPLES_SCOREDIST) {
        int Lpad = (a.L + 15) / 16 * 16;
        if (launch_pack_aa(ctx, qb->raw + q0 * a.L, nq, a.L, qb->aa_idx + q0 * Lpad, qb->aa_mask + q0 * (Lpad / 16), 0, true, st)) return 1;
        if (qb->sd_q4) {
            const int64_t n_img = round_up(qb->n_pad, 256) + 256;
            const bool last = q0 + nq >= qb->n;  // the last chunk also zeroes the image's padding rows
            if (launch_sd_expand(ctx, qb->raw + q0 * a.L, nq, qb->sd_q4, last ? n_img - q0 : nq, st, nullptr, q0, true, qb->sd_nvq, nullptr,
                                 qb->aa_mask + q0 * (Lpad / 16), 0)) return 1;
        }
        return 0;
    }
    if (launch_pack_rows(ctx, qb->raw + q0 * a.L, nq, a.L, qb->planes, qb->packed + q0 * a.G * (qb->planes + 1), 0, true,
                         ctx->d_exotic, st, nullptr, qb->q_ex ? qb->q_ex + q0 : nullptr)) return 1;
    if (qb->packed8 && launch_pack_rows(ctx, qb->raw + q0 * a.L, nq, a.L, 8, qb->packed8 + q0 * a.G * 9, 0, true, nullptr, st)) return 1;
    if (qb->qf4) {
        const int64_t n128 = round_up(qb->n_pad, 256) + 256;
        const bool last = q0 + nq >= qb->n;  // the last chunk also zeroes the image's padding rows
        if (a.ref_f4) {  // compact, tiled images beside a reference image: addressed by image row
            if (launch_expand_queries_f4(ctx, qb->raw + q0 * a.L, nq, qb->qf4, last ? n128 - q0 : nq, st, nullptr, q0)) return 1;
        } else if (launch_expand_queries_f4(ctx, qb->raw + q0 * a.L, nq, qb->qf4 + q0 * (int64_t)a.G * 256, last ? n128 - q0 : nq, st)) return 1;
    }
    return 0;
}

// read and clear the exotic-symbol flag (after the packing on `st` has drained)
int take_exotic(apples_ctx *ctx, hipStream_t st, int *exotic) {
    *exotic = 0;
    HIP_TRY(ctx, hipMemcpyAsync(exotic, ctx->d_exotic, sizeof(int), hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    if (*exotic) {
        HIP_TRY(ctx, hipMemsetAsync(ctx->d_exotic, 0, sizeof(int), st));
        HIP_TRY(ctx, hipStreamSynchronize(st));
    }
    return 0;
}

int make_block(apples_ctx *ctx, const uint8_t *queries, int64_t n, const int32_t *self_row, QueryBlock *qb);

// A packed 2-plane block of an ex_ok context whose queries carry bytes beyond ACGT- (the context's flag said so): the reference's
// 8-plane form comes into being, the block gets its own, and the queries concerned are placed by the 8-plane kernels -- a few of
// them: as a block of their own (ex_block: placed behind the main pass, whose placements for them are overwritten); many (more
// than a twentieth of the block): the whole block (exact8: no matrix-core pass for it).  `queries` / `self_row`: the caller's.
int exotic_queries(apples_ctx *ctx, QueryBlock *qb, const uint8_t *queries, const int32_t *self_row) {
    DevAlign &a = ctx->aln;
    if (ensure_packed8(ctx)) return 1;
    if (!qb->packed8) {
        const int64_t w8 = qb->n_pad * a.G * 9;
        if (blk_alloc(ctx, &qb->packed8, w8)) return 1;
        HIP_TRY(ctx, hipMemsetAsync(qb->packed8, 0, (size_t)w8 * sizeof(uint4), ctx->stream));
        if (launch_pack_rows(ctx, qb->raw, qb->n, a.L, 8, qb->packed8, 0, true, nullptr)) return 1;
    }
    std::vector<int32_t> flag((size_t)qb->n), idx;
    HIP_TRY(ctx, hipMemcpyAsync(flag.data(), qb->q_ex, (size_t)qb->n * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    for (int64_t i = 0; i < qb->n; ++i)
        if (flag[i]) idx.push_back((int32_t)i);
    if (idx.empty()) return 0;
    if ((int64_t)idx.size() > std::min<int64_t>(std::max<int64_t>(16, qb->n / 20), 2048)) { qb->exact8 = true; return 0; }
    std::vector<uint8_t> rows(idx.size() * (size_t)a.L);
    std::vector<int32_t> self(idx.size(), -1);
    for (size_t k = 0; k < idx.size(); ++k) {
        memcpy(rows.data() + k * (size_t)a.L, queries + (int64_t)idx[k] * a.L, (size_t)a.L);
        if (self_row) self[k] = self_row[idx[k]];
    }
    if (qb->ex_block) { free_block(ctx, qb->ex_block); delete qb->ex_block; qb->ex_block = nullptr; }
    blk_free(ctx, qb->ex_idx); qb->ex_idx = nullptr;
    qb->ex_block = new QueryBlock();
    if (alloc_block(ctx, (int64_t)idx.size(), self_row ? self.data() : nullptr, 2, qb->ex_block, ctx->stream)) return 1;
    if (fill_block(ctx, qb->ex_block, rows.data(), 0, (int64_t)idx.size(), ctx->stream)) return 1;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));  // (rows is a local)
    int dummy = 0;
    if (take_exotic(ctx, ctx->stream, &dummy)) return 1;  // (the flag these rows raised again)
    qb->ex_block->exact8 = true;
    qb->ex_block->live = true;
    if (blk_alloc(ctx, &qb->ex_idx, (int64_t)idx.size())) return 1;
    HIP_TRY(ctx, hipMemcpy(qb->ex_idx, idx.data(), idx.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    return 0;
}

// whole block at once on the context's stream (apples_queries_upload)
int make_block(apples_ctx *ctx, const uint8_t *queries, int64_t n, const int32_t *self_row, QueryBlock *qb) {
    DevAlign &a = ctx->aln;
    for (int attempt = 0; attempt < 2; ++attempt) {
        if (alloc_block(ctx, n, self_row, a.planes, qb, ctx->stream)) return 1;
        if (fill_block(ctx, qb, queries, 0, n, ctx->stream)) return 1;
        if (ctx->params.model == APPLES_SCOREDIST) break;
        int exotic = 0;
        if (take_exotic(ctx, ctx->stream, &exotic)) return 1;
        if (!(exotic && qb->planes == 2)) break;
        if (a.ex_ok) {  // the context keeps its matrix-core forms: the queries with such bytes take the 8-plane forms
            if (exotic_queries(ctx, qb, queries, self_row)) return 1;
            break;
        }
        free_block(ctx, qb);  // symbols beyond ACGT-: widen the reference, pack again with 8 planes
        if (repack_to_bytes(ctx)) return 1;
    }
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    qb->live = true;
    return 0;
}

struct PhaseTimer {
    apples_ctx *ctx;
    double acc[APPLES_T_COUNT] = {};
    void begin(int k) { (void)hipEventRecord(ctx->ev[2 * (k & 3)], ctx->stream); }
    void end(int k) { (void)hipEventRecord(ctx->ev[2 * (k & 3) + 1], ctx->stream); pending[k & 3] = true; kind[k & 3] = k; }
    bool pending[4] = {false, false, false, false};
    int kind[4] = {0, 0, 0, 0};
    void flush() {
        for (int i = 0; i < 4; ++i)
            if (pending[i]) {
                (void)hipEventSynchronize(ctx->ev[2 * i + 1]);
                float ms = 0;
                (void)hipEventElapsedTime(&ms, ctx->ev[2 * i], ctx->ev[2 * i + 1]);
                acc[kind[i]] += ms;
                pending[i] = false;
            }
    }
};

SelectArgs select_args_alignment(apples_ctx *ctx, const QueryBlock &qb, int64_t q0) {
    const DevAlign &a = ctx->aln;
    Workspace &w = ctx->ws;
    SelectArgs s{};
    s.dist = w.dist; s.stride = a.slots_pad; s.gather = nullptr;
    s.slot_node = a.slot_node; s.slot_level = a.slot_level; s.slot_rep = a.slot_rep; s.slot_mpos = a.slot_mpos;
    s.rep_slot = a.rep_slot; s.rep_moff = a.rep_moff; s.mem_slot = a.mem_slot;
    s.n_members = a.n_refs; s.n_reps = a.n_reps; s.all_singleton = a.all_singleton ? 1 : 0; s.table_mode = 0;
    s.self_slot = qb.self_slot + q0;
    s.thr = ctx->params.filt_threshold; s.baseobs = ctx->params.base_observation; s.height = ctx->tree.height;
    s.obs_node = w.obs_node; s.obs_dist = w.obs_dist; s.obs_cap = w.obs_cap; s.n_obs = w.n_obs;
    if (w.ragged) {  // (common.h, Workspace::ragged)
        s.row_off = w.row_off; s.row_small = w.row_small; s.row_big_base = w.batch * w.row_small; s.row_big_pitch = w.stride; s.row_big_n = (int32_t)w.row_big;
        s.row_big_cursor = w.cls_count + 22; s.row_fail = ctx->d_rag;
    }
    s.cnt_gt = ctx->tree.scan ? nullptr : w.cnt_gt;  // (the scan sweep takes its leaves in node-id order and needs no per-level offsets)
    s.out = qb.out + q0;
    s.big_threshold = route_threshold(ctx); s.overflow_list = w.route_list; s.overflow_count = w.route_count;
    s.route_classes = (w.big.lean && !hybrid_records(ctx)) ? 1 : 0;  // (what run_sweep's launch_big will run)
    s.cls_list = w.cls_list; s.cls_count = w.cls_count; s.cls_stride = w.batch;
    s.row_cursor = w.cls_count + 21;
    s.seg_slot = w.seg_slot; s.seg_cnt = w.seg_cnt; s.node_level = ctx->tree.level;
    s.slow_list = w.slow_list; s.slow_count = w.slow_count; s.qlist = nullptr; s.qcount = nullptr;
    s.slow_hint = nullptr; s.qhint = nullptr;
    s.seg_lut = nullptr;
    return s;
}

SweepArgs sweep_args(apples_ctx *ctx, const Workspace::Sweep &sw, apples_placement *out, bool keep_edges) {
    Workspace &w = ctx->ws;
    SweepArgs s{};
    s.tree = ctx->tree;
    s.obs_node = w.obs_node; s.obs_dist = w.obs_dist; s.obs_cap = w.obs_cap; s.cnt_gt = w.cnt_gt; s.n_obs = w.n_obs;
    s.row_off = w.ragged ? w.row_off : nullptr;
    s.grp_off = sw.grp_off; s.A = sw.A; s.B = sw.B; s.xe = sw.xe;
    s.grp_stride = (int)lean_grp_stride(ctx->tree);
    s.map = sw.map; s.map_ver = sw.ver; s.order = sw.order; s.ent = sw.ent;
    s.lean = sw.lean; s.lean_cap1 = sw.lean_cap1; s.lean_leaf1 = sw.lean_leaf1;
    s.lean_leaf = sw.lean_leaf; s.lean_teams = sw.teams; s.lean_meta = sw.lean_meta; s.pool_cursor = (unsigned int *)(w.cls_count + 7);
    s.prof = ctx->lean_prof;
    s.blk_pool = ctx->blk_active ? ctx->blk_pool : nullptr;  // (clade blocks in this device batch's observation lists: run_block)
    s.map_bits = 1;
    while ((1u << s.map_bits) <= 2u * ((uint32_t)ctx->tree.n_nodes + 2u)) ++s.map_bits;
    if (knob_on(ctx, "APPLES_MAP_BITS")) s.map_bits = std::min(30, std::max(s.map_bits, (int)knob(ctx, "APPLES_MAP_BITS", 0)));  // test knob: few tags, early wrap
    s.cap = sw.cap;
    s.leaf_cap = sw.leaf_cap;
    s.method = ctx->params.method; s.criterion = ctx->params.criterion; s.negative = ctx->params.negative_branch;
    // (HYBRID on a level-loop workspace -- a tree the lean sweep does not serve, or one regrown with records for apples_sweep_edges --
    // ranks the per-edge records; the lean sweep ranks what it keeps in its entries)
    s.keep_edges = (keep_edges || (ctx->params.criterion == APPLES_HYBRID && (hybrid_records(ctx) || !sw.lean))) ? 1 : 0;
    const int dbg = (int)knob(ctx, "APPLES_SWEEP_DEBUG_PHASE", 0);
    s.debug_phase = dbg;
    s.work_list = nullptr; s.work_count = nullptr; s.big_threshold = route_threshold(ctx);
    s.cls_list = nullptr; s.cls_count = nullptr; s.cls_stride = w.batch; s.cursor = w.cls_count + 4;
    s.overflow_list = w.overflow_list; s.overflow_count = w.overflow_count;
    s.out = out;
    return s;
}

ScanArgs scan_args(apples_ctx *ctx, const Workspace::Sweep &sw, apples_placement *out, bool keep_edges, bool big) {
    Workspace &w = ctx->ws;
    const DevTree &t = ctx->tree;
    ScanArgs s{};
    s.leaf_info = t.leaf_info; s.anc = t.anc; s.rmq = t.rmq; s.euler_len = t.euler_len; s.n_nodes = t.n_nodes; s.height = t.height;
    s.obs_node = w.obs_node; s.obs_dist = w.obs_dist; s.obs_cap = w.obs_cap; s.n_obs = w.n_obs;
    s.row_off = w.ragged ? w.row_off : nullptr;
    s.ent_f = sw.ent_f; s.ent_i = sw.ent_i; s.xe = sw.xe; s.leaf_g = sw.leaf_g; s.meta = sw.meta;
    s.cap = sw.cap; s.leaf_cap = sw.leaf_cap;
    s.lds_leaves = big ? SCAN_LDS_LEAVES_BIG : SCAN_LDS_LEAVES_SMALL;
    s.method = ctx->params.method; s.criterion = ctx->params.criterion; s.negative = ctx->params.negative_branch;
    s.keep_edges = (keep_edges || hybrid_records(ctx)) ? 1 : 0;
    s.big_threshold = big_threshold(ctx);
    s.work_list = nullptr; s.work_count = nullptr; s.cls_list = nullptr; s.cls_count = nullptr; s.cls_stride = w.batch;
    s.cursor = w.cls_count + 4;
    s.overflow_list = w.overflow_list; s.overflow_count = w.overflow_count;
    s.out = out;
    s.prof = ctx->scan_prof;
    return s;
}

// the same launch structure as run_sweep below, for the scan formulation (sweep_scan.hip)
int run_scan(apples_ctx *ctx, apples_placement *out, int64_t nq, hipStream_t st) {
    Workspace &w = ctx->ws;
    const int small_team = (int)knob(ctx, "APPLES_SWEEP_TEAM", 64);  // tuning knob
    ScanArgs b = scan_args(ctx, w.big, out, false, true);
    b.overflow_list = nullptr;  // a big team's scratch holds the whole tree and any number of leaves
    b.overflow_count = nullptr;
    b.cursor = w.cls_count + 5;
    if (small_team != 64) {  // diagnostic mode: workgroup-sized teams for everything
        b.cursor = w.cls_count + 4;
        return launch_scan(ctx, b, nq, w.big.wgs, 256, st);
    }
    b.work_list = w.route_list;
    b.work_count = w.route_count;
    HIP_TRY(ctx, hipMemsetAsync(w.overflow_count, 0, sizeof(int32_t), st));
    ScanArgs sm = scan_args(ctx, w.small, out, false, false);
    sm.cls_list = w.cls_list;
    sm.cls_count = w.cls_count;
    sm.cursor = w.cls_count + 4;
    if (launch_scan_mixed(ctx, sm, b, nq, w.small.wgs, w.big.wgs, st)) return 1;
    if (w.small.cap >= ctx->tree.n_nodes) return 0;  // a small team's arrays hold any subtree: nothing can overflow
    b.work_list = w.overflow_list;
    b.work_count = w.overflow_count;
    b.cursor = w.cls_count + 6;
    return launch_scan(ctx, b, nq, w.big.wgs, 256, st);
}

// The sweep for one device batch: wavefront-sized teams first, then workgroup-sized teams with
// full-size scratch for the queries whose induced subtree did not fit (usually none).
// `st` = the stream the small-team sweep (and the final overflow launch) runs on; the caller has
// made `st` wait for the selection kernel of this batch.
// `part`: 0 = all of it; 1 = everything but the closing launch for the overflow list and 2 = that launch alone (run_block's
// overlapped top-up chain appends to the list between the two; the caller has cleared the list's counter)
int run_sweep(apples_ctx *ctx, apples_placement *out, int64_t nq, hipStream_t st = nullptr, int part = 0) {
    Workspace &w = ctx->ws;
    if (!st) st = ctx->stream;
    if (ctx->tree.scan) return run_scan(ctx, out, nq, st);
    const int small_team = (int)knob(ctx, "APPLES_SWEEP_TEAM", 64);  // tuning knob
    SweepArgs b = sweep_args(ctx, w.big, out, false);
    b.overflow_list = nullptr;  // a big team's scratch holds the whole tree: it cannot overflow
    b.overflow_count = nullptr;
    b.cursor = w.cls_count + 5;
    // workgroup-sized teams: sweep_lean.hip's where the workspace has its field arrays (big binary trees), else the level loop
    auto launch_big = [&](const SweepArgs &x, hipStream_t s) {
        return (w.big.lean && !x.keep_edges) ? launch_sweep_lean_big(ctx, x, nq, w.big.wgs, s) : launch_sweep(ctx, x, nq, w.big.wgs, 256, s);
    };
    if (small_team != 64) {  // diagnostic mode: workgroup-sized teams for everything
        b.cursor = w.cls_count + 4;
        return launch_big(b, st);
    }
    // one launch: the first w.big.wgs workgroups first serve the queries the selection kernel
    // routed to workgroup-sized teams (many observed leaves), then all workgroups split into
    // wavefront-sized teams for the size-class queues
    b.work_list = w.route_list;
    b.work_count = w.route_count;
    b.route_classes = (w.big.lean && !b.keep_edges) ? 1 : 0;  // (as the selection kernels filed them: select_args_alignment)
    const bool can_overflow = w.small.cap < ctx->tree.n_nodes || w.small.lean_leaf != nullptr;  // (pool: a query may ask for more than its share)
    if (can_overflow && part == 0) HIP_TRY(ctx, hipMemsetAsync(w.overflow_count, 0, sizeof(int32_t), st));
    SweepArgs sm = sweep_args(ctx, w.small, out, false);
    sm.cls_list = w.cls_list;  // size-class queues written by the selection kernels, largest first
    sm.cls_count = w.cls_count;
    sm.cursor = w.cls_count + 4;
    if (part == 2) {
        // (nothing of the first part to launch)
    } else if (w.small.lean && !sm.keep_edges) {
        // big binary trees: the wavefront-sized teams run sweep_lean.hip; the queries routed to workgroup-sized teams
        // (many observed leaves: the longest jobs) run beside them.  Their launch goes first and on this stream, the lean
        // kernel on a second stream behind an event: its persistent workgroups would otherwise take every slot of the
        // chip and the long jobs would start when the short ones are done (measured: 3.3 + 3.1 ms per launch instead
        // of the two side by side)
        HIP_TRY(ctx, hipEventRecord(ctx->ev_sel, st));
        if (launch_big(b, st)) return 1;
        HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream_big, ctx->ev_sel, 0));
        SweepArgs down = sm;
        down.cursor = w.cls_count + 11;
        // APPLES_LEAN_HALVES=1: a small device batch takes its queue in two halves, the first half's top-down kernel beside the second
        // half's bottom-up kernel.  Measured SLOWER (round 5, profiles/r05_lean_halves_exp.txt: config 3's 12 500-query shards 6.6 - 6.8 ->
        // 7.0 - 7.3 ms, config 5's block 3.28 -> 3.74, config 2 2.38 -> 2.77): four short launches have four tails.  Off by default.
        const bool want_halves = knob_on(ctx, "APPLES_LEAN_HALVES");  // experiment knob
        const bool halves = want_halves && nq <= LEAN_SMALL_BATCH && nq >= 2048;
        if (halves && !ctx->stream3) HIP_TRY(ctx, hipStreamCreate(&ctx->stream3));
        if (launch_sweep_lean(ctx, sm, down, nq, ctx->stream_big, halves ? w.cls_count + 22 : nullptr, halves ? ctx->stream3 : nullptr,
                              ctx->ev_half)) return 1;
        HIP_TRY(ctx, hipEventRecord(ctx->ev_big, ctx->stream_big));
        HIP_TRY(ctx, hipStreamWaitEvent(st, ctx->ev_big, 0));
    } else if (launch_sweep_mixed(ctx, sm, b, nq, w.small.wgs, w.big.wgs, st)) return 1;
    // whatever did not fit a small team's scratch (usually nothing; nothing at all when that scratch
    // holds the whole tree: the overflow test in the kernel is `cap < n_nodes && ...`)
    if (!can_overflow || part == 1) return 0;
    b.work_list = w.overflow_list;
    b.work_count = w.overflow_count;
    b.route_classes = 0;
    b.cursor = w.cls_count + 6;
    return launch_big(b, st);
}

// The top-up chain's own sweep (run_block): the wavefront-sized teams of sweep_lean.hip over the SECOND set of size-class
// queues (what the chain's selection kernels enlisted), with per-leaf scratch of their own, on `st` beside the batch's main
// sweep.  Pool, per-query records and the overflow list are the main sweep's (atomic cursors).
int run_sweep_second(apples_ctx *ctx, apples_placement *out, int64_t nq, hipStream_t st) {
    Workspace &w = ctx->ws;
    SweepArgs sm = sweep_args(ctx, w.small, out, false);
    sm.cls_list = w.cls_list + 8 * w.batch;
    sm.cls_count = w.cls_count + 32;
    sm.cursor = w.cls_count + 32 + 4;
    sm.lean_leaf = reinterpret_cast<char *>(w.small.lean_leaf) + w.small.teams * w.small.lean_leaf1 * LEAN_BYTES_PER_LEAF;
    SweepArgs down = sm;
    down.cursor = w.cls_count + 32 + 11;
    return launch_sweep_lean(ctx, sm, down, nq, st);
}

// every launch_* below reads ctx->stream when it is called: a chain of them on another stream for the length of a scope
struct StreamScope {
    apples_ctx *ctx;
    hipStream_t saved;
    StreamScope(apples_ctx *c, hipStream_t s) : ctx(c), saved(c->stream) { c->stream = s; }
    ~StreamScope() { ctx->stream = saved; }
};

int dist_tile_for(const apples_ctx *ctx, int64_t nq) {
    const int forced = (int)knob(ctx, "APPLES_DIST_TILE", 0);  // tuning knob
    if (forced > 0) return forced;
    // 16 queries per reference pass: 32 accumulators + 12 reference words stay within 96 VGPRs
    // (5 waves/SIMD); 32 queries per pass spill the epilogue to 150 VGPRs and run slower
    return nq >= 16 ? 16 : (nq >= 8 ? 8 : (nq >= 4 ? 4 : 1));
}

// One pass of the hot path over a resident query block.  The block is cut into sub-batches that
// run distance -> selection -> sweep back to back on one stream (default).  With APPLES_PIPELINE > 1
// they flow through two streams instead (front = distance + selection, back = sweep, two sets of
// batch buffers); measured slower on MI355X, kept as an experiment knob.
// `feed` (optional): the block's queries are still in the caller's host buffer.  Chunk i = the queries of
// sub-batch i is uploaded and packed on stream2 while the main stream works on sub-batch i - 1, so that
// the copy hides behind the kernels (apples_place_from_sequences: host buffer in, host buffer out).
struct Feeder {
    const uint8_t *host;
};

// the back stream of the pipeline experiments, made on first use
int back_stream(apples_ctx *ctx) {
    if (!ctx->stream3) HIP_TRY(ctx, hipStreamCreate(&ctx->stream3));
    return 0;
}

int run_block_main(apples_ctx *ctx, QueryBlock &qb, const Feeder *feed);

// ragged rows (common.h, Workspace::ragged): every query of the device batch starts on its small row
__global__ void k_row_init(int64_t *__restrict__ row_off, int64_t nq, int64_t row_small) {
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q < nq) row_off[q] = q * row_small;
}

__global__ void k_scatter_placements(apples_placement *__restrict__ out, const apples_placement *__restrict__ src,
                                     const int32_t *__restrict__ idx, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[idx[i]] = src[i];
}

// the block's queries with bytes beyond ACGT- (QueryBlock::ex_block, exotic_queries) through the 8-plane kernels, their
// placements over what the main pass left for them; the timers add up
int run_ex_block(apples_ctx *ctx, QueryBlock &qb) {
    if (!qb.ex_block) return 0;
    double t0[APPLES_T_COUNT];
    for (int i = 0; i < APPLES_T_COUNT; ++i) t0[i] = ctx->t_ms[i];
    QueryBlock &x = *qb.ex_block;
    if (run_block_main(ctx, x, nullptr)) return 1;
    hipLaunchKernelGGL(k_scatter_placements, dim3((unsigned)((x.n + 255) / 256)), dim3(256), 0, ctx->stream, qb.out, x.out, qb.ex_idx, x.n);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    for (int i = 0; i < APPLES_T_COUNT; ++i) ctx->t_ms[i] += t0[i];
    return 0;
}

int run_block(apples_ctx *ctx, QueryBlock &qb, const Feeder *feed = nullptr) {
    if (run_block_main(ctx, qb, feed)) return 1;
    return run_ex_block(ctx, qb);
}

int run_block_main(apples_ctx *ctx, QueryBlock &qb, const Feeder *feed) {
    const DevAlign &a = ctx->aln;
    const bool hybrid = hybrid_records(ctx);  // (HYBRID on the lean sweep is a pass like any other)
    // fused path: threshold compaction in the distance kernel's epilogue; only queries that need
    // the top-up rule get full distance rows
    const bool no_fuse = (ctx->dbg & APPLES_DBG_NO_FUSE) != 0;  // diagnostic switch
    const int n_pipe = (int)knob(ctx, "APPLES_PIPELINE", 1);  // >1: measured slower (sweep and distance kernels contend), kept as a knob
    // clustered references: the matrix-core pass runs over the representatives only, k_select_clusters expands
    // the accepted clusters (needs the panels of setup_alignment, the tabulated distances and their integer
    // threshold form)
    // (scoredist contexts, csd: full rows of the distances to the representatives in place of the matrix-core pass, k_cluster_dist_sd
    // for the members)
    const bool csd = !no_fuse && !a.all_singleton && ctx->params.model == APPLES_SCOREDIST && a.aa_rep_idx;
    const bool cfused = csd || (!no_fuse && !a.all_singleton && ctx->params.model == APPLES_JC69 && a.rep_packed && a.packed_rm &&
                                fused_counts_format(ctx, qb));
    // scoredist with singleton clusters: threshold compaction in the distance kernel's epilogue as well
    const bool sfused = !no_fuse && a.all_singleton && ctx->params.model == APPLES_SCOREDIST;
    const bool fused = !no_fuse && (((a.all_singleton || cfused) && ctx->params.model == APPLES_JC69) || sfused || csd);
    const bool pipelined = n_pipe > 1 && qb.n >= 1024;
    int64_t want = qb.n;
    if (pipelined) want = round_up((qb.n + n_pipe - 1) / n_pipe, 32);
    // (clustered references on the fused route keep full rows per query: with their member distances parked in the tail of
    // the query's own observation row and 4 device batches of 25 000 in place of 7 of 14 300 the pass measured 80.5 ms
    // against 69.2 -- lists of 3 126 observed leaves per query outgrow the Infinity Cache in the bigger batches -- and the
    // selection kernel's load schedule suffered from the second home of the distances: 4.0 against 2.6 ms per batch)
    const bool slim = fused && a.all_singleton && fused_counts_format(ctx, qb) && !pipelined;
    if (ensure_workspace(ctx, a.n_refs, a.slots_pad, want, true, false, hybrid, fused, pipelined, slim,
                         (cfused && !pipelined && !(ctx->dbg & APPLES_DBG_NO_CLUSTER_TOPUP)) ? a.reps_pad : 0,
                         cfused && !pipelined && !hybrid && !(ctx->dbg & APPLES_DBG_CLUSTER_BY_QUERY))) return 1;
    Workspace &w = ctx->ws;
    if (w.ragged) HIP_TRY(ctx, hipMemsetAsync(ctx->d_rag, 0, 2 * sizeof(int32_t), ctx->stream));
    int64_t step = pipelined ? std::min<int64_t>(w.batch, want) : w.batch;
    const int64_t n_sub = (qb.n + step - 1) / step;
    if (!pipelined && n_sub > 1) step = std::min(step, round_up((qb.n + n_sub - 1) / n_sub, 32));  // equal sub-batches
    // (the GEMM-form distance pass reads whole contiguous kilobytes when a sub-batch starts on a 256-row image tile)
    if (!pipelined && n_sub > 1 && step >= 2048 && dist_gemm_usable(ctx) && round_up(step, 256) <= w.batch) step = round_up(step, 256);
    if (pipelined && back_stream(ctx)) return 1;
    hipStream_t front = ctx->stream, back = pipelined ? ctx->stream3 : ctx->stream;
    // timing events come from a pool that lives with the context (creating and destroying a dozen
    // events per call costs host time inside every pass)
    while (ctx->ev_pool.size() < (size_t)n_sub * 8 + 2) {
        hipEvent_t e;
        HIP_TRY(ctx, hipEventCreate(&e));
        ctx->ev_pool.push_back(e);
    }
    hipEvent_t *ev = ctx->ev_pool.data() + 2;
    hipEvent_t e_start = ctx->ev_pool[0], e_stop = ctx->ev_pool[1];
    if (n_sub == 0) feed = nullptr;  // an empty block: nothing to upload, nothing to place
    // A chunk of a host buffer travels in two pieces where the distance pass can start on the first (the matrix-core passes
    // take any range of query rows): the first quarter is on the device and under way while the rest is still on the bus
    // (a 12 500-query chunk of config 3: 0.22 ms of copy + 0.08 ms of packing in front of 3.6 ms of GEMM).  head(nq) = rows of
    // the first piece (nq: one piece).
    const bool split_feed = feed && fused && !cfused && !pipelined && ((sfused && qb.sd_q4 && sd_gemm_usable(ctx)) ||
                                                                        (!sfused && fused_counts_format(ctx, qb) && dist_gemm_usable(ctx)));
    auto head = [&](int64_t nq) -> int64_t { return split_feed && nq >= 4096 ? round_up(nq / 4, 256) : nq; };
    // upload + pack sub-batch i on stream2, an event behind each piece.  The copy of a pageable host buffer holds the calling
    // thread until the bytes are on their way: the first chunk's second piece is therefore sent (feed_rest) only after the
    // kernels of its first piece have been launched -- sent right behind the first piece, nothing reached the main stream before
    // the whole chunk had gone (config 2's timeline: the distance pass started 0.2 ms after its first piece was ready).
    bool rest_pending = false;
    auto feed_head = [&](int64_t i) -> int {
        const int64_t q0 = i * step, nq = std::min(step, qb.n - q0), h = head(nq);
        if (fill_block(ctx, &qb, feed->host, q0, h, ctx->stream2)) return 1;
        HIP_TRY(ctx, hipEventRecord(ctx->ev_feed[2 * i], ctx->stream2));
        rest_pending = h < nq;
        if (!rest_pending) HIP_TRY(ctx, hipEventRecord(ctx->ev_feed[2 * i + 1], ctx->stream2));
        return 0;
    };
    auto feed_rest = [&](int64_t i) -> int {
        if (!rest_pending) return 0;
        const int64_t q0 = i * step, nq = std::min(step, qb.n - q0), h = head(nq);
        rest_pending = false;
        if (fill_block(ctx, &qb, feed->host, q0 + h, nq - h, ctx->stream2)) return 1;
        HIP_TRY(ctx, hipEventRecord(ctx->ev_feed[2 * i + 1], ctx->stream2));
        return 0;
    };
    auto feed_chunk = [&](int64_t i) -> int { return feed_head(i) || feed_rest(i); };
    if (feed) {
        while (ctx->ev_feed.size() < (size_t)n_sub * 2) {
            hipEvent_t e;
            HIP_TRY(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
            ctx->ev_feed.push_back(e);
        }
        if (feed_head(0)) return 1;
    }
    HIP_TRY(ctx, hipEventRecord(e_start, front));
    if (pipelined) HIP_TRY(ctx, hipStreamWaitEvent(back, e_start, 0));
    int launches = 0;
    std::vector<char> sd_filter_timed((size_t)n_sub, 0), blk_timed((size_t)n_sub, 0);  // per sub-batch: e[5] was recorded (the scoredist filter ran there)
    // The queries on the top-up / slow list get full distance rows; a slim workspace holds rows for a slice of the batch
    // only: the list's length comes to the host (one short wait per batch) and the list is walked in slices of that many
    // queries.  fn(list, count pointer, entries at most) runs the listed distance pass + selection for one slice.
    auto for_slow_slices = [&](int64_t nq, auto fn, const int32_t *list = nullptr, const int32_t *count = nullptr) -> int {
        Workspace &w = ctx->ws;
        hipStream_t front = ctx->stream;
        if (!list) { list = w.slow_list; count = w.slow_count; }
        if (w.dist_rows >= nq) return fn(list, count, nq);
        int32_t hcnt = 0;
        HIP_TRY(ctx, hipMemcpyAsync(&hcnt, count, sizeof(int32_t), hipMemcpyDeviceToHost, front));
        HIP_TRY(ctx, hipStreamSynchronize(front));
        const int64_t R = w.dist_rows;
        int32_t lens[64];
        int n_sl = 0;
        for (int64_t off = 0; off < hcnt && n_sl < 64; off += R) lens[n_sl++] = (int32_t)std::min<int64_t>(R, hcnt - off);
        if ((int64_t)n_sl * R < hcnt) { ctx->err = "top-up list longer than 64 slices of the batch's full rows"; return 1; }
        if (n_sl > 1) {
            if (!ctx->d_slice_cnt && dev_alloc(ctx, &ctx->d_slice_cnt, 64)) return 1;
            HIP_TRY(ctx, hipMemcpyAsync(ctx->d_slice_cnt, lens, n_sl * sizeof(int32_t), hipMemcpyHostToDevice, front));
            HIP_TRY(ctx, hipStreamSynchronize(front));  // (lens is on the stack)
        }
        for (int k = 0; k < n_sl; ++k)
            if (fn(list + (int64_t)k * R, n_sl > 1 ? ctx->d_slice_cnt + k : count, lens[k])) return 1;
        return 0;
    };
    for (int64_t i = 0; i < n_sub; ++i) {
        const int64_t q0 = i * step;
        const int64_t nq = std::min(step, qb.n - q0);
        const int set = (int)(i & 1);
        if (pipelined && i > 0) swap_bufs(w);  // host view: w.* now names buffer set `set`
        if (pipelined && i >= 2) HIP_TRY(ctx, hipStreamWaitEvent(front, ctx->ev_back[set], 0));  // set free again
        hipEvent_t *e = &ev[(size_t)i * 8];
        ctx->cur_batch_queries = nq;  // (route_threshold)
        // rows of the chunk's first piece.  Only the first chunk is still on the bus when its kernels could start: chunk i + 1
        // travels while batch i runs and is there, whole, when its turn comes -- one launch then, not two
        const int64_t nh = (feed && i == 0) ? head(nq) : nq;
        if (feed) HIP_TRY(ctx, hipStreamWaitEvent(front, ctx->ev_feed[2 * i + (nh < nq ? 0 : 1)], 0));  // chunk i (its first piece) is on the device
        HIP_TRY(ctx, hipMemsetAsync(w.cls_count, 0, 64 * sizeof(int32_t), front));  // every counter of the batch
        if (w.ragged) {  // every query's rows: the small ones (common.h, Workspace::ragged)
            hipLaunchKernelGGL(k_row_init, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, front, w.row_off, nq, w.row_small);
            HIP_TRY(ctx, hipGetLastError());
        }
        // The top-up chain -- full rows (or rows of bounds) for the queries k_select_fast listed, their selection -- runs BESIDE the
        // sweep of the queries k_select_fast placed on its own: on stream2, its selection kernels filing into the second set of
        // size-class queues, which a second launch of the wavefront-sized teams serves (run_sweep_second); what would be routed to
        // workgroup-sized teams goes to the overflow list, whose launch closes the batch.  chain() = the chain's launches (they
        // read ctx->stream, redirected for the scope).  Sets swept_here when the batch's sweep has been launched in here.
        // Small device batches only (a shard of a multi-GPU job): there the chain's short launches find idle compute units beside
        // the sweep (two of config 3's 12 500-query shards: 6.99 -> 6.79 and 7.02 -> 6.92 ms); in a full-size batch the sweep's
        // persistent workgroups hold every slot until their queues are empty, the chain runs after them all the same and the two
        // sweeps get in each other's way (config 3: 48.1 -> 49.7 ms, config 4: 23.0 -> 23.6).
        bool swept_here = false;
        auto overlapped = [&](SelectArgs &sa, auto chain) -> int {
            const bool on = !pipelined && !hybrid && !(ctx->dbg & APPLES_DBG_NO_TOPUP_OVERLAP) && !ctx->tree.scan && w.small.lean &&
                            w.small.lean_leaf && w.big.lean && nq <= LEAN_SMALL_BATCH;
            if (!on) {
                if (chain()) return 1;
                HIP_TRY(ctx, hipEventRecord(e[2], front));
                return 0;
            }
            HIP_TRY(ctx, hipEventRecord(e[2], front));  // (the selection phase of the timers: k_select_fast alone)
            HIP_TRY(ctx, hipEventRecord(ctx->ev_top[0], front));
            HIP_TRY(ctx, hipEventRecord(e[3], front));
            if (run_sweep(ctx, qb.out + q0, nq, front, 1)) return 1;
            {
                StreamScope scope(ctx, ctx->stream2);
                HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream2, ctx->ev_top[0], 0));
                sa.cls_list = w.cls_list + 8 * w.batch; sa.cls_count = w.cls_count + 32; sa.row_cursor = w.cls_count + 32 + 21;
                sa.overflow_list = w.overflow_list; sa.overflow_count = w.overflow_count; sa.route_classes = 0;
                if (chain()) return 1;
                if (run_sweep_second(ctx, qb.out + q0, nq, ctx->stream2)) return 1;
                HIP_TRY(ctx, hipEventRecord(ctx->ev_top[1], ctx->stream2));
            }
            HIP_TRY(ctx, hipStreamWaitEvent(front, ctx->ev_top[1], 0));
            if (run_sweep(ctx, qb.out + q0, nq, front, 2)) return 1;
            HIP_TRY(ctx, hipEventRecord(e[4], front));
            swept_here = true;
            return 0;
        };
        if (cfused) {
            HIP_TRY(ctx, hipEventRecord(e[0], front));
            if (csd) {
                if (ctx->sd_rep_d_cap < w.batch * a.reps_pad) {
                    dev_free(ctx->sd_rep_d); ctx->sd_rep_d = nullptr; ctx->sd_rep_d_cap = 0;
                    if (dev_alloc(ctx, &ctx->sd_rep_d, w.batch * a.reps_pad)) return 1;
                    ctx->sd_rep_d_cap = w.batch * a.reps_pad;
                }
                if (launch_scoredist_reps(ctx, qb, q0, nq, ctx->sd_rep_d, w.seg_slot, w.seg_cnt)) return 1;  // (writes every segment's count)
            } else {
                HIP_TRY(ctx, hipMemsetAsync(w.seg_cnt, 0, (size_t)nq * (a.reps_pad / 64) * sizeof(int32_t), front));
                if (launch_counts_reps(ctx, qb, q0, nq, w.seg_slot, w.seg_cnt)) return 1;
            }
            HIP_TRY(ctx, hipEventRecord(e[1], front));
            ++launches;
            SelectArgs sa = select_args_alignment(ctx, qb, q0);
            sa.seg_lut = ctx->jc_lut;
            if (csd) {
                const int Lpad = (a.L + 15) / 16 * 16;
                sa.seg_lut = nullptr; sa.rep_dist = ctx->sd_rep_d;
                sa.aa_idx = a.aa_idx; sa.aa_mask = a.aa_mask; sa.q_aa = qb.aa_idx + q0 * Lpad; sa.q_aam = qb.aa_mask + q0 * (Lpad / 16);
                sa.Lpad = Lpad; sa.table = ctx->blosum;
                sa.aa_cm_idx = a.aa_cm_idx; sa.aa_cm_mask = a.aa_cm_mask; sa.cm_pad = a.slots_pad;
            }
            sa.packed_rm = a.packed_rm; sa.qpacked = csd ? nullptr : qb.packed + q0 * a.G * 3; sa.G = a.G; sa.L = a.L;
            // the member distances of accepted clusters on the matrix cores (k_cluster_dist_mfma) unless APPLES_DBG_NO_CLUSTER_MFMA
            sa.cl_mfma = (!csd && !(ctx->dbg & APPLES_DBG_NO_CLUSTER_MFMA)) ? 1 : 0;
            sa.overlap = ctx->params.overlap_frac; sa.rep_stride = a.reps_pad; sa.tmp_d = w.dist;
            sa.slow_hint = w.slow_list + w.batch;
            sa.lvl_slots = a.lvl_slots;
            {   // scratch of the cluster-major distance pass: per-cluster counters, the (query, offset) lists, the tile table
                // (beyond BIG_CAP clusters: the third form, whose offsets take 64 KB of LDS beside the bitmap -- up to the compute unit's 160 KB)
                const int64_t bitmap_bytes = ((std::max<int64_t>(a.n_e, a.n_refs) + 63) >> 6) * 10;
                const bool huge_fits = a.n_reps <= SELECT_CLUSTERS_HUGE_CAP && bitmap_bytes + (SELECT_CLUSTERS_HUGE_CAP + 1) * 4 + 6 * 1024 <= 160 * 1024 &&
                                       !knob_on(ctx, "APPLES_NO_CLUSTER_HUGE");  // (test knob: such references as before)
                const bool big_form = a.n_reps > SELECT_CLUSTERS_ACC_CAP && (a.n_reps <= SELECT_CLUSTERS_BIG_CAP || huge_fits) &&
                                      !(ctx->dbg & APPLES_DBG_NO_CLUSTER_BIG);  // (diagnostic switch: queries beyond ACC_CAP clusters to the general route)
                if (big_form && a.n_reps > SELECT_CLUSTERS_BIG_CAP && !ctx->cl_big_scr &&
                    dev_alloc(ctx, &ctx->cl_big_scr, (int64_t)SELECT_CLUSTERS_BIG_LIST * 3 * SELECT_CLUSTERS_HUGE_CAP)) return 1;
                const int64_t n_ints = 3 * (int64_t)a.n_reps + 8 + SELECT_CLUSTERS_BIG_LIST + 8 + nq + 32,  // (... + the short-form launch's list, its count in front)
                              n_items = nq * SELECT_CLUSTERS_ACC_CAP + (big_form ? std::min<int64_t>(nq, SELECT_CLUSTERS_BIG_LIST) * a.n_reps : 0),
                              n_tiles = n_items / SELECT_CLUSTERS_MIN_TILE + a.n_reps + 1;
                if (n_ints > ctx->cl_ints_cap) {
                    dev_free(ctx->cl_ints); ctx->cl_ints = nullptr; ctx->cl_ints_cap = 0;
                    if (dev_alloc(ctx, &ctx->cl_ints, n_ints)) return 1;
                    ctx->cl_ints_cap = n_ints;
                }
                if (n_items > ctx->cl_items_cap) {
                    dev_free(ctx->cl_items); ctx->cl_items = nullptr; ctx->cl_items_cap = 0;
                    if (dev_alloc(ctx, &ctx->cl_items, n_items)) return 1;
                    ctx->cl_items_cap = n_items;
                }
                if (n_tiles > ctx->cl_tiles_cap) {
                    dev_free(ctx->cl_tiles); ctx->cl_tiles = nullptr; ctx->cl_tiles_cap = 0;
                    if (dev_alloc(ctx, &ctx->cl_tiles, n_tiles)) return 1;
                    ctx->cl_tiles_cap = n_tiles;
                }
                sa.cl_count = ctx->cl_ints; sa.cl_start = ctx->cl_ints + a.n_reps; sa.cl_fill = sa.cl_start + a.n_reps + 1;
                sa.cl_ntiles = sa.cl_fill + a.n_reps;
                sa.big_count = big_form ? sa.cl_ntiles + 4 : nullptr;
                sa.big_list = big_form ? sa.cl_ntiles + 8 : nullptr;
                sa.big_scr = ctx->cl_big_scr;
                sa.gen_count = sa.cl_ntiles + 8 + SELECT_CLUSTERS_BIG_LIST + 8; sa.gen_list = sa.gen_count + 8;
                sa.cl_items = ctx->cl_items; sa.cl_tiles = ctx->cl_tiles; sa.cl_tiles_cap = ctx->cl_tiles_cap;
            }
            // clade blocks (build_blocks): the sweep inside whole subtrees of one cluster on a static schedule, cluster-major
            // (sweep_lean.hip: k_blocks_up before the selection's last phase, k_blocks_down + k_blocks_finish after the sweep above
            // them).  With the lean sweep only (MLSE / ME on a big binary tree): its kernels know a block root among the leaves.
            ctx->blk_active = a.n_blocks > 0 && ctx->params.criterion != APPLES_HYBRID && !pipelined && !ctx->tree.scan && w.small.lean && w.small.lean_leaf && w.big.lean;
            if (ctx->blk_active) {
                const int64_t n_items = nq * SELECT_CLUSTERS_ACC_CAP + std::min<int64_t>(nq, SELECT_CLUSTERS_BIG_LIST) * a.n_reps;
                const int64_t n_ints = 3 * n_items + 3 * w.batch + 16 + a.n_reps, n_tiles = n_items / 64 + a.n_reps + 1;
                if (n_ints > ctx->blk_ints_cap) {
                    dev_free(ctx->blk_ints); ctx->blk_ints = nullptr; ctx->blk_ints_cap = 0; ctx->blk_counters = nullptr;
                    if (dev_alloc(ctx, &ctx->blk_ints, n_ints)) return 1;
                    ctx->blk_ints_cap = n_ints;
                }
                if (n_tiles > ctx->blk_tiles_cap) {
                    dev_free(ctx->blk_tiles); ctx->blk_tiles = nullptr; ctx->blk_tiles_cap = 0;
                    if (dev_alloc(ctx, &ctx->blk_tiles, n_tiles)) return 1;
                    ctx->blk_tiles_cap = n_tiles;
                }
                if (!ctx->blk_pool) {
                    // the tuples of the blocks' internal nodes, 48 bytes per (query, node): room for every query of a batch observing an
                    // eighth of the reference, 16 GiB or a third of the free memory at most; a tile that finds no room goes without
                    // blocks (k_cluster_tiles).  APPLES_BLK_POOL_MB: test knob (a pool that runs dry)
                    size_t fr = 0, tot = 0;
                    int64_t bytes = std::min<int64_t>((int64_t)(w.ragged ? 40 : 16) << 30, w.batch * a.n_refs * 6);  // (ragged rows: batches three times as long)
                    if (hipMemGetInfo(&fr, &tot) == hipSuccess) bytes = std::min<int64_t>(bytes, (int64_t)(fr / 3));
                    if (knob_on(ctx, "APPLES_BLK_POOL_MB")) bytes = (int64_t)knob(ctx, "APPLES_BLK_POOL_MB", 0) << 20;
                    bytes = std::max<int64_t>(bytes, 1 << 20);
                    // (another context on the device may have taken the memory meanwhile -- two ranks on one GPU size themselves from
                    // the same hipMemGetInfo: then this context goes without blocks, it does not fail)
                    if (dev_alloc(ctx, &ctx->blk_pool, bytes / 8)) { ctx->blk_pool = nullptr; ctx->err.clear(); (void)hipGetLastError(); ctx->blk_active = false; }
                    else ctx->blk_pool_cap = bytes / 8;
                }
            }
            if (ctx->blk_active) {
                const int64_t n_items = nq * SELECT_CLUSTERS_ACC_CAP + std::min<int64_t>(nq, SELECT_CLUSTERS_BIG_LIST) * a.n_reps;
                int32_t *bi = ctx->blk_ints;
                sa.item_sbase = bi; sa.q_item = bi + n_items; sa.q_items = reinterpret_cast<int2 *>(bi + 2 * n_items);
                sa.q_blk = bi + 2 * n_items + 2 * w.batch; sa.q_item_cursor = bi + 2 * n_items + 3 * w.batch; sa.blk_ntiles = sa.q_item_cursor + 2;
                ctx->blk_counters = sa.q_item_cursor;  // (apples_describe: items and tiles of the last device batch)
                HIP_TRY(ctx, hipMemsetAsync(bi + 2 * n_items, 0, (size_t)(3 * w.batch + 16) * sizeof(int32_t), front));
                sa.blk_rec_i = a.blk_rec_i; sa.blk_rec_e = a.blk_rec_e; sa.blk_rec_c = a.blk_rec_c; sa.blk_stat = a.blk_stat[ctx->params.method == APPLES_BME ? 1 : 0]; sa.rep_soff = a.rep_soff; sa.mem_block = a.mem_block; sa.cl_order = a.cl_order;
                sa.blk_root = a.blk_root; sa.blk_rslot = a.blk_rslot; sa.blk_nodes = a.blk_nodes;
                sa.rep_boff = a.rep_boff; sa.rep_loff = a.rep_loff; sa.loose_mp = a.loose_mp;
                sa.e_of_slot = a.e_of_slot; sa.e_of_blk = a.e_of_blk; sa.e_node = a.e_node; sa.lvl_e = a.lvl_e; sa.n_e = a.n_e;
                sa.item_bad = bi + 2 * n_items + 3 * w.batch + 16; sa.cl_bbase = sa.item_bad + n_items;
                sa.blk_pool = ctx->blk_pool; sa.blk_pool_cap = ctx->blk_pool_cap;
                sa.blk_tiles = ctx->blk_tiles; sa.blk_tiles_cap = ctx->blk_tiles_cap;
                sa.method = ctx->params.method;
            }
            ctx->ev_blk_time[0] = ctx->blk_active ? e[6] : nullptr;  // (k_blocks_up's timer: launch_select_clusters records the two events
            ctx->ev_blk_time[1] = ctx->blk_active ? e[7] : nullptr;  // where it launches the kernel and clears the first)
            if (launch_select_clusters(ctx, sa, nq)) return 1;
            blk_timed[(size_t)i] = ctx->blk_active && ctx->ev_blk_time[0] == nullptr;
            // queries whose accepted clusters hold fewer than -b valid distances: the top-up rule over the representatives
            // (phase 4 of k_select_clusters); what that cannot hold: full rows + general selection
            const bool no_listed = (ctx->dbg & APPLES_DBG_NO_CLUSTER_TOPUP) != 0;  // diagnostic switch: everything through the general route
            int32_t *fwd_list = w.slow_list + 2 * w.batch, *fwd_count = w.cls_count + 20;
            if (!no_listed) {
                sa.rep_panel = a.rep_packed; sa.slow2_list = fwd_list; sa.slow2_count = fwd_count;
                sa.qlist = w.slow_list; sa.qcount = w.slow_count; sa.qhint = w.slow_list + w.batch;
                if (launch_select_clusters_listed(ctx, sa, nq)) return 1;
            }
            sa.seg_lut = nullptr;
            sa.dist = w.dist_slow;
            if (for_slow_slices(nq, [&](const int32_t *lst, const int32_t *cntp, int64_t n_max) -> int {
                    if (csd ? launch_scoredist_listed(ctx, qb, q0, n_max, lst, cntp, w.dist_slow)
                            : launch_counts_listed(ctx, qb, q0, n_max, lst, cntp, w.dist_slow, nullptr, nullptr)) return 1;
                    sa.qlist = lst;
                    sa.qcount = cntp;
                    sa.qhint = no_listed ? lst + w.batch : nullptr;
                    return launch_select(ctx, sa, n_max);
                }, no_listed ? nullptr : fwd_list, no_listed ? nullptr : fwd_count)) return 1;
            HIP_TRY(ctx, hipEventRecord(e[2], front));
        } else if (sfused) {
            HIP_TRY(ctx, hipEventRecord(e[0], front));
            if (qb.sd_q4 && sd_gemm_usable(ctx)) {
                // lower bounds on the matrix cores -> candidates per segment -> exact distances of the candidates, the
                // segments closed up in place (dist_sd.hip)
                HIP_TRY(ctx, hipMemsetAsync(w.seg_cnt, 0, (size_t)nq * (w.stride / 64) * sizeof(int32_t), front));
                if (launch_sd_filter(ctx, qb, q0, nh, w.seg_slot, w.seg_cnt)) return 1;
                if (nh < nq) {  // the rest of the chunk travels meanwhile
                    if (feed_rest(i)) return 1;
                    HIP_TRY(ctx, hipStreamWaitEvent(front, ctx->ev_feed[2 * i + 1], 0));
                    if (launch_sd_filter(ctx, qb, q0 + nh, nq - nh, w.seg_slot + nh * w.stride, w.seg_cnt + nh * (w.stride / 64))) return 1;
                    ++launches;
                }
                HIP_TRY(ctx, hipEventRecord(e[5], front));
                sd_filter_timed[(size_t)i] = 1;
                if (launch_sd_exact(ctx, qb, q0, nq, w.dist, w.seg_slot, w.seg_cnt, w.n_obs)) return 1;
            } else if (launch_scoredist_fused(ctx, qb, q0, nq, w.dist, w.seg_slot, w.seg_cnt, nullptr)) return 1;
            HIP_TRY(ctx, hipEventRecord(e[1], front));
            ++launches;
            SelectArgs sa = select_args_alignment(ctx, qb, q0);
            sa.seg_lut = nullptr;
            if (qb.sd_q4 && sd_gemm_usable(ctx)) sa.seg_surv = w.n_obs;  // (k_sd_exact left the survivor counts where k_select_fast puts the observed counts)
            if (launch_select_fast(ctx, sa, nq)) return 1;
            // the queries that need the top-up rule: full rows for the listed queries only (row r of dist_slow = list
            // entry r), a slice of the list at a time, then the general selection over those rows
            sa.dist = w.dist_slow;
            const bool no_sd_topup = (ctx->dbg & APPLES_DBG_NO_SD_TOPUP) != 0;  // diagnostic switch: full rows for the listed queries
            const bool lb_topup = qb.sd_q4 && sd_gemm_usable(ctx) && !no_sd_topup && w.dist_rows >= nq;
            if (lb_topup && ctx->sd_list_rows < w.batch) {
                dev_free(ctx->sd_list_img); ctx->sd_list_img = nullptr; ctx->sd_list_rows = 0;
                dev_free(ctx->sd_list_ints); ctx->sd_list_ints = nullptr;
                if (dev_alloc(ctx, &ctx->sd_list_img, sd_query_image_bytes(ctx, round_up(w.batch, 256)))) return 1;
                if (dev_alloc(ctx, &ctx->sd_list_ints, 2 * w.batch + 1)) return 1;
                ctx->sd_list_rows = w.batch;
            }
            // compact lists of the evaluated references for the selection (dist_sd.hip:k_sd_topup<true>) instead of rows of
            // n_slots values; what a compact row cannot hold goes through the row form afterwards
            const bool compact = lb_topup && !(ctx->dbg & APPLES_DBG_NO_SD_COMPACT);
            int32_t *c_len = compact ? ctx->sd_list_ints : nullptr, *c_count2 = compact ? ctx->sd_list_ints + w.batch : nullptr,
                    *c_list2 = compact ? ctx->sd_list_ints + w.batch + 1 : nullptr;
            if (overlapped(sa, [&]() -> int {
                    return for_slow_slices(nq, [&](const int32_t *lst, const int32_t *cntp, int64_t n_max) -> int {
                        // rows of lower bounds on the matrix cores (into the listed queries' rows of w.dist: k_select_fast is done
                        // with them), exact distances only where the `-b` nearest can be (dist_sd.hip:k_sd_topup); else full rows
                        if (lb_topup ? launch_sd_topup(ctx, qb, q0, n_max, lst, cntp, ctx->sd_list_img, w.dist, w.dist_slow, c_len, c_list2, c_count2)
                                     : launch_scoredist_listed(ctx, qb, q0, n_max, lst, cntp, w.dist_slow)) return 1;
                        sa.qlist = lst;
                        sa.qcount = cntp;
                        if (!compact) return launch_select(ctx, sa, n_max);
                        sa.row_len = c_len; sa.row_cap = sd_compact_cap(ctx);
                        if (launch_select(ctx, sa, n_max)) return 1;
                        // ... and the queries whose lists did not fit (usually none: the launches find an empty list)
                        sa.row_len = nullptr; sa.row_cap = 0;
                        if (launch_sd_topup_rows_again(ctx, qb, q0, n_max, c_list2, c_count2, w.dist, w.dist_slow)) return 1;
                        sa.qlist = c_list2;
                        sa.qcount = c_count2;
                        return launch_select(ctx, sa, n_max);
                    });
                })) return 1;
        } else if (fused) {
            HIP_TRY(ctx, hipEventRecord(e[0], front));
            if (fused_counts_format(ctx, qb))  // the matrix-core kernel writes only the non-empty segments' counts
                HIP_TRY(ctx, hipMemsetAsync(w.seg_cnt, 0, (size_t)nq * (w.stride / 64) * sizeof(int32_t), front));
            if (launch_counts_fused(ctx, qb, q0, nh, dist_tile_for(ctx, nq), w.dist, w.seg_slot, w.seg_cnt)) return 1;
            if (nh < nq) {  // the rest of the chunk travels meanwhile (split_feed: the GEMM form, packed survivors in seg_slot alone)
                if (feed_rest(i)) return 1;
                HIP_TRY(ctx, hipStreamWaitEvent(front, ctx->ev_feed[2 * i + 1], 0));
                if (launch_counts_fused(ctx, qb, q0 + nh, nq - nh, dist_tile_for(ctx, nq), w.dist, w.seg_slot + nh * w.stride,
                                        w.seg_cnt + nh * (w.stride / 64))) return 1;
                ++launches;
            }
            // reference rows with bytes beyond ACGT- (the matrix-core pass took them for gaps): the survivors' exact counts
            // reference rows with bytes beyond ACGT- (the matrix-core pass took them for gaps): k_select_fast counts those sites for the
            // survivors on such rows and tests them once more (APPLES_EXOTIC_FIX_KERNEL: the same as a pass of its own, diagnostic)
            const bool exfix = fused_counts_format(ctx, qb) && (a.ex_off != nullptr || ctx->jc_mmax_true != nullptr);
            const bool ex_kernel = knob_on(ctx, "APPLES_EXOTIC_FIX_KERNEL");
            if (exfix && ex_kernel && launch_exotic_fix(ctx, qb, q0, nq, w.seg_slot, w.seg_cnt, w.n_obs)) return 1;
            HIP_TRY(ctx, hipEventRecord(e[1], front));
            ++launches;
            SelectArgs sa = select_args_alignment(ctx, qb, q0);
            sa.seg_lut = fused_counts_format(ctx, qb) ? ctx->jc_lut : nullptr;
            if (exfix && ex_kernel) sa.seg_surv = w.n_obs;  // (k_exotic_fix left the survivor counts where k_select_fast puts the observed counts)
            if (exfix && !ex_kernel) {
                sa.ex_off = a.ex_off; sa.ex_site = a.ex_site; sa.ex_mmax = ctx->jc_mmax_true ? ctx->jc_mmax_true : ctx->jc_mmax;
                sa.ex_all = ctx->jc_mmax_true ? 1 : 0; sa.q_raw = qb.raw + q0 * (int64_t)a.L; sa.L = a.L;
            }
            if (launch_select_fast(ctx, sa, nq)) return 1;
            sa.seg_surv = nullptr; sa.ex_mmax = nullptr;
            sa.seg_lut = nullptr;
            // top-up path for the queries k_select_fast listed: full rows + per-segment minima (in the
            // rows of the fused buffers, which k_select_fast has consumed), then the `-b` nearest
            const bool no_topup = (ctx->dbg & APPLES_DBG_NO_TOPUP_KERNEL) != 0;  // diagnostic switch
            // (short rows are cheaper to stream twice than to rank: 2 x `-b` block arg-min rounds)
            const int64_t topup_min = (int64_t)knob(ctx, "APPLES_TOPUP_MIN_ROWS", 40000);
            const bool topup = !no_topup && ctx->params.base_observation <= 256 && ctx->aln.n_refs >= topup_min;
            sa.dist = w.dist_slow;
            sa.segmin_d = w.dist;
            sa.segmin_i = w.seg_slot;
            if (overlapped(sa, [&]() -> int {
                    return for_slow_slices(nq, [&](const int32_t *lst, const int32_t *cntp, int64_t n_max) -> int {
                        if (launch_counts_listed(ctx, qb, q0, n_max, lst, cntp, w.dist_slow, topup ? w.dist : nullptr,
                                                 topup ? w.seg_slot : nullptr)) return 1;
                        sa.qlist = lst;
                        sa.qcount = cntp;
                        return topup ? launch_select_topup(ctx, sa, n_max) : launch_select(ctx, sa, n_max);
                    });
                })) return 1;
        } else {
            HIP_TRY(ctx, hipEventRecord(e[0], front));
            if (ctx->params.model == APPLES_SCOREDIST) {
                if (launch_scoredist(ctx, qb, q0, nq, w.dist, nullptr)) return 1;
            } else {
                if (launch_counts(ctx, qb, q0, nq, dist_tile_for(ctx, nq), w.dist, nullptr)) return 1;
            }
            HIP_TRY(ctx, hipEventRecord(e[1], front));
            ++launches;
            if (launch_select(ctx, select_args_alignment(ctx, qb, q0), nq)) return 1;
            HIP_TRY(ctx, hipEventRecord(e[2], front));
        }
        if (pipelined) {
            HIP_TRY(ctx, hipEventRecord(ctx->ev_front[set], front));
            HIP_TRY(ctx, hipStreamWaitEvent(back, ctx->ev_front[set], 0));
        }
        if (feed && feed_rest(i)) return 1;  // (a route that did not ask for the second piece itself: nothing left behind)
        if (!swept_here) {
            HIP_TRY(ctx, hipEventRecord(e[3], back));
            if (cfused && ctx->blk_active) HIP_TRY(ctx, hipStreamWaitEvent(back, ctx->ev_blk[1], 0));  // the blocks' tuples (k_blocks_up on stream_big)
            if (run_sweep(ctx, qb.out + q0, nq, back)) return 1;
            if (cfused && ctx->blk_active) {  // the top-down pass inside the clade blocks, then the better of the two placements
                const int64_t n_items = nq * SELECT_CLUSTERS_ACC_CAP + std::min<int64_t>(nq, SELECT_CLUSTERS_BIG_LIST) * a.n_reps;
                int32_t *bi = ctx->blk_ints;
                BlockArgs b{};
                b.tiles = ctx->blk_tiles; b.n_tiles = bi + 2 * n_items + 3 * w.batch + 2; b.items = ctx->cl_items;
                b.rec_i = a.blk_rec_i; b.rec_e = a.blk_rec_e; b.rec_c = a.blk_rec_c; b.rec_p = a.blk_rec_p; b.pk_i = a.blk_pk_i; b.pk_e = a.blk_pk_e; b.stat = a.blk_stat[ctx->params.method == APPLES_BME ? 1 : 0]; b.rep_soff = a.rep_soff; b.rep_moff = a.rep_moff; b.slot_rep = a.slot_rep; b.slot_mpos = a.slot_mpos;
                b.self_slot = qb.self_slot + q0; b.tmp_d = w.dist; b.stride = a.slots_pad; b.row_off = w.ragged ? w.row_off : nullptr; b.pool = ctx->blk_pool;
                b.item_sbase = bi; b.item_bad = bi + 2 * n_items + 3 * w.batch + 16;
                b.q_item = bi + n_items; b.q_items = reinterpret_cast<const int2 *>(bi + 2 * n_items);
                b.q_blk = bi + 2 * n_items + 2 * w.batch; b.cursor = bi + 2 * n_items + 3 * w.batch + 1;
                b.method = ctx->params.method; b.criterion = ctx->params.criterion; b.negative = ctx->params.negative_branch;
                b.out = qb.out + q0; b.nq = nq;
                if (launch_blocks_down(ctx, b, back)) return 1;
            }
            HIP_TRY(ctx, hipEventRecord(e[4], back));
        }
        ctx->blk_active = false;
        if (pipelined) HIP_TRY(ctx, hipEventRecord(ctx->ev_back[set], back));
        if (feed && i + 1 < n_sub && feed_chunk(i + 1)) return 1;  // the next chunk travels while this sub-batch's kernels run
    }
    if (pipelined) {
        HIP_TRY(ctx, hipEventRecord(e_stop, back));
        HIP_TRY(ctx, hipStreamWaitEvent(front, e_stop, 0));
        if ((n_sub & 1) == 0) swap_bufs(w);  // leave the host view on set 0
    }
    int32_t rag[2] = {0, 0};
    if (w.ragged) HIP_TRY(ctx, hipMemcpyAsync(rag, ctx->d_rag, sizeof(rag), hipMemcpyDeviceToHost, front));
    HIP_TRY(ctx, hipEventRecord(e_stop, front));
    HIP_TRY(ctx, hipEventSynchronize(e_stop));
    if (w.ragged) {
        HIP_TRY(ctx, hipStreamSynchronize(front));  // (rag is on the stack)
        if (rag[0]) {
            // a device batch ran out of big rows (common.h, Workspace::ragged): its queries shared the spare row and their placements
            // are not to be used.  The block once more with full rows, and full rows for this context from here on (the block is on
            // the device by now: no feed)
            ctx->no_ragged = true;
            return run_block_main(ctx, qb, nullptr);
        }
    }
    for (int i = 0; i < APPLES_T_COUNT; ++i) ctx->t_ms[i] = 0;
    for (int64_t i = 0; i < n_sub; ++i) {
        hipEvent_t *e = &ev[(size_t)i * 8];
        float ms = 0;
        if (blk_timed[(size_t)i]) {
            float bms = 0;
            (void)hipEventElapsedTime(&bms, e[6], e[7]);
            ctx->t_ms[APPLES_T_BLOCKS] += bms;
        }
        (void)hipEventElapsedTime(&ms, e[0], e[1]); ctx->t_ms[APPLES_T_DIST] += ms;
        if (sd_filter_timed[(size_t)i]) {  // (a sub-batch without the filter adds nothing: T_FILTER = 0 means no filter ran)
            float fms = 0;
            (void)hipEventElapsedTime(&fms, e[0], e[5]);
            ctx->t_ms[APPLES_T_FILTER] += fms;
        }
        (void)hipEventElapsedTime(&ms, e[1], e[2]); ctx->t_ms[APPLES_T_SELECT] += ms;
        (void)hipEventElapsedTime(&ms, e[3], e[4]); ctx->t_ms[APPLES_T_SWEEP] += ms;
    }
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e_start, e_stop);
    ctx->t_ms[APPLES_T_TOTAL] = ms;
    ctx->t_ms[APPLES_T_DIST_LAUNCHES] = launches;
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}

}  // namespace

extern "C" {

const char *apples_last_error(const apples_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

uint32_t apples_abi_version(void) { return APPLES_ABI_VERSION; }
size_t apples_params_size(void) { return sizeof(apples_params); }

int apples_ctx_create(const apples_tree *tree, const apples_alignment *aln, const apples_params *params, int device,
                      apples_ctx **out) {
    *out = nullptr;
    apples_ctx *ctx = new apples_ctx();
    auto fail = [&]() {
        g_create_error = ctx->err;
        apples_ctx_destroy(ctx);
        return 1;
    };
    ctx->device = device;
    ctx->params = *params;
    ctx->dbg = params->debug & APPLES_DBG_ALL;  // (bits this build does not know are ignored: a caller built against a newer header)
    {
        static const struct { const char *env; uint32_t bit; } knobs[] = {
            {"APPLES_NO_FUSE", APPLES_DBG_NO_FUSE}, {"APPLES_SWEEP_SCAN", APPLES_DBG_SWEEP_SCAN}, {"APPLES_NODE_MAP", APPLES_DBG_NODE_MAP},
            {"APPLES_SWEEP_MERGE", APPLES_DBG_SWEEP_MERGE}, {"APPLES_NO_SWEEP_MERGE", APPLES_DBG_NO_SWEEP_MERGE},
            {"APPLES_NO_DIST_GEMM", APPLES_DBG_NO_DIST_GEMM}, {"APPLES_NO_SWEEP_LEAN", APPLES_DBG_NO_SWEEP_LEAN},
            {"APPLES_NO_SD_GEMM", APPLES_DBG_NO_SD_GEMM}, {"APPLES_CLUSTER_BY_QUERY", APPLES_DBG_CLUSTER_BY_QUERY},
            {"APPLES_NO_CLUSTER_TOPUP", APPLES_DBG_NO_CLUSTER_TOPUP}, {"APPLES_NO_STREAM_SELECT", APPLES_DBG_NO_STREAM_SELECT},
            {"APPLES_NO_TOPUP_KERNEL", APPLES_DBG_NO_TOPUP_KERNEL}, {"APPLES_NO_CLUSTER_BIG", APPLES_DBG_NO_CLUSTER_BIG},
            {"APPLES_NO_SD_TOPUP", APPLES_DBG_NO_SD_TOPUP}, {"APPLES_SD_FP6", APPLES_DBG_SD_FP6},
            {"APPLES_NO_TOPUP_OVERLAP", APPLES_DBG_NO_TOPUP_OVERLAP}, {"APPLES_STREAM_THIRD_PASS", APPLES_DBG_STREAM_THIRD_PASS},
            {"APPLES_NO_SD_COMPACT", APPLES_DBG_NO_SD_COMPACT}, {"APPLES_SD_COMPACT_TINY", APPLES_DBG_SD_COMPACT_TINY},
            {"APPLES_NO_BLOCKS", APPLES_DBG_NO_BLOCKS}, {"APPLES_HYBRID_RECORDS", APPLES_DBG_HYBRID_RECORDS},
            {"APPLES_NO_CLUSTER_MFMA", APPLES_DBG_NO_CLUSTER_MFMA}};
        for (const auto &k : knobs)
            if (getenv(k.env)) ctx->dbg |= k.bit;
    }
    {   // the tuning / experiment / test knobs of this context: the process environment's APPLES_* variables, then apples_params.knobs
        // over them ("NAME=value;NAME=value", the APPLES_ prefix optional, a bare NAME = 1).  Read here and nowhere else.
        extern char **environ;
        auto put = [&](const std::string &kv) {
            if (kv.empty()) return;
            const size_t eq = kv.find('=');
            std::string name = kv.substr(0, eq);
            if (name.rfind("APPLES_", 0) != 0) name = "APPLES_" + name;
            const std::string val = eq == std::string::npos ? "1" : kv.substr(eq + 1);
            ctx->knobs[name] = val.empty() ? 1 : atoll(val.c_str());
        };
        for (char **e = environ; e && *e; ++e)
            if (strncmp(*e, "APPLES_", 7) == 0) put(*e);
        if (params->knobs) {
            std::string all(params->knobs);
            size_t at = 0;
            while (at <= all.size()) {
                const size_t semi = all.find(';', at);
                put(all.substr(at, semi == std::string::npos ? std::string::npos : semi - at));
                if (semi == std::string::npos) break;
                at = semi + 1;
            }
        }
        ctx->params.knobs = nullptr;  // (the caller's string is not kept)
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        ctx->err = "no HIP device available: the APPLES hot path needs an MI355X (there is no CPU fallback)";
        return fail();
    }
    if (hipSetDevice(device) != hipSuccess) { ctx->err = "hipSetDevice failed"; return fail(); }
    // Three streams and nothing on the null stream: the runtime spreads a process's streams over four hardware queues
    // (GPU_MAX_HW_QUEUES) and two streams on one queue run their kernels one after the other (scripts/stream_queue_probe.hip);
    // the sweep's launches side by side need queues of their own.  stream3 (the pipeline experiments' back stream) is made
    // when an experiment asks for it (back_stream).
    if (hipStreamCreate(&ctx->stream) != hipSuccess || hipStreamCreate(&ctx->stream2) != hipSuccess ||
        hipStreamCreate(&ctx->stream_big) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_front[0], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_front[1], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_back[0], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_back[1], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_bigfree, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_cl[0], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_cl[1], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_half[0], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_half[1], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_blk[0], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_blk[1], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_sel, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_top[0], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_top[1], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_big, hipEventDisableTiming) != hipSuccess) { ctx->err = "hipStreamCreate failed"; return fail(); }
    for (int i = 0; i < 8; ++i)
        if (hipEventCreate(&ctx->ev[i]) != hipSuccess) { ctx->err = "hipEventCreate failed"; return fail(); }
    if (tree->n_nodes < 2) { ctx->err = "tree needs at least two nodes"; return fail(); }
    if (dev_alloc(ctx, &ctx->d_exotic, 1) || hipMemsetAsync(ctx->d_exotic, 0, sizeof(int), ctx->stream) != hipSuccess) return fail();
    if (knob_on(ctx, "APPLES_SCAN_PROFILE"))
        if (dev_alloc(ctx, &ctx->scan_prof, 8) || hipMemsetAsync(ctx->scan_prof, 0, 64, ctx->stream) != hipSuccess) return fail();
    if (knob_on(ctx, "APPLES_LEAN_PROFILE"))
        if (dev_alloc(ctx, &ctx->lean_prof, 16) || hipMemsetAsync(ctx->lean_prof, 0, 128, ctx->stream) != hipSuccess) return fail();
    if (upload_tree(ctx, tree)) return fail();
    if (aln) {
        if (setup_alignment(ctx, tree, aln)) return fail();
        ctx->has_aln = true;
    }
    if (apples_set_params(ctx, params)) return fail();
    if (hipStreamSynchronize(ctx->stream) != hipSuccess) { ctx->err = "context upload failed"; return fail(); }
    *out = ctx;
    return 0;
}

// mmax[valid] as a linear rule in float arithmetic for dist_gemm.hip's epilogue: from the first valid count that
// admits anything, mmax[valid] = floor(p_f valid) with p_f = 3/4 (1 - exp(-4 f / 3)); the kernel evaluates
// t = fmaf(8192 valid, slope, off) and keeps 4 mism + 2049 <= t (its accumulators start at -2049, the first addend of
// the decode, so everything it compares carries that shift and `off` includes it).  A candidate (slope, off) is
// accepted only if it reproduces the table for EVERY valid count: 4 mmax + 2049 <= t < 4 mmax + 2049 + 4.
static GemmThreshold gemm_threshold(const std::vector<int32_t> &mmax, double f) {
    GemmThreshold g;
    const int L = (int)mmax.size() - 1;
    if (L > GEMM_MAX_L13) return g;  // (longer alignments: the 2^12 form tests through the table)
    int vmin = 0;
    while (vmin <= L && mmax[vmin] < 0) ++vmin;
    for (int v = vmin; v <= L; ++v)
        if (mmax[v] < 0) return g;
    const double pf = 0.75 * (1.0 - std::exp(-4.0 * f / 3.0));
    const float s0 = (float)(4.0 * pf / 8192.0);
    const float offs[] = {0.f, 0.0005f, -0.0005f, 0.001f, -0.001f, 0.002f, -0.002f, 0.004f, -0.004f};
    for (int ds = 0; ds < 5; ++ds) {
        float slope = s0;
        for (int k = 0; k < (ds + 1) / 2; ++k) slope = std::nextafterf(slope, (ds & 1) ? 1.f : -1.f);
        for (float off : offs) {
            bool good = true;
            for (int v = vmin; v <= L && good; ++v) {
                const float t = std::fmaf(8192.f * (float)v, slope, off + 2049.f);
                good = 4.f * (float)mmax[v] + 2049.f <= t && t < 4.f * (float)mmax[v] + 2049.f + 4.f;
            }
            if (good) {
                g.slope = slope; g.off = off + 2049.f; g.vmin8 = 8192.f * (float)vmin; g.ok = true;
                return g;
            }
        }
    }
    return g;
}

int apples_set_params(apples_ctx *ctx, const apples_params *params) {
    HIP_TRY(ctx, hipSetDevice(ctx->device));  // the caller's thread may have another device current
    int model = ctx->params.model;
    if (ctx->has_aln && params->model != model) { ctx->err = "the distance model is fixed at context creation"; return 1; }
    ctx->params = *params;
    ctx->params.debug = ctx->dbg;  // (the switches are fixed at creation)
    ctx->params.knobs = nullptr;   // (so are the knobs: read once, apples_ctx_create)
    ctx->params.jc_lut = nullptr;
    if (params->jc_lut && params->jc_lut_len > 0) {
        int64_t L = ctx->has_aln ? ctx->aln.L : 0;
        int64_t need = (L + 1) * (L + 2) / 2;
        if (ctx->has_aln && params->jc_lut_len < need) { ctx->err = "jc_lut too short for this alignment length"; return 1; }
        dev_free(ctx->jc_lut);
        ctx->jc_lut = nullptr;
        if (dev_upload(ctx, &ctx->jc_lut, params->jc_lut, params->jc_lut_len)) return 1;
        ctx->jc_lut_len = params->jc_lut_len;
        dev_free(ctx->jc_mmax);
        ctx->jc_mmax = nullptr;
        ctx->gemm_thr = GemmThreshold();
        if (ctx->has_aln) {  // integer form of 0 <= d <= threshold, valid only if the table is monotone in mism
            std::vector<int32_t> mmax(L + 1, -1);
            bool monotone = true;
            for (int64_t v = 0; v <= L; ++v) {
                const double *row = params->jc_lut + v * (v + 1) / 2;
                int last = -1;
                for (int64_t m = 0; m <= v; ++m)
                    if (row[m] >= 0 && row[m] <= params->filt_threshold) last = (int)m;
                for (int64_t m = 0; m <= last; ++m)
                    if (!(row[m] >= 0 && row[m] <= params->filt_threshold)) monotone = false;
                mmax[v] = last;
            }
            dev_free(ctx->jc_mmax_true);
            ctx->jc_mmax_true = nullptr;
            if (monotone && ctx->aln.ex_max > 0) {
                // reference rows with bytes beyond ACGT- (counted as gaps by the matrix-core pass): a pair's true counts are (valid + k,
                // mism + k), 0 <= k <= ex_max, so the pass must keep (valid, mism) whenever SOME k passes: mmaxF[v] = max_k (mmax[v + k] - k).
                // It differs from the rule only at the low end, where -V could keep the smaller count out; k_exotic_fix then tests
                // every survivor against the rule itself.
                std::vector<int32_t> mf(mmax);
                bool loosened = false;
                for (int64_t v = 0; v <= L; ++v)
                    for (int64_t k = 1; k <= ctx->aln.ex_max && v + k <= L; ++k)
                        if (mmax[v + k] >= 0 && mmax[v + k] - (int32_t)k > mf[v]) { mf[v] = mmax[v + k] - (int32_t)k; loosened = true; }
                if (loosened) {
                    if (dev_upload(ctx, &ctx->jc_mmax_true, mmax.data(), L + 1)) return 1;
                    mmax = mf;
                }
            }
            if (monotone && dev_upload(ctx, &ctx->jc_mmax, mmax.data(), L + 1)) return 1;
            ctx->gemm_thr = monotone ? gemm_threshold(mmax, params->filt_threshold) : GemmThreshold();
        }
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    } else {
        dev_free(ctx->jc_lut);
        ctx->jc_lut = nullptr;
        ctx->jc_lut_len = 0;
        dev_free(ctx->jc_mmax);
        ctx->jc_mmax = nullptr;
        ctx->gemm_thr = GemmThreshold();
    }
    return 0;
}

void apples_ctx_destroy(apples_ctx *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    if (ctx->stream2) (void)hipStreamSynchronize(ctx->stream2);
    if (ctx->stream3) (void)hipStreamSynchronize(ctx->stream3);
    if (ctx->stream_big) (void)hipStreamSynchronize(ctx->stream_big);
    if (ctx->scan_prof) {  // diagnostic: where the scan sweep's teams spent their cycles
        unsigned long long h[8] = {};
        (void)hipMemcpy(h, ctx->scan_prof, sizeof h, hipMemcpyDeviceToHost);
        double tot = 0;
        for (int i = 0; i < 6; ++i) tot += (double)h[i];
        fprintf(stderr, "scan sweep phases (share of team cycles): queue %.3f  phase0 %.3f  phase1 %.3f  bottom-up %.3f  top-down %.3f  select %.3f  (queries %llu)\n",
                h[0] / tot, h[1] / tot, h[2] / tot, h[3] / tot, h[4] / tot, h[5] / tot, h[6]);
        dev_free(ctx->scan_prof);
    }
    if (ctx->lean_prof) {  // diagnostic: where the lean sweep's wavefront-sized teams spent their cycles
        unsigned long long h[16] = {};
        (void)hipMemcpy(h, ctx->lean_prof, sizeof h, hipMemcpyDeviceToHost);
        double tot = 0;
        for (int i = 0; i < 7; ++i) tot += (double)h[i];
        if (tot > 0)
            fprintf(stderr, "lean sweep phases (share of team cycles): queue %.3f  up-front %.3f  bottom-up on chip %.3f (%llu steps, %.0f cycles each)  "
                            "bottom-up general %.3f (%llu steps, %.0f)  top-down pairs %.3f (%llu steps, %.0f)  top-down general %.3f (%llu steps, %.0f)  "
                            "select %.3f  (queries %llu, %.0f cycles each; %llu handed to the workgroup-sized teams for want of room)\n",
                    h[0] / tot, h[1] / tot, h[2] / tot, h[8], h[8] ? (double)h[2] / h[8] : 0.0, h[3] / tot, h[9], h[9] ? (double)h[3] / h[9] : 0.0,
                    h[4] / tot, h[10], h[10] ? (double)h[4] / h[10] : 0.0, h[5] / tot, h[11], h[11] ? (double)h[5] / h[11] : 0.0, h[6] / tot,
                    h[12], h[12] ? tot / h[12] : 0.0, h[13]);
        dev_free(ctx->lean_prof);
    }
    for (auto &qb : ctx->blocks) free_block(ctx, &qb);
    for (auto &c : ctx->blk_cache) dev_free(c.second);
    ctx->blk_cache.clear();
    dev_free(ctx->d_exotic);
    dev_free(ctx->d_slice_cnt);
    dev_free(ctx->cl_ints); dev_free(ctx->cl_items); dev_free(ctx->cl_tiles); dev_free(ctx->cl_big_scr); dev_free(ctx->d_rag); dev_free(ctx->sd_rep_d); dev_free(ctx->blk_pool); dev_free(ctx->blk_ints); dev_free(ctx->blk_tiles);
    for (auto &e : ctx->ev_feed) (void)hipEventDestroy(e);
    free_workspace(ctx->ws);
    DevTree &t = ctx->tree;
    dev_free(t.parent); dev_free(t.edge_len); dev_free(t.child_off); dev_free(t.child_idx); dev_free(t.level); dev_free(t.rec); dev_free(t.lvlw); dev_free(t.lnode); dev_free(t.rec_l); dev_free(t.npos); dev_free(t.pe); dev_free(t.leaf_info); dev_free(t.anc); dev_free(t.rmq);
    DevAlign &a = ctx->aln;
    dev_free(a.raw); dev_free(a.d_slot_row); dev_free(a.packed); dev_free(a.ref_f4); dev_free(a.rep_boff); dev_free(a.rep_loff); dev_free(a.loose_mp); dev_free(a.blk_rec_i); dev_free(a.blk_rec_e); dev_free(a.blk_rec_c); dev_free(a.blk_rec_p); dev_free(a.blk_pk_i); dev_free(a.blk_pk_e); dev_free(a.blk_stat[0]); dev_free(a.blk_stat[1]); dev_free(a.rep_soff); dev_free(a.cl_order); dev_free(a.mem_block); dev_free(a.blk_root); dev_free(a.blk_rslot); dev_free(a.blk_nodes); dev_free(a.e_of_slot); dev_free(a.e_of_blk); dev_free(a.e_node); dev_free(a.lvl_e); dev_free(a.rep_packed); dev_free(a.packed_rm); dev_free(a.aa_rep_idx); dev_free(a.aa_rep_mask); dev_free(a.aa_cm_idx); dev_free(a.aa_cm_mask); dev_free(a.aa_idx); dev_free(a.aa_mask); dev_free(a.sd_ref4); dev_free(a.sd_nvr); dev_free(a.aa_rows); dev_free(a.aa_mrows); dev_free(ctx->sd_tq4); dev_free(ctx->sd_list_img); dev_free(ctx->sd_list_ints); dev_free(a.slot_node); dev_free(a.slot_level); dev_free(a.lvl_slots);
    dev_free(a.slot_rep); dev_free(a.slot_mpos); dev_free(a.rep_slot); dev_free(a.rep_moff); dev_free(a.mem_slot);
    dev_free(ctx->jc_lut); dev_free(ctx->jc_mmax); dev_free(ctx->jc_mmax_true); dev_free(ctx->blosum); dev_free(ctx->d_col_perm); dev_free(ctx->d_col_node);
    dev_free(ctx->d_col_level);
    for (int i = 0; i < 8; ++i)
        if (ctx->ev[i]) (void)hipEventDestroy(ctx->ev[i]);
    for (auto &e : ctx->ev_pool) (void)hipEventDestroy(e);
    ctx->ev_pool.clear();
    for (int i = 0; i < 2; ++i) {
        if (ctx->ev_front[i]) (void)hipEventDestroy(ctx->ev_front[i]);
        if (ctx->ev_back[i]) (void)hipEventDestroy(ctx->ev_back[i]);
    }
    if (ctx->ev_bigfree) (void)hipEventDestroy(ctx->ev_bigfree);
    for (auto &e : ctx->ev_cl) if (e) (void)hipEventDestroy(e);
    for (auto &e : ctx->ev_blk) if (e) (void)hipEventDestroy(e);
    for (auto &e : ctx->ev_half) if (e) (void)hipEventDestroy(e);
    if (ctx->stream3) (void)hipStreamDestroy(ctx->stream3);
    if (ctx->stream_big) (void)hipStreamDestroy(ctx->stream_big);
    if (ctx->ev_sel) (void)hipEventDestroy(ctx->ev_sel);
    if (ctx->ev_big) (void)hipEventDestroy(ctx->ev_big);
    for (int i = 0; i < 2; ++i)
        if (ctx->ev_top[i]) (void)hipEventDestroy(ctx->ev_top[i]);
    if (ctx->stream2) (void)hipStreamDestroy(ctx->stream2);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

static int new_block_slot(apples_ctx *ctx) {
    size_t slot = ctx->blocks.size();
    for (size_t i = 0; i < ctx->blocks.size(); ++i)
        if (!ctx->blocks[i].live) { slot = i; break; }
    if (slot == ctx->blocks.size()) ctx->blocks.emplace_back();
    return (int)slot;
}

int apples_queries_upload(apples_ctx *ctx, const uint8_t *queries, int64_t n_queries, const int32_t *self_row,
                          int64_t *handle) {
    if (!ctx->has_aln) { ctx->err = "context has no alignment"; return 1; }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    QueryBlock qb;
    if (make_block(ctx, queries, n_queries, self_row, &qb)) { free_block(ctx, &qb); return 1; }
    const int slot = new_block_slot(ctx);
    ctx->blocks[slot] = qb;
    *handle = (int64_t)slot;
    return 0;
}

int apples_queries_free(apples_ctx *ctx, int64_t handle) {
    if (handle < 0 || handle >= (int64_t)ctx->blocks.size() || !ctx->blocks[handle].live) { ctx->err = "bad handle"; return 1; }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    free_block(ctx, &ctx->blocks[handle]);
    return 0;
}

static int run_table_block(apples_ctx *ctx, QueryBlock &qb);

int apples_place_resident(apples_ctx *ctx, int64_t handle) {
    if (handle < 0 || handle >= (int64_t)ctx->blocks.size() || !ctx->blocks[handle].live) { ctx->err = "bad handle"; return 1; }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (ctx->blocks[handle].table) return run_table_block(ctx, ctx->blocks[handle]);
    return run_block(ctx, ctx->blocks[handle]);
}

int apples_fetch_placements(apples_ctx *ctx, int64_t handle, apples_placement *out) {
    if (handle < 0 || handle >= (int64_t)ctx->blocks.size() || !ctx->blocks[handle].live) { ctx->err = "bad handle"; return 1; }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    QueryBlock &qb = ctx->blocks[handle];
    HIP_TRY(ctx, hipMemcpyAsync(out, qb.out, (size_t)qb.n * sizeof(apples_placement), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

int apples_placements_device_ptr(apples_ctx *ctx, int64_t handle, void **ptr) {
    if (handle < 0 || handle >= (int64_t)ctx->blocks.size() || !ctx->blocks[handle].live) { ctx->err = "bad handle"; return 1; }
    *ptr = ctx->blocks[handle].out;
    return 0;
}

// Host buffer in, placements left on the device: the queries are uploaded and packed chunk by chunk on
// a second stream while the kernels of the previous chunk run (Feeder above), so the copy of all but the
// first chunk costs nothing.
int apples_place_sequences_streamed(apples_ctx *ctx, const uint8_t *queries, int64_t n_queries, const int32_t *self_row,
                                    int64_t *handle) {
    if (!ctx->has_aln) { ctx->err = "context has no alignment"; return 1; }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    QueryBlock qb;
    const int planes = ctx->aln.planes;
    if (alloc_block(ctx, n_queries, self_row, planes, &qb, ctx->stream2)) { free_block(ctx, &qb); return 1; }
    qb.live = true;
    Feeder feed{queries};
    int rc = run_block(ctx, qb, &feed);
    int exotic = 0;
    if (!rc && ctx->params.model != APPLES_SCOREDIST) rc = take_exotic(ctx, ctx->stream2, &exotic);
    if (!rc && exotic && planes == 2 && ctx->aln.ex_ok) {
        // a query carried a symbol beyond ACGT-, which the pass took for a gap: those queries once more through the 8-plane
        // kernels (a few: a block of their own; many: the whole block) -- the context keeps its matrix-core forms
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        rc = exotic_queries(ctx, &qb, queries, self_row);
        if (!rc) rc = qb.exact8 ? run_block(ctx, qb) : run_ex_block(ctx, qb);
    } else if (!rc && exotic && planes == 2) {
        // a query carried a symbol beyond ACGT-: the 2-plane images are not valid for it.  Rare: widen
        // the reference to raw bytes and run the block again from a whole-block upload
        free_block(ctx, &qb);
        rc = make_block(ctx, queries, n_queries, self_row, &qb);
        if (!rc) rc = run_block(ctx, qb);
    }
    if (rc) { (void)hipDeviceSynchronize(); free_block(ctx, &qb); return 1; }
    const int slot = new_block_slot(ctx);
    ctx->blocks[slot] = qb;
    *handle = (int64_t)slot;
    return 0;
}

int apples_place_from_sequences(apples_ctx *ctx, const uint8_t *queries, int64_t n_queries, const int32_t *self_row,
                                apples_placement *out) {
    if (n_queries == 0) return 0;
    int64_t h;
    if (apples_place_sequences_streamed(ctx, queries, n_queries, self_row, &h)) return 1;
    int rc = apples_fetch_placements(ctx, h, out);
    free_block(ctx, &ctx->blocks[h]);
    return rc;
}

int apples_distances_resident(apples_ctx *ctx, int64_t handle, int32_t query_tile) {
    if (handle < 0 || handle >= (int64_t)ctx->blocks.size() || !ctx->blocks[handle].live) { ctx->err = "bad handle"; return 1; }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    QueryBlock &qb = ctx->blocks[handle];
    const DevAlign &a = ctx->aln;
    if (ensure_workspace(ctx, a.n_refs, a.slots_pad, qb.n, true, false, false)) return 1;
    Workspace &w = ctx->ws;
    PhaseTimer pt{ctx};
    int launches = 0;
    for (int64_t q0 = 0; q0 < qb.n; q0 += w.batch) {
        int64_t nq = std::min(w.batch, qb.n - q0);
        pt.flush();
        pt.begin(APPLES_T_DIST);
        if (ctx->params.model == APPLES_SCOREDIST) {
            if (launch_scoredist(ctx, qb, q0, nq, w.dist, nullptr)) return 1;
        } else {
            if (launch_counts(ctx, qb, q0, nq, query_tile > 0 ? query_tile : dist_tile_for(ctx, nq), w.dist, nullptr)) return 1;
        }
        pt.end(APPLES_T_DIST);
        ++launches;
    }
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    pt.flush();
    for (int i = 0; i < APPLES_T_COUNT; ++i) ctx->t_ms[i] = pt.acc[i];
    ctx->t_ms[APPLES_T_TOTAL] = pt.acc[APPLES_T_DIST];
    ctx->t_ms[APPLES_T_DIST_LAUNCHES] = launches;
    return 0;
}

int apples_distances(apples_ctx *ctx, const uint8_t *queries, int64_t n_queries, uint32_t *out_counts, double *out_dist) {
    if (!ctx->has_aln) { ctx->err = "context has no alignment"; return 1; }
    if (n_queries == 0) return 0;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const DevAlign &a = ctx->aln;
    int64_t h;
    if (apples_queries_upload(ctx, queries, n_queries, nullptr, &h)) return 1;
    QueryBlock &qb = ctx->blocks[h];
    int rc = ensure_workspace(ctx, a.n_refs, a.slots_pad, n_queries, true, out_counts != nullptr, false);
    Workspace &w = ctx->ws;
    std::vector<double> hd;
    std::vector<uint32_t> hc;
    for (int64_t q0 = 0; !rc && q0 < qb.n; q0 += w.batch) {
        int64_t nq = std::min(w.batch, qb.n - q0);
        if (ctx->params.model == APPLES_SCOREDIST) rc = launch_scoredist(ctx, qb, q0, nq, w.dist, out_counts ? w.counts : nullptr);
        else rc = launch_counts(ctx, qb, q0, nq, dist_tile_for(ctx, nq), w.dist, out_counts ? w.counts : nullptr);
        if (rc) break;
        hd.resize((size_t)nq * a.slots_pad);
        if (hipMemcpyAsync(hd.data(), w.dist, hd.size() * 8, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) { ctx->err = "copy back failed"; rc = 1; break; }
        if (out_counts) {
            hc.resize((size_t)nq * a.slots_pad);
            if (hipMemcpyAsync(hc.data(), w.counts, hc.size() * 4, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) { ctx->err = "copy back failed"; rc = 1; break; }
        }
        if (hipStreamSynchronize(ctx->stream) != hipSuccess) { ctx->err = "distance kernel failed"; rc = 1; break; }
        for (int64_t q = 0; q < nq; ++q)
            for (int64_t s = 0; s < a.n_rows; ++s) {
                int64_t r = a.slot_row[s];
                if (out_dist) out_dist[(q0 + q) * a.n_rows + r] = hd[(size_t)q * a.slots_pad + s];
                if (out_counts) {
                    uint32_t c = hc[(size_t)q * a.slots_pad + s];
                    if (ctx->params.model == APPLES_SCOREDIST) {
                        out_counts[((q0 + q) * a.n_rows + r) * 2] = 0;
                        out_counts[((q0 + q) * a.n_rows + r) * 2 + 1] = c;
                    } else {
                        out_counts[((q0 + q) * a.n_rows + r) * 2] = c >> 16;
                        out_counts[((q0 + q) * a.n_rows + r) * 2 + 1] = c & 0xffffu;
                    }
                }
            }
    }
    free_block(ctx, &ctx->blocks[h]);
    return rc;
}

// column layout of a distance table: columns sorted by tree level (deepest first), cached while
// the caller keeps passing the same col_node
static int setup_columns(apples_ctx *ctx, int64_t n_cols, const int32_t *col_node) {
    const DevTree &t = ctx->tree;
    bool same = ctx->dcols == n_cols && (int64_t)ctx->h_col_node.size() == n_cols &&
                std::equal(col_node, col_node + n_cols, ctx->h_col_node.begin());
    if (same) return 0;
    std::vector<int32_t> level(t.n_nodes);
    HIP_TRY(ctx, hipMemcpyAsync(level.data(), t.level, (size_t)t.n_nodes * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    for (int64_t c = 0; c < n_cols; ++c)
        if (col_node[c] >= t.n_nodes) { ctx->err = "col_node out of range"; return 1; }
    if (ctx->tree.merge_ok) {  // two columns on one tree leaf: no merged level lists (the workspace is laid out again)
        std::vector<uint8_t> seen(t.n_nodes, 0);
        for (int64_t c = 0; c < n_cols; ++c) {
            const int nd = col_node[c];
            if (nd < 0) continue;
            if (seen[nd]) {
                ctx->tree.merge_ok = false;
                HIP_TRY(ctx, hipDeviceSynchronize());
                free_workspace(ctx->ws);
                break;
            }
            seen[nd] = 1;
        }
    }
    std::vector<int32_t> &perm = ctx->h_col_perm;
    perm = level_order(col_node, n_cols, level, ctx->tree.scan);
    std::vector<int32_t> s_node(n_cols), s_level(n_cols);
    for (int64_t s = 0; s < n_cols; ++s) {
        int nd = col_node[perm[s]];
        s_node[s] = nd;
        s_level[s] = nd >= 0 ? level[nd] : -1;
    }
    dev_free(ctx->d_col_perm); dev_free(ctx->d_col_node); dev_free(ctx->d_col_level);
    if (dev_upload(ctx, &ctx->d_col_perm, perm.data(), n_cols)) return 1;
    if (dev_upload(ctx, &ctx->d_col_node, s_node.data(), n_cols)) return 1;
    if (dev_upload(ctx, &ctx->d_col_level, s_level.data(), n_cols)) return 1;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    ctx->h_col_node.assign(col_node, col_node + n_cols);
    ctx->dcols = n_cols;
    ++ctx->col_gen;  // resident tables permuted with the previous layout are stale from here on
    return 0;
}

// selection (table rule) + sweep over nq rows of a device-resident table in slot order; d_self = per-query own
// column as a slot index or -1
// `pipe` (run_table_block's pipelined form): the selection on the context's stream between pipe[0] and pipe[1], the sweep on
// the back stream (stream3) behind `ready`, between pipe[2] and pipe[3]; no phase timer
static int run_table_batch(apples_ctx *ctx, const double *d_rows, int64_t nq, int64_t n_cols, const int32_t *d_self,
                           apples_placement *d_out, PhaseTimer &pt, hipEvent_t *pipe = nullptr, hipEvent_t ready = nullptr) {
    Workspace &w = ctx->ws;
    ctx->cur_batch_queries = nq;  // (route_threshold)
    SelectArgs s{};
    s.dist = d_rows; s.stride = n_cols; s.gather = nullptr;  // rows were permuted into slot order
    s.slot_node = ctx->d_col_node; s.slot_level = ctx->d_col_level; s.slot_rep = ctx->d_col_perm;
    s.node_level = ctx->tree.level;
    s.n_members = n_cols; s.n_reps = n_cols; s.all_singleton = 1; s.table_mode = 1; s.self_slot = d_self;
    s.cols_all_in_tree = std::all_of(ctx->h_col_node.begin(), ctx->h_col_node.end(), [](int32_t v) { return v >= 0; });
    s.thr = ctx->params.filt_threshold; s.baseobs = ctx->params.base_observation; s.height = ctx->tree.height;
    s.obs_node = w.obs_node; s.obs_dist = w.obs_dist; s.obs_cap = w.obs_cap; s.n_obs = w.n_obs;
    s.cnt_gt = ctx->tree.scan ? nullptr : w.cnt_gt;
    s.out = d_out;
    s.big_threshold = route_threshold(ctx); s.overflow_list = w.route_list; s.overflow_count = w.route_count;
    s.route_classes = (w.big.lean && !hybrid_records(ctx)) ? 1 : 0;
    s.cls_list = w.cls_list; s.cls_count = w.cls_count; s.cls_stride = w.batch;
    s.row_cursor = w.cls_count + 21;
    HIP_TRY(ctx, hipMemsetAsync(w.cls_count, 0, 64 * sizeof(int32_t), ctx->stream));  // every counter of the batch
    if (pipe) {
        HIP_TRY(ctx, hipEventRecord(pipe[0], ctx->stream));
        if (launch_select(ctx, s, nq)) return 1;
        HIP_TRY(ctx, hipEventRecord(pipe[1], ctx->stream));
        HIP_TRY(ctx, hipEventRecord(ready, ctx->stream));
        HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream3, ready, 0));
        HIP_TRY(ctx, hipEventRecord(pipe[2], ctx->stream3));
        if (run_sweep(ctx, d_out, nq, ctx->stream3)) return 1;
        HIP_TRY(ctx, hipEventRecord(pipe[3], ctx->stream3));
        return 0;
    }
    pt.flush();
    pt.begin(APPLES_T_SELECT);
    if (launch_select(ctx, s, nq)) return 1;
    pt.end(APPLES_T_SELECT);
    pt.begin(APPLES_T_SWEEP);
    if (run_sweep(ctx, d_out, nq)) return 1;
    pt.end(APPLES_T_SWEEP);
    return 0;
}

int apples_place_from_distances(apples_ctx *ctx, const double *dist, int64_t n_queries, int64_t n_cols,
                                const int32_t *col_node, const int32_t *self_col, apples_placement *out) {
    if (n_queries == 0) return 0;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (setup_columns(ctx, n_cols, col_node)) return 1;
    const std::vector<int32_t> &perm = ctx->h_col_perm;
    std::vector<int32_t> col_slot(n_cols);
    for (int64_t s = 0; s < n_cols; ++s) col_slot[perm[s]] = (int32_t)s;
    const bool hybrid = hybrid_records(ctx);
    if (ensure_workspace(ctx, n_cols, n_cols, n_queries, true, false, hybrid)) return 1;
    Workspace &w = ctx->ws;
    apples_placement *d_out = nullptr;
    int32_t *d_self = nullptr;
    double *d_stage = nullptr;
    if (dev_alloc(ctx, &d_out, w.batch)) return 1;
    if (dev_alloc(ctx, &d_self, w.batch)) return 1;
    if (dev_alloc(ctx, &d_stage, std::min<int64_t>(w.batch, n_queries) * n_cols)) return 1;
    PhaseTimer pt{ctx};
    int rc = 0;
    for (int64_t q0 = 0; q0 < n_queries && !rc; q0 += w.batch) {
        int64_t nq = std::min(w.batch, n_queries - q0);
        std::vector<int32_t> self(nq, -1);
        if (self_col)
            for (int64_t i = 0; i < nq; ++i)
                if (self_col[q0 + i] >= 0 && self_col[q0 + i] < n_cols) self[i] = col_slot[self_col[q0 + i]];
        if (hipMemcpyAsync(d_self, self.data(), (size_t)nq * 4, hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
            hipMemcpyAsync(d_stage, dist + q0 * n_cols, (size_t)nq * n_cols * 8, hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
            hipStreamSynchronize(ctx->stream) != hipSuccess) { ctx->err = "upload of the distance table failed"; rc = 1; break; }
        rc = launch_permute_cols(ctx, d_stage, w.dist, ctx->d_col_perm, nq, n_cols);
        if (rc) break;
        rc = run_table_batch(ctx, w.dist, nq, n_cols, d_self, d_out, pt);
        if (rc) break;
        if (hipMemcpyAsync(out + q0, d_out, (size_t)nq * sizeof(apples_placement), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
            hipStreamSynchronize(ctx->stream) != hipSuccess) { ctx->err = "placement kernels failed"; rc = 1; break; }
    }
    pt.flush();
    for (int i = 0; i < APPLES_T_COUNT; ++i) ctx->t_ms[i] = pt.acc[i];
    ctx->t_ms[APPLES_T_TOTAL] = pt.acc[APPLES_T_SELECT] + pt.acc[APPLES_T_SWEEP];
    dev_free(d_out);
    dev_free(d_self);
    dev_free(d_stage);
    return rc;
}

int apples_table_upload(apples_ctx *ctx, const double *dist, int64_t n_queries, int64_t n_cols, const int32_t *col_node,
                        const int32_t *self_col, int64_t *handle) {
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (setup_columns(ctx, n_cols, col_node)) return 1;
    const std::vector<int32_t> &perm = ctx->h_col_perm;
    std::vector<int32_t> col_slot(n_cols);
    for (int64_t s = 0; s < n_cols; ++s) col_slot[perm[s]] = (int32_t)s;
    std::vector<int32_t> self(std::max<int64_t>(n_queries, 1), -1);
    if (self_col)
        for (int64_t i = 0; i < n_queries; ++i)
            if (self_col[i] >= 0 && self_col[i] < n_cols) self[i] = col_slot[self_col[i]];
    QueryBlock qb;
    qb.n = n_queries;
    qb.n_cols = n_cols;
    qb.col_gen = ctx->col_gen;
    double *d_stage = nullptr;
    if (dev_upload(ctx, &d_stage, dist, n_queries * n_cols) || dev_alloc(ctx, &qb.table, n_queries * n_cols) ||
        dev_upload(ctx, &qb.self_slot, self.data(), (int64_t)self.size()) ||
        dev_alloc(ctx, &qb.out, std::max<int64_t>(n_queries, 1)) ||
        launch_permute_cols(ctx, d_stage, qb.table, ctx->d_col_perm, n_queries, n_cols)) {
        dev_free(d_stage);
        free_block(ctx, &qb);
        return 1;
    }
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    dev_free(d_stage);
    qb.live = true;
    const int slot = new_block_slot(ctx);
    ctx->blocks[slot] = qb;
    *handle = (int64_t)slot;
    return 0;
}

static int run_table_block(apples_ctx *ctx, QueryBlock &qb) {
    const bool hybrid = hybrid_records(ctx);
    if (qb.n_cols != ctx->dcols || qb.col_gen != ctx->col_gen) {
        ctx->err = "the column layout changed since this table was uploaded (another table with different columns was "
                   "placed on this context): upload the table again";
        return 1;
    }
    // APPLES_TABLE_PIPELINE=k (an experiment, off by default): the block as a pipeline of k sub-batches with two sets of batch
    // buffers, the selection of sub-batch i + 1 -- a stream over the table, bound by HBM -- beside the sweep of sub-batch i, which
    // waits on memory round trips.  Measured: a sweep takes about as long for 2 048 rows as for 4 096 (its
    // longest queries decide), so k sweeps cost more than the overlap returns.
    const int n_pipe = (int)knob(ctx, "APPLES_TABLE_PIPELINE", 0);  // sub-batches; 0 / 1: off
    const bool pipelined = n_pipe > 1 && !hybrid && !ctx->tree.scan && qb.n >= 2048;
    int64_t want = qb.n;
    if (pipelined) want = round_up((qb.n + n_pipe - 1) / n_pipe, 32);
    if (pipelined && back_stream(ctx)) return 1;
    if (ensure_workspace(ctx, qb.n_cols, qb.n_cols, want, false, false, hybrid, false, pipelined)) return 1;
    Workspace &w = ctx->ws;
    const int64_t step = pipelined ? std::min<int64_t>(w.batch, want) : w.batch;
    const int64_t n_sub = (qb.n + step - 1) / step;
    PhaseTimer pt{ctx};
    while (ctx->ev_pool.size() < (size_t)(2 + (pipelined ? 4 * n_sub : 0))) {  // timing events live with the context
        hipEvent_t e;
        HIP_TRY(ctx, hipEventCreate(&e));
        ctx->ev_pool.push_back(e);
    }
    hipEvent_t e_start = ctx->ev_pool[0], e_stop = ctx->ev_pool[1];
    HIP_TRY(ctx, hipEventRecord(e_start, ctx->stream));
    for (int64_t i = 0; i < n_sub; ++i) {
        const int64_t q0 = i * step, nq = std::min(step, qb.n - q0);
        if (!pipelined) {
            if (run_table_batch(ctx, qb.table + q0 * qb.n_cols, nq, qb.n_cols, qb.self_slot + q0, qb.out + q0, pt)) return 1;
            continue;
        }
        const int set = (int)(i & 1);
        if (i > 0) swap_bufs(w);  // host view: w.* now names buffer set `set`
        if (i >= 2) HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_back[set], 0));  // the sweep that read this set is done
        if (run_table_batch(ctx, qb.table + q0 * qb.n_cols, nq, qb.n_cols, qb.self_slot + q0, qb.out + q0, pt,
                            ctx->ev_pool.data() + 2 + 4 * i, ctx->ev_front[set])) return 1;
        HIP_TRY(ctx, hipEventRecord(ctx->ev_back[set], ctx->stream3));
    }
    if (pipelined) {
        if ((n_sub & 1) == 0) swap_bufs(w);  // leave the host view on set 0
        HIP_TRY(ctx, hipEventRecord(ctx->ev_bigfree, ctx->stream3));
        HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_bigfree, 0));
    }
    HIP_TRY(ctx, hipEventRecord(e_stop, ctx->stream));
    HIP_TRY(ctx, hipEventSynchronize(e_stop));
    pt.flush();
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e_start, e_stop);
    for (int i = 0; i < APPLES_T_COUNT; ++i) ctx->t_ms[i] = pt.acc[i];
    if (pipelined)
        for (int64_t i = 0; i < n_sub; ++i) {  // each kernel family's own time; the two overlap, so they add up to more than the total
            float a = 0, b = 0;
            (void)hipEventElapsedTime(&a, ctx->ev_pool[2 + 4 * i], ctx->ev_pool[2 + 4 * i + 1]);
            (void)hipEventElapsedTime(&b, ctx->ev_pool[2 + 4 * i + 2], ctx->ev_pool[2 + 4 * i + 3]);
            ctx->t_ms[APPLES_T_SELECT] += a;
            ctx->t_ms[APPLES_T_SWEEP] += b;
        }
    ctx->t_ms[APPLES_T_TOTAL] = ms;
    return 0;
}

static int sweep_edges_scan(apples_ctx *ctx, const int32_t *obs_node, const double *obs_dist, int32_t n_obs, uint8_t *valid,
                            double *S, double *R, double *x, double *err, int32_t *lca, apples_placement *out) {
    const DevTree &t = ctx->tree;
    std::vector<int32_t> level;  // unused by the id ordering
    std::vector<int32_t> ord = level_order(obs_node, n_obs, level, true);
    std::vector<int32_t> s_node(n_obs);
    std::vector<double> s_dist(n_obs);
    for (int i = 0; i < n_obs; ++i) { s_node[i] = obs_node[ord[i]]; s_dist[i] = obs_dist[ord[i]]; }
    for (int i = 1; i < n_obs; ++i)
        if (s_node[i] == s_node[i - 1]) { ctx->err = "an observed leaf is listed twice"; return 1; }
    if (ensure_workspace(ctx, std::max<int64_t>(n_obs, ctx->ws.obs_cap), std::max<int64_t>(n_obs, 1), 1, true, false, true)) return 1;
    Workspace &w = ctx->ws;
    apples_placement *d_out = nullptr;
    if (dev_alloc(ctx, &d_out, 1)) return 1;
    apples_placement init{};
    init.n_obs = n_obs;
    HIP_TRY(ctx, hipMemcpy(d_out, &init, sizeof(init), hipMemcpyHostToDevice));
    HIP_TRY(ctx, hipMemcpy(w.obs_node, s_node.data(), (size_t)n_obs * 4, hipMemcpyHostToDevice));
    HIP_TRY(ctx, hipMemcpy(w.obs_dist, s_dist.data(), (size_t)n_obs * 8, hipMemcpyHostToDevice));
    HIP_TRY(ctx, hipMemcpy(w.n_obs, &n_obs, 4, hipMemcpyHostToDevice));
    HIP_TRY(ctx, hipMemsetAsync(w.cls_count, 0, 64 * sizeof(int32_t), ctx->stream));
    ScanArgs sa = scan_args(ctx, w.big, d_out, true, true);
    sa.overflow_list = nullptr; sa.overflow_count = nullptr;
    if (launch_scan(ctx, sa, 1, 1, 256, ctx->stream)) { dev_free(d_out); return 1; }
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    apples_placement res;
    HIP_TRY(ctx, hipMemcpy(&res, d_out, sizeof(res), hipMemcpyDeviceToHost));
    dev_free(d_out);
    if (out) *out = res;
    int32_t meta[4];
    HIP_TRY(ctx, hipMemcpy(meta, w.big.meta, sizeof meta, hipMemcpyDeviceToHost));
    const int V = meta[0];
    if (lca) *lca = meta[1];
    std::vector<double> xe((size_t)std::max(V, 1) * 18);
    HIP_TRY(ctx, hipMemcpy(xe.data(), w.big.xe, (size_t)V * 144, hipMemcpyDeviceToHost));
    if (valid) memset(valid, 0, t.n_nodes);
    for (int i = 0; i < V; ++i) {
        const double *xp = &xe[(size_t)i * 18];
        const int v = (int)(xp[17] < 0 ? -xp[17] : xp[17]) - 1;  // the record's last slot carries the node id
        if (valid) valid[v] = 1;
        if (S) memcpy(S + (size_t)v * 6, xp + 11, 48);
        if (R) memcpy(R + (size_t)v * 6, xp + 5, 48);
        if (x) memcpy(x + (size_t)v * 4, xp, 32);
        if (err) err[v] = xp[4];
    }
    return 0;
}

int apples_sweep_edges(apples_ctx *ctx, const int32_t *obs_node, const double *obs_dist, int32_t n_obs, uint8_t *valid,
                       double *S, double *R, double *x, double *err, int32_t *lca, apples_placement *out) {
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const DevTree &t = ctx->tree;
    if (n_obs < 2) { ctx->err = "need at least two observed leaves"; return 1; }
    for (int i = 0; i < n_obs; ++i)
        if (obs_node[i] < 0 || obs_node[i] >= t.n_nodes) { ctx->err = "obs_node out of range"; return 1; }
    if (t.scan) return sweep_edges_scan(ctx, obs_node, obs_dist, n_obs, valid, S, R, x, err, lca, out);
    std::vector<int32_t> level(t.n_nodes);
    HIP_TRY(ctx, hipMemcpy(level.data(), t.level, (size_t)t.n_nodes * 4, hipMemcpyDeviceToHost));
    for (int i = 0; i < n_obs; ++i)
        if (obs_node[i] < 0 || obs_node[i] >= t.n_nodes) { ctx->err = "obs_node out of range"; return 1; }
    std::vector<int32_t> ord = level_order(obs_node, n_obs, level);
    std::vector<int32_t> s_node(n_obs);
    std::vector<double> s_dist(n_obs);
    for (int i = 0; i < n_obs; ++i) { s_node[i] = obs_node[ord[i]]; s_dist[i] = obs_dist[ord[i]]; }
    std::vector<int32_t> cg(t.height + 2);
    for (int l = -1; l <= t.height; ++l) {
        int c = 0;
        for (int i = 0; i < n_obs; ++i) c += level[s_node[i]] > l;
        cg[l + 1] = c;
    }
    if (ensure_workspace(ctx, std::max<int64_t>(n_obs, ctx->ws.obs_cap), std::max<int64_t>(n_obs, 1), 1, true, false, true)) return 1;
    Workspace &w = ctx->ws;
    apples_placement *d_out = nullptr;
    if (dev_alloc(ctx, &d_out, 1)) return 1;
    apples_placement init{};
    init.n_obs = n_obs;
    HIP_TRY(ctx, hipMemcpy(d_out, &init, sizeof(init), hipMemcpyHostToDevice));
    HIP_TRY(ctx, hipMemcpy(w.obs_node, s_node.data(), (size_t)n_obs * 4, hipMemcpyHostToDevice));
    HIP_TRY(ctx, hipMemcpy(w.obs_dist, s_dist.data(), (size_t)n_obs * 8, hipMemcpyHostToDevice));
    HIP_TRY(ctx, hipMemcpy(w.cnt_gt, cg.data(), cg.size() * 4, hipMemcpyHostToDevice));
    HIP_TRY(ctx, hipMemcpy(w.n_obs, &n_obs, 4, hipMemcpyHostToDevice));
    HIP_TRY(ctx, hipMemsetAsync(w.cls_count, 0, 64 * sizeof(int32_t), ctx->stream));
    if (launch_sweep(ctx, sweep_args(ctx, w.big, d_out, true), 1, 1, 256)) { dev_free(d_out); return 1; }
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    apples_placement res;
    HIP_TRY(ctx, hipMemcpy(&res, d_out, sizeof(res), hipMemcpyDeviceToHost));
    dev_free(d_out);
    if (out) *out = res;
    int V = res.n_valid;
    int32_t VI = 0;
    HIP_TRY(ctx, hipMemcpy(&VI, w.big.grp_off + t.height + 2, 4, hipMemcpyDeviceToHost));
    if (lca) HIP_TRY(ctx, hipMemcpy(lca, w.big.grp_off + t.height + 3, 4, hipMemcpyDeviceToHost));
    // internal nodes: compact records; observed leaves: rebuilt from the level-sorted list
    std::vector<int32_t> order(V);
    std::vector<double> hS((size_t)V * 6), hR((size_t)V * 6), hx((size_t)V * 5), ha((size_t)std::max(VI, 1) * 8);
    std::vector<double> hxi((size_t)std::max(VI, 1) * 18), hxl((size_t)n_obs * 18);
    HIP_TRY(ctx, hipMemcpy(ha.data(), w.big.A, (size_t)VI * 64, hipMemcpyDeviceToHost));
    HIP_TRY(ctx, hipMemcpy(hxi.data(), w.big.xe, (size_t)VI * 144, hipMemcpyDeviceToHost));
    HIP_TRY(ctx, hipMemcpy(hxl.data(), w.big.xe + (size_t)w.big.cap * 18, (size_t)n_obs * 144, hipMemcpyDeviceToHost));
    std::vector<int32_t> level_h;
    for (int i = 0; i < V; ++i) {
        const double *xr;
        if (i < VI) {
            memcpy(&hS[(size_t)i * 6], &hxi[(size_t)i * 18 + 11], 48);  // the record's tuple is R by now
            int32_t nd;
            memcpy(&nd, reinterpret_cast<const char *>(&ha[(size_t)i * 8]) + 56, 4);
            order[i] = nd;
            xr = &hxi[(size_t)i * 18];
        } else {
            int j = i - VI;
            order[i] = s_node[j];
            double D = s_dist[j];
            double *S6 = &hS[(size_t)i * 6];
            S6[0] = 1; S6[1] = 0; S6[2] = 0; S6[3] = 0;
            int m = ctx->params.method;
            if (m == APPLES_FM) { S6[4] = 1.0 / D; S6[5] = 1.0 / (D * D); }
            else if (m == APPLES_BE) { S6[4] = D; S6[5] = 1.0 / D; }
            else { S6[4] = D * D; S6[5] = D; }
            xr = &hxl[(size_t)j * 18];
        }
        memcpy(&hx[(size_t)i * 5], xr, 40);
        memcpy(&hR[(size_t)i * 6], xr + 5, 48);
    }
    if (valid) memset(valid, 0, t.n_nodes);
    for (int i = 0; i < V; ++i) {
        int v = order[i];
        if (valid) valid[v] = 1;
        if (S) memcpy(S + (size_t)v * 6, &hS[(size_t)i * 6], 48);
        if (R) memcpy(R + (size_t)v * 6, &hR[(size_t)i * 6], 48);
        if (x) memcpy(x + (size_t)v * 4, &hx[(size_t)i * 5], 32);
        if (err) err[v] = hx[(size_t)i * 5 + 4];
    }
    return 0;
}

int apples_last_timing(const apples_ctx *ctx, double *ms, int32_t n) {
    for (int i = 0; i < n && i < APPLES_T_COUNT; ++i) ms[i] = ctx->t_ms[i];
    return n < APPLES_T_COUNT ? n : APPLES_T_COUNT;
}

const char *apples_describe(apples_ctx *ctx) {
    hipDeviceProp_t prop;
    char buf[2560];
    const char *name = "?";
    int cus = 0;
    if (hipGetDeviceProperties(&prop, ctx->device) == hipSuccess) { name = prop.name; cus = prop.multiProcessorCount; }
    const DevAlign &a = ctx->aln;
    int32_t blk_cnt[3] = {0, 0, 0};  // clade blocks, the last device batch: (query, cluster) items, -, tiles
    if (ctx->blk_counters && hipStreamSynchronize(ctx->stream) == hipSuccess)
        (void)hipMemcpy(blk_cnt, ctx->blk_counters, sizeof blk_cnt, hipMemcpyDeviceToHost);
    int32_t big_used = 0;  // ragged rows (common.h, Workspace::ragged): the big rows the last device batch handed out
    if (ctx->ws.ragged && hipStreamSynchronize(ctx->stream) == hipSuccess)
        (void)hipMemcpy(&big_used, ctx->ws.cls_count + 22, sizeof big_used, hipMemcpyDeviceToHost);
    snprintf(buf, sizeof buf,
             "{\"device\": \"%s\", \"compute_units\": %d, \"n_nodes\": %d, \"height\": %d, \"max_children\": %d, \"n_rows\": %lld, "
             "\"n_refs\": %lld, \"n_reps\": %lld, \"length\": %d, \"code_planes\": %d, \"all_singleton\": %d, "
             "\"packed_bytes\": %lld, \"batch\": %lld, \"sweep_workgroups\": %d, \"sweep_team_cap\": %lld, \"sweep_big_workgroups\": %d, \"jc_lut\": %d, \"sweep\": \"%s\", "
             "\"fused_distance_pass\": \"%s\", \"fp4_reference_image_bytes\": %lld, \"sweep_layout\": \"%s\", \"cluster_fused\": %d, "
             "\"scoredist_filter\": %d, \"scoredist_image_bytes\": %lld, \"cluster_blocks\": %d, \"block_items_last_batch\": %d, \"block_tiles_last_batch\": %d, "
             "\"exotic_symbols_as_gaps\": %d, \"exotic_sites_max_per_row\": %d, \"eight_plane_copy\": %d, "
             "\"ragged_rows\": %d, \"row_small\": %lld, \"big_rows\": %lld, \"big_rows_last_batch\": %d, \"full_rows_for_good\": %d}",
             name, cus, ctx->tree.n_nodes, ctx->tree.height, ctx->tree.max_children, (long long)a.n_rows, (long long)a.n_refs,
             (long long)a.n_reps, a.L, a.planes, a.all_singleton ? 1 : 0,
             (long long)((int64_t)a.G * (a.planes + 1) * a.slots_pad * 16), (long long)ctx->ws.batch, ctx->ws.small.wgs,
             (long long)ctx->ws.small.cap, ctx->ws.big.wgs, ctx->jc_lut ? 1 : 0, ctx->tree.scan ? "scan" : "levels",
             // which kernel the fused pass of ACGT- query blocks runs on (dist_gemm.hip / dist.hip k_jc69_mfma / k_jc69 or k_scoredist)
             dist_gemm_usable(ctx) ? (ctx->gemm_thr.ok ? "fp4 gemm, linear threshold" : "fp4 gemm, threshold table")
                                   : (a.planes == 2 && dist_mfma_enabled(ctx) ? "fp4 mfma, bit-plane fed" : "valu"),
             (long long)(a.ref_f4 ? a.slots_pad * (int64_t)a.G * 128 : 0),
             // how the level-loop sweep knows a query's subtree: merged level lists / node bits in LDS / tagged node map
             // (lean: sweep_lean.hip on big binary trees -- the workspace decides; before there is one, what a plain MLSE / ME pass will get)
             ctx->tree.scan ? "scan" : (ctx->ws.small.lean || (ctx->ws.batch == 0 && sweep_lean_layout(ctx->tree, false))) ? "lean"
             : sweep_merge_lists(ctx->tree) ? "merge" : sweep_bits_in_lds(ctx->tree) ? "bits" : "map",
             // clustered references with the panels of the fast path (representatives for the matrix-core pass, members cluster-major)
             (!a.all_singleton && !(ctx->dbg & APPLES_DBG_NO_FUSE) &&
              ((a.rep_packed && a.packed_rm && ctx->params.model == APPLES_JC69) || (a.aa_rep_idx && ctx->params.model == APPLES_SCOREDIST))) ? 1 : 0,
             // scoredist: the fused pass filters on the matrix cores (dist_sd.hip) at the present threshold
             (ctx->params.model == APPLES_SCOREDIST && sd_gemm_usable(ctx) && !(ctx->dbg & APPLES_DBG_NO_FUSE)) ? 1 : 0,
             (long long)(a.sd_ref4 ? a.slots_pad * (int64_t)sd_steps(a.L) * 64 : 0),
             // clade blocks of a clustered reference (build_blocks): whole subtrees of one cluster, swept on a static schedule
             (int)a.n_blocks, (int)blk_cnt[0], (int)blk_cnt[2],
             // bytes beyond ACGT- (DevAlign::ex_ok): the context keeps them as gaps in its 2-plane rows and fp4 images; the most such
             // sites in one reference row; the 8-plane copy exists (built with the first such byte, in the reference or in a query)
             a.ex_ok ? 1 : 0, (int)a.ex_max, a.packed8 ? 1 : 0,
             ctx->ws.ragged ? 1 : 0, (long long)ctx->ws.row_small, (long long)ctx->ws.row_big, (int)big_used, ctx->no_ragged ? 1 : 0);
    ctx->desc = buf;
    return ctx->desc.c_str();
}

}  // extern "C"
