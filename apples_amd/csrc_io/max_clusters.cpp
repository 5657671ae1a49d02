// Max-diameter tree clustering behind include/apples_io.h: the sweep of apples_amd/treecluster.py
// (the TreeCluster "max" method the reference runs as an external tool, apples/Reference.py:87-88),
// statement for statement, on the tree's CSR arrays.
#include <algorithm>
#include <cstddef>
#include <cstdint>
#include <deque>
#include <utility>
#include <vector>

#include "apples_io.h"

extern "C" int apples_max_clusters(int32_t n_nodes, const int32_t *child_off, const int32_t *child_idx,
                                   const double *edge_len, int32_t root, double threshold, int32_t *leaf_order,
                                   int32_t *cluster_end, int32_t *n_clusters) {
    const int32_t n = n_nodes;
    std::vector<std::vector<int32_t>> children((std::size_t)n);
    int max_deg = 0;
    for (int32_t v = 0; v < n; ++v) {
        children[v].assign(child_idx + child_off[v], child_idx + child_off[v + 1]);
        if ((int)children[v].size() > max_deg) max_deg = (int)children[v].size();
    }
    std::vector<double> elen(edge_len, edge_len + n);
    // polytomies -> zero-length binary nodes: repeatedly replace the last two children by a new parent of them
    if (n > 0 && max_deg > 2) {
        std::deque<int32_t> q{root};
        while (!q.empty()) {
            const int32_t v = q.front();
            q.pop_front();
            while (children[v].size() > 2) {
                const int32_t c1 = children[v].back(); children[v].pop_back();
                const int32_t c2 = children[v].back(); children[v].pop_back();
                const int32_t nid = (int32_t)children.size();
                children.push_back({c1, c2});  // (may reallocate: children[v] is looked up again below)
                elen.push_back(0.0);
                children[v].push_back(nid);
            }
            for (int32_t c : children[v]) q.push_back(c);
        }
    }
    const int32_t total = (int32_t)children.size();
    // post-order: the ids themselves when no node was added
    std::vector<int32_t> order;
    order.reserve((std::size_t)total);
    if (total == n) {
        for (int32_t v = 0; v < n; ++v) order.push_back(v);
    } else {
        std::vector<std::pair<int32_t, int32_t>> st{{root, 0}};
        while (!st.empty()) {
            auto [v, i] = st.back();
            st.pop_back();
            if (i < (int32_t)children[v].size()) {
                st.push_back({v, i + 1});
                st.push_back({children[v][i], 0});
            } else {
                order.push_back(v);
            }
        }
    }
    std::vector<uint8_t> deleted((std::size_t)total, 0);
    std::vector<double> left((std::size_t)total, 0.0), right((std::size_t)total, 0.0);
    int32_t n_out = 0, n_cl = 0;
    cluster_end[0] = 0;
    std::vector<int32_t> stack;
    // marks v's remaining subtree deleted; its leaves, left to right, are appended to leaf_order
    auto cut = [&](int32_t v) -> int32_t {
        int32_t added = 0;
        stack.assign(1, v);
        while (!stack.empty()) {
            const int32_t u = stack.back();
            stack.pop_back();
            if (deleted[u]) continue;
            deleted[u] = 1;
            if (children[u].empty()) { leaf_order[n_out++] = u; ++added; }
            for (auto it = children[u].rbegin(); it != children[u].rend(); ++it) stack.push_back(*it);
        }
        return added;
    };
    auto close = [&](int32_t added) { if (added > 0) cluster_end[++n_cl] = n_out; };
    for (int32_t v : order) {
        if (deleted[v]) continue;
        const std::vector<int32_t> &ch = children[v];
        if (ch.empty()) { left[v] = right[v] = 0.0; continue; }
        if (ch.size() == 1) {  // unifurcation: pass the child's depth through
            const int32_t c = ch[0];
            left[v] = deleted[c] ? 0.0 : (left[c] > right[c] ? left[c] : right[c]) + elen[c];
            right[v] = 0.0;
            continue;
        }
        const int32_t a = ch[0], b = ch[1];
        if (deleted[a] && deleted[b]) {
            const int32_t before = n_out;
            cut(v);
            n_out = before;  // (nothing below is left; as the Python sweep, the result is not a cluster)
            continue;
        }
        left[v] = deleted[a] ? 0.0 : (left[a] > right[a] ? left[a] : right[a]) + elen[a];
        right[v] = deleted[b] ? 0.0 : (left[b] > right[b] ? left[b] : right[b]) + elen[b];
        if (left[v] + right[v] > threshold) {
            if (left[v] > right[v]) { close(cut(a)); left[v] = 0.0; }
            else { close(cut(b)); right[v] = 0.0; }
        }
    }
    if (n > 0) close(cut(root));
    *n_clusters = n_cl;
    return 0;
}

// Consensus rows of the multi-member clusters (apples/PoolRepresentativeWorker.py:17-85): per column the most frequent symbol of the
// alphabet among the members' rows, ties to the first in alphabet order, symbols outside the alphabet not counted (a column with
// none of them: the alphabet's first symbol, as numpy's argmax of zeros).  Cluster c = rows member_row[member_off[c] ..
// member_off[c + 1]) of `seqs` ([n_rows][L] bytes); out = [n_clusters][L].  Threads take clusters in turn.
#include <atomic>
#include <thread>

extern "C" int apples_consensus(const uint8_t *seqs, int64_t L, const int32_t *member_row, const int64_t *member_off,
                                int64_t n_clusters, const uint8_t *alphabet, int32_t n_alpha, uint8_t *out, int32_t n_threads) {
    if (n_alpha <= 0 || n_alpha > 32) return 1;
    uint8_t lut[256];
    for (int i = 0; i < 256; ++i) lut[i] = 255;
    for (int a = n_alpha - 1; a >= 0; --a) lut[alphabet[a]] = (uint8_t)a;
    int T = n_threads > 0 ? n_threads : (int)std::thread::hardware_concurrency();
    T = std::max(1, std::min<int>(T, 64));
    if (n_clusters < 64) T = 1;
    std::atomic<int64_t> next{0};
    auto work = [&]() {
        std::vector<uint32_t> cnt((size_t)n_alpha * L);
        for (;;) {
            const int64_t c = next.fetch_add(1);
            if (c >= n_clusters) break;
            std::fill(cnt.begin(), cnt.end(), 0u);
            for (int64_t m = member_off[c]; m < member_off[c + 1]; ++m) {
                const uint8_t *row = seqs + (int64_t)member_row[m] * L;
                for (int64_t s = 0; s < L; ++s) {
                    const uint8_t a = lut[row[s]];
                    if (a != 255) ++cnt[(size_t)a * L + s];
                }
            }
            uint8_t *o = out + c * L;
            for (int64_t s = 0; s < L; ++s) {
                int best = 0;
                uint32_t bc = cnt[s];
                for (int a = 1; a < n_alpha; ++a)
                    if (cnt[(size_t)a * L + s] > bc) { bc = cnt[(size_t)a * L + s]; best = a; }
                o[s] = alphabet[best];
            }
        }
    };
    std::vector<std::thread> th;
    for (int t = 1; t < T; ++t) th.emplace_back(work);
    work();
    for (auto &x : th) x.join();
    return 0;
}
