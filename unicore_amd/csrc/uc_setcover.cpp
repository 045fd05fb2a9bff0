// uc_setcover.cpp — stage E7: greedy set cover on the host (the north star keeps it host-side).
// Stands for MMseqs2 `clust --cluster-mode 0` inside `foldseek cluster` (cluster.rs:45-56).
// Rule (spec UC-1): undirected graph on accepted pairs; repeatedly the unassigned node covering the most
// unassigned nodes (its neighbours + itself; ties: smallest id) becomes a representative and takes all its
// unassigned neighbours.
// Implementation: CSR adjacency by counting sort (+ per-node sort/unique, threaded), then a monotone bucket
// queue: counts only ever decrease, so buckets are drained from the top; a node sits in ONE bucket (a min-heap
// by id) and is moved down lazily when it surfaces with a stale count — no push per decrement.
#include <algorithm>
#include <cstdint>
#include <functional>
#include <queue>
#include <thread>
#include <vector>

#include "uc_common.h"

namespace uc {

void set_cover(uint32_t n, const uint32_t *edges, uint64_t n_edges, uint32_t *assign) {
    // ---- adjacency (both directions, self loops dropped), duplicates removed per node ----
    std::vector<uint64_t> off((size_t)n + 2, 0);
    for (uint64_t e = 0; e < n_edges; e++) {
        const uint32_t a = edges[2 * e], b = edges[2 * e + 1];
        if (a >= n || b >= n) fail(UC_ERR_ARGS, "set cover: edge (%u,%u) out of range", a, b);
        if (a == b) continue;
        off[a + 2]++; off[b + 2]++;
    }
    for (uint32_t i = 0; i < n; i++) off[i + 2] += off[i + 1];
    std::vector<uint32_t> adj(off[n + 1]);
    for (uint64_t e = 0; e < n_edges; e++) {       // off[i+1] is the fill cursor of node i
        const uint32_t a = edges[2 * e], b = edges[2 * e + 1];
        if (a == b) continue;
        adj[off[a + 1]++] = b;
        adj[off[b + 1]++] = a;
    }
    // now off[i] .. off[i+1] is node i's (unsorted, possibly duplicated) list
    std::vector<uint32_t> deg(n);
    {
        const unsigned nthr = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
        auto work = [&](uint32_t lo, uint32_t hi) {
            for (uint32_t i = lo; i < hi; i++) {
                uint32_t *b = adj.data() + off[i], *e = adj.data() + off[i + 1];
                std::sort(b, e);
                deg[i] = (uint32_t)(std::unique(b, e) - b);
            }
        };
        if (n < 4096 || nthr == 1) work(0, n);
        else {
            std::vector<std::thread> th;
            for (unsigned t = 0; t < nthr; t++)
                th.emplace_back(work, (uint32_t)((uint64_t)n * t / nthr), (uint32_t)((uint64_t)n * (t + 1) / nthr));
            for (auto &x : th) x.join();
        }
    }

    constexpr uint32_t NONE = UINT32_MAX;
    std::vector<uint32_t> cnt(n);
    uint32_t maxc = 1;
    for (uint32_t i = 0; i < n; i++) { cnt[i] = deg[i] + 1; maxc = std::max(maxc, cnt[i]); assign[i] = NONE; }
    using MinHeap = std::priority_queue<uint32_t, std::vector<uint32_t>, std::greater<uint32_t>>;
    std::vector<MinHeap> bucket((size_t)maxc + 1);
    {
        std::vector<std::vector<uint32_t>> init((size_t)maxc + 1);
        for (uint32_t i = 0; i < n; i++) init[cnt[i]].push_back(i);     // ascending ids: already a valid heap
        for (uint32_t c = 0; c <= maxc; c++) bucket[c] = MinHeap(std::greater<uint32_t>(), std::move(init[c]));
    }
    std::vector<uint32_t> newly;
    for (uint32_t c = maxc; c >= 1;) {
        if (bucket[c].empty()) { c--; continue; }
        const uint32_t u = bucket[c].top();
        bucket[c].pop();
        if (assign[u] != NONE) continue;
        if (cnt[u] != c) { bucket[cnt[u]].push(u); continue; }   // stale: move down lazily
        newly.clear();
        assign[u] = u;
        newly.push_back(u);
        for (uint64_t k = off[u]; k < off[u] + deg[u]; k++) {
            const uint32_t v = adj[k];
            if (assign[v] == NONE) { assign[v] = u; newly.push_back(v); }
        }
        for (uint32_t v : newly)
            for (uint64_t k = off[v]; k < off[v] + deg[v]; k++) {
                const uint32_t w = adj[k];
                if (assign[w] == NONE) cnt[w]--;
            }
    }
    for (uint32_t i = 0; i < n; i++)
        if (assign[i] == NONE) assign[i] = i;   // unreachable: every node has cnt >= 1
}

}  // namespace uc
