// uc_setcover.cpp — stage E7: greedy set cover on the host (the north star keeps it host-side).
// Stands for MMseqs2 `clust --cluster-mode 0` inside `foldseek cluster` (cluster.rs:45-56).
// Rule (spec UC-1): undirected graph on accepted pairs; repeatedly the unassigned node covering the most
// unassigned nodes (its neighbours + itself; ties: smallest id) becomes a representative and takes all its
// unassigned neighbours.  Implementation: CSR adjacency + monotone bucket queue with lazy min-heaps.
#include <algorithm>
#include <cstdint>
#include <functional>
#include <queue>
#include <vector>

#include "uc_common.h"

namespace uc {

void set_cover(uint32_t n, const uint32_t *edges, uint64_t n_edges, uint32_t *assign) {
    std::vector<uint64_t> key;
    key.reserve(2 * n_edges);
    for (uint64_t e = 0; e < n_edges; e++) {
        uint32_t a = edges[2 * e], b = edges[2 * e + 1];
        if (a >= n || b >= n) fail(UC_ERR_ARGS, "set cover: edge (%u,%u) out of range", a, b);
        if (a == b) continue;
        key.push_back(((uint64_t)a << 32) | b);
        key.push_back(((uint64_t)b << 32) | a);
    }
    std::sort(key.begin(), key.end());
    key.erase(std::unique(key.begin(), key.end()), key.end());
    std::vector<uint64_t> aoff((size_t)n + 1, 0);
    for (uint64_t k : key) aoff[(k >> 32) + 1]++;
    for (uint32_t i = 0; i < n; i++) aoff[i + 1] += aoff[i];
    std::vector<uint32_t> adj(key.size());
    for (size_t i = 0; i < key.size(); i++) adj[i] = (uint32_t)key[i];
    std::vector<uint64_t>().swap(key);

    constexpr uint32_t NONE = UINT32_MAX;
    std::vector<uint32_t> cnt(n);
    uint32_t maxc = 1;
    for (uint32_t i = 0; i < n; i++) { cnt[i] = (uint32_t)(aoff[i + 1] - aoff[i]) + 1; maxc = std::max(maxc, cnt[i]); assign[i] = NONE; }
    using MinHeap = std::priority_queue<uint32_t, std::vector<uint32_t>, std::greater<uint32_t>>;
    std::vector<MinHeap> bucket((size_t)maxc + 1);
    {   // initial fill in one shot per bucket (ids ascending -> heapify is cheap)
        std::vector<std::vector<uint32_t>> init((size_t)maxc + 1);
        for (uint32_t i = 0; i < n; i++) init[cnt[i]].push_back(i);
        for (uint32_t c = 0; c <= maxc; c++) bucket[c] = MinHeap(std::greater<uint32_t>(), std::move(init[c]));
    }
    std::vector<uint32_t> newly;
    for (uint32_t c = maxc; c >= 1;) {
        if (bucket[c].empty()) { c--; continue; }
        uint32_t u = bucket[c].top();
        bucket[c].pop();
        if (assign[u] != NONE || cnt[u] != c) continue;   // stale entry
        newly.clear();
        assign[u] = u;
        newly.push_back(u);
        for (uint64_t k = aoff[u]; k < aoff[u + 1]; k++) {
            uint32_t v = adj[k];
            if (assign[v] == NONE) { assign[v] = u; newly.push_back(v); }
        }
        for (uint32_t v : newly)
            for (uint64_t k = aoff[v]; k < aoff[v + 1]; k++) {
                uint32_t w = adj[k];
                if (assign[w] == NONE) { cnt[w]--; bucket[cnt[w]].push(w); }
            }
    }
    for (uint32_t i = 0; i < n; i++)
        if (assign[i] == NONE) assign[i] = i;   // unreachable: every node has cnt >= 1
}

}  // namespace uc
