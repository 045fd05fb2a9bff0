// uc_setcover.cpp — stage E7: greedy set cover on the host (the north star keeps it host-side).
// Stands for MMseqs2 `clust --cluster-mode 0` inside `foldseek cluster` (cluster.rs:45-56).
// Rule (spec UC-1): undirected graph on accepted pairs; repeatedly the unassigned node covering the most
// unassigned nodes (its neighbours + itself; ties: smallest id) becomes a representative and takes all its
// unassigned neighbours.
// Implementation: CSR adjacency by counting sort (+ per-node sort/unique, threaded), then a monotone bucket
// queue: counts only ever decrease, so buckets are drained from the top; a node sits in ONE bucket (a min-heap
// by id) and is moved down lazily when it surfaces with a stale count — no push per decrement.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <functional>
#include <memory>
#include <queue>
#include <thread>
#include <vector>

#include "uc_common.h"

namespace uc {

// the greedy cover on a CSR graph (deg[i] entries of node i start at off[i])
static void greedy_cover(uint32_t n, const uint64_t *off, const uint32_t *adj, const uint32_t *deg, uint32_t *assign) {
    // Monotone bucket queue with lazy move-down.  Counts only ever decrease and a pick never raises another node's count, so
    // when the sweep reaches bucket c its content is final: it is sorted once (ascending id = the tie-break) and scanned;
    // a node whose count dropped since it was filed is re-filed in its current (lower) bucket when it is met.
    constexpr uint32_t NONE = UINT32_MAX;
    std::vector<uint32_t> cnt(n);
    uint32_t maxc = 1;
    for (uint32_t i = 0; i < n; i++) { cnt[i] = deg[i] + 1; maxc = std::max(maxc, cnt[i]); assign[i] = NONE; }
    std::vector<uint32_t> fill((size_t)maxc + 2, 0);
    for (uint32_t i = 0; i < n; i++) fill[cnt[i] + 1]++;
    for (uint32_t c = 0; c <= maxc; c++) fill[c + 1] += fill[c];
    std::vector<uint32_t> init(n);                       // counting sort: ascending ids inside every bucket
    {
        std::vector<uint32_t> cur(fill.begin(), fill.end() - 1);
        for (uint32_t i = 0; i < n; i++) init[cur[cnt[i]]++] = i;
    }
    std::vector<std::vector<uint32_t>> moved((size_t)maxc + 1);   // nodes re-filed into a bucket (arbitrary order)
    std::vector<uint32_t> cand, newly;
    for (uint32_t c = maxc; c >= 1; c--) {
        const uint32_t *ib = init.data() + fill[c], *ie = init.data() + fill[c + 1];
        std::vector<uint32_t> &mv = moved[c];
        const uint32_t *list = ib;
        size_t len = (size_t)(ie - ib);
        if (!mv.empty()) {                               // merge the re-filed nodes in
            std::sort(mv.begin(), mv.end());
            cand.resize(len + mv.size());
            std::merge(ib, ie, mv.begin(), mv.end(), cand.begin());
            list = cand.data();
            len = cand.size();
        }
        for (size_t k = 0; k < len; k++) {
            const uint32_t u = list[k];
            if (assign[u] != NONE) continue;
            if (cnt[u] != c) { moved[cnt[u]].push_back(u); continue; }   // stale: cnt[u] < c
            newly.clear();
            assign[u] = u;
            newly.push_back(u);
            for (uint64_t e = off[u]; e < off[u] + deg[u]; e++) {
                const uint32_t v = adj[e];
                if (assign[v] == NONE) { assign[v] = u; newly.push_back(v); }
            }
            for (uint32_t v : newly)
                for (uint64_t e = off[v]; e < off[v] + deg[v]; e++) {
                    const uint32_t w = adj[e];
                    if (assign[w] == NONE) cnt[w]--;
                }
        }
        std::vector<uint32_t>().swap(mv);
    }
    for (uint32_t i = 0; i < n; i++)
        if (assign[i] == NONE) assign[i] = i;   // unreachable: every node has cnt >= 1
}

void set_cover(uint32_t n, const uint32_t *edges, uint64_t n_edges, uint32_t *assign) {
    const bool timing = getenv("UC_TIMING") != nullptr;
    auto t0 = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!timing) return;
        auto t1 = std::chrono::steady_clock::now();
        fprintf(stderr, "set_cover: %-10s %.2f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t0).count());
        t0 = t1;
    };
    // ---- adjacency (both directions, self loops dropped), duplicates removed per node ----
    // threaded counting sort with relaxed atomic counters / cursors (the order inside a node's list is irrelevant:
    // every list is sorted next, so the result does not depend on the interleaving)
    const unsigned nthr_all = std::max(1u, std::min(32u, std::thread::hardware_concurrency()));
    const unsigned T = (n_edges < (1u << 16) || n < 1024) ? 1u : std::min(16u, nthr_all);
    std::vector<std::atomic<uint32_t>> cnt_a(n);
    for (uint32_t i = 0; i < n; i++) cnt_a[i].store(0, std::memory_order_relaxed);
    std::vector<int> bad(T, 0);
    auto run_threads = [&](auto &&fn) {
        if (T == 1) { fn(0u); return; }
        std::vector<std::thread> th;
        for (unsigned t = 0; t < T; t++) th.emplace_back(fn, t);
        for (auto &x : th) x.join();
    };
    run_threads([&](unsigned t) {
        const uint64_t lo = n_edges * t / T, hi = n_edges * (t + 1) / T;
        for (uint64_t e = lo; e < hi; e++) {
            const uint32_t a = edges[2 * e], b = edges[2 * e + 1];
            if (a >= n || b >= n) { bad[t] = 1; return; }
            if (a == b) continue;
            cnt_a[a].fetch_add(1, std::memory_order_relaxed);
            cnt_a[b].fetch_add(1, std::memory_order_relaxed);
        }
    });
    for (unsigned t = 0; t < T; t++)
        if (bad[t]) fail(UC_ERR_ARGS, "set cover: edge endpoint out of range");
    std::vector<uint64_t> off((size_t)n + 1, 0);
    for (uint32_t i = 0; i < n; i++) off[i + 1] = off[i] + cnt_a[i].load(std::memory_order_relaxed);
    // uninitialised on purpose: a zero-filled 33 MB vector is first-touched (page-faulted) by ONE thread, ~10 ms at C2
    std::unique_ptr<uint32_t[]> adj_mem(new uint32_t[std::max<uint64_t>(off[n], 1)]);
    uint32_t *adj = adj_mem.get();
    for (uint32_t i = 0; i < n; i++) cnt_a[i].store(0, std::memory_order_relaxed);   // now the fill cursor of node i
    run_threads([&](unsigned t) {
        const uint64_t lo = n_edges * t / T, hi = n_edges * (t + 1) / T;
        for (uint64_t e = lo; e < hi; e++) {
            const uint32_t a = edges[2 * e], b = edges[2 * e + 1];
            if (a == b) continue;
            adj[off[a] + cnt_a[a].fetch_add(1, std::memory_order_relaxed)] = b;
            adj[off[b] + cnt_a[b].fetch_add(1, std::memory_order_relaxed)] = a;
        }
    });
    lap("csr");
    // now off[i] .. off[i+1] is node i's (unsorted, possibly duplicated) list
    std::vector<uint32_t> deg(n);
    {
        const unsigned nthr = nthr_all;
        auto work = [&](uint32_t lo, uint32_t hi) {
            for (uint32_t i = lo; i < hi; i++) {
                uint32_t *b = adj + off[i], *e = adj + off[i + 1];
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

    lap("sort");
    greedy_cover(n, off.data(), adj, deg.data(), assign);
    lap("greedy");
}

}  // namespace uc
