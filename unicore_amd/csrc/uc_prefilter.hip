// uc_prefilter.hip — stages E1-E4 on the GPU: k-mer index build, similar-k-mer matching with the
// same-diagonal double-hit rule, ungapped rescoring, per-query top-M selection.
// Stands for the MMseqs2-style prefilter inside `foldseek cluster`
// (call site /root/reference/src/modules/cluster.rs:45-56; algorithm SURVEY.md A.2; spec UC-1 E1-E4).
//
// All of this is HBM-bound scan/sort work:
//   E1  extract (k-mer, seq, pos) per target residue (coalesced scan of the 3Di track) -> radix sort by
//       k-mer -> CSR offsets by binary search per k-mer slot (20^6 + 1 u32 = 256 MB, lives in HBM/MALL).
//   E2  per query residue: enumerate similar k-mers (sorted-letter DFS with score bound; tables in LDS),
//       gather each k-mer's CSR range, emit one 64-bit key (query | target | diagonal) per hit at an offset
//       fixed by a count + exclusive-scan pre-pass (no atomics, deterministic), radix-sort the keys, then one
//       pass over the sorted keys run-length-counts diagonals per (query,target) and keeps the best one.
//   E3  ungapped diagonal score per surviving candidate.   E4  sort by (query, score desc, target) on the
//       GPU; the final per-query truncation to max_seqs is a linear host pass over the sorted list.
// Sort and scan primitives come from rocPRIM; every kernel below is hand-written for wave64.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "uc_engine.h"

namespace uc {

constexpr uint32_t KMER_INVALID = 0xFFFFFFFFu;

struct KmerCfg {
    int koff[K];
    int span;
    int thr;
};

// sequence id containing padded residue offset p, searched in off[lo..hi)
__device__ __forceinline__ uint32_t find_seq(const uint32_t *off, uint32_t lo, uint32_t hi, uint32_t p) {
    while (hi - lo > 1) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        if (off[mid] <= p) lo = mid; else hi = mid;
    }
    return lo;
}

// ---------------------------------------------------------------- E1: index build
__global__ void __launch_bounds__(256) kmer_extract_kernel(const DeviceDb db, uint32_t tbegin, uint32_t tend, KmerCfg cfg,
                                                           uint32_t p0, uint32_t p1, uint32_t *keys, uint64_t *vals) {
    for (uint64_t idx = (uint64_t)blockIdx.x * 256 + threadIdx.x; idx < p1 - p0; idx += (uint64_t)gridDim.x * 256) {
        const uint32_t p = p0 + (uint32_t)idx;
        const uint32_t s = find_seq(db.off, tbegin, tend, p);
        const uint32_t j = p - db.off[s], len = db.len[s];
        uint32_t key = KMER_INVALID;
        if (j + cfg.span <= len && j <= 65535u) {
            uint32_t v = 0, mul = 1;
            bool ok = true;
#pragma unroll
            for (int m = 0; m < K; m++) {
                const uint32_t c = db.s3[p + cfg.koff[m]];
                ok &= c < KA;
                v += c * mul;
                mul *= KA;
            }
            if (ok) key = v;
        }
        keys[idx] = key;
        vals[idx] = ((uint64_t)s << 16) | j;
    }
}

__global__ void __launch_bounds__(256) kmer_offsets_kernel(const uint32_t *keys, uint32_t n, uint32_t *koff) {
    for (uint64_t k = (uint64_t)blockIdx.x * 256 + threadIdx.x; k <= KSPACE; k += (uint64_t)gridDim.x * 256) {
        uint32_t lo = 0, hi = n;   // first index with key >= k
        while (lo < hi) {
            const uint32_t mid = lo + ((hi - lo) >> 1);
            if (keys[mid] < (uint32_t)k) lo = mid + 1; else hi = mid;
        }
        koff[k] = lo;
    }
}

__global__ void __launch_bounds__(256) split_entries_kernel(const uint64_t *vals, uint32_t n, uint32_t *eseq, uint16_t *epos) {
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
        const uint64_t v = vals[i];
        eseq[i] = (uint32_t)(v >> 16);
        epos[i] = (uint16_t)(v & 0xFFFF);
    }
}

// ---------------------------------------------------------------- E2: similar k-mers
struct SimTables {           // LDS: per query letter a, target letters sorted by score descending
    int8_t sc[KA][KA];
    uint8_t ord[KA][KA];
    int8_t rowmax[KA];
};

__device__ void build_sim_tables(SimTables &t, const int8_t *S3) {
    for (int a = threadIdx.x; a < KA; a += blockDim.x) {
        int8_t sc[KA];
        uint8_t ord[KA];
        for (int b = 0; b < KA; b++) { sc[b] = S3[a * 21 + b]; ord[b] = (uint8_t)b; }
        for (int i = 1; i < KA; i++) {   // insertion sort: score desc, letter asc
            const int8_t s = sc[i];
            const uint8_t o = ord[i];
            int k = i - 1;
            while (k >= 0 && (sc[k] < s)) { sc[k + 1] = sc[k]; ord[k + 1] = ord[k]; k--; }
            sc[k + 1] = s;
            ord[k + 1] = o;
        }
        for (int b = 0; b < KA; b++) { t.sc[a][b] = sc[b]; t.ord[a][b] = ord[b]; }
        t.rowmax[a] = sc[0];
    }
}

// calls f(kmer_value) for every k-mer with sum_m S3[c[m]][c'[m]] >= thr
template <typename F>
__device__ __forceinline__ void for_each_similar(const SimTables &t, const uint32_t c[K], int thr, F &&f) {
    int rest[K + 1];
    rest[K] = 0;
#pragma unroll
    for (int m = K - 1; m >= 0; m--) rest[m] = rest[m + 1] + t.rowmax[c[m]];
    if (rest[0] < thr) return;
    for (int k0 = 0; k0 < KA; k0++) {
        const int s0 = t.sc[c[0]][k0];
        if (s0 + rest[1] < thr) break;
        const uint32_t v0 = t.ord[c[0]][k0];
        for (int k1 = 0; k1 < KA; k1++) {
            const int s1 = s0 + t.sc[c[1]][k1];
            if (s1 + rest[2] < thr) break;
            const uint32_t v1 = v0 + t.ord[c[1]][k1] * 20u;
            for (int k2 = 0; k2 < KA; k2++) {
                const int s2 = s1 + t.sc[c[2]][k2];
                if (s2 + rest[3] < thr) break;
                const uint32_t v2 = v1 + t.ord[c[2]][k2] * 400u;
                for (int k3 = 0; k3 < KA; k3++) {
                    const int s3 = s2 + t.sc[c[3]][k3];
                    if (s3 + rest[4] < thr) break;
                    const uint32_t v3 = v2 + t.ord[c[3]][k3] * 8000u;
                    for (int k4 = 0; k4 < KA; k4++) {
                        const int s4 = s3 + t.sc[c[4]][k4];
                        if (s4 + rest[5] < thr) break;
                        const uint32_t v4 = v3 + t.ord[c[4]][k4] * 160000u;
                        for (int k5 = 0; k5 < KA; k5++) {
                            if (s4 + t.sc[c[5]][k5] < thr) break;
                            f(v4 + t.ord[c[5]][k5] * 3200000u);
                        }
                    }
                }
            }
        }
    }
}

// shared by the count and the emit pass: decode the query position handled by this thread
__device__ __forceinline__ bool query_kmer_at(const DeviceDb &db, const KmerCfg &cfg, uint32_t qbegin, uint32_t qend,
                                              uint32_t p, uint32_t *q, uint32_t *i, uint32_t c[K]) {
    const uint32_t s = find_seq(db.off, qbegin, qend, p);
    const uint32_t j = p - db.off[s], len = db.len[s];
    *q = s;
    *i = j;
    if (!(j + cfg.span <= len && j <= 65535u)) return false;
    bool ok = true;
#pragma unroll
    for (int m = 0; m < K; m++) { c[m] = db.s3[p + cfg.koff[m]]; ok &= c[m] < KA; }
    return ok;
}

__global__ void __launch_bounds__(256) sim_count_kernel(const DeviceDb db, KmerCfg cfg, uint32_t qbegin, uint32_t qend,
                                                        uint32_t p0, uint32_t p1, const uint32_t *koff, uint32_t *cnt,
                                                        unsigned long long *n_sim_total) {
    __shared__ SimTables tab;
    build_sim_tables(tab, db.S3);
    __syncthreads();
    unsigned long long nsim = 0;
    for (uint64_t idx = (uint64_t)blockIdx.x * 256 + threadIdx.x; idx < p1 - p0; idx += (uint64_t)gridDim.x * 256) {
        uint32_t q, i, c[K], n = 0;
        if (query_kmer_at(db, cfg, qbegin, qend, p0 + (uint32_t)idx, &q, &i, c))
            for_each_similar(tab, c, cfg.thr, [&](uint32_t v) { n += koff[v + 1] - koff[v]; nsim++; });
        cnt[idx] = n;
    }
    // one atomic per wave
    for (int o = 32; o > 0; o >>= 1) nsim += __shfl_down(nsim, o, 64);
    if ((threadIdx.x & 63) == 0 && nsim) atomicAdd(n_sim_total, nsim);
}

// hit key: [ query - qbegin : 23 | target : 24 | diag + 65536 : 17 ]
__global__ void __launch_bounds__(256) sim_emit_kernel(const DeviceDb db, KmerCfg cfg, uint32_t qbegin, uint32_t qend,
                                                       uint32_t p0, uint32_t p1, const uint32_t *koff,
                                                       const uint32_t *eseq, const uint16_t *epos,
                                                       const uint64_t *hoff, uint64_t *keys) {
    __shared__ SimTables tab;
    build_sim_tables(tab, db.S3);
    __syncthreads();
    for (uint64_t idx = (uint64_t)blockIdx.x * 256 + threadIdx.x; idx < p1 - p0; idx += (uint64_t)gridDim.x * 256) {
        uint32_t q, i, c[K];
        if (!query_kmer_at(db, cfg, qbegin, qend, p0 + (uint32_t)idx, &q, &i, c)) continue;
        uint64_t w = hoff[idx];
        const uint64_t qbits = (uint64_t)(q - qbegin) << 41;
        for_each_similar(tab, c, cfg.thr, [&](uint32_t v) {
            const uint32_t e1 = koff[v + 1];
            for (uint32_t e = koff[v]; e < e1; e++)
                keys[w++] = qbits | ((uint64_t)eseq[e] << 17) | (uint64_t)((int)i - (int)epos[e] + 65536);
        });
    }
}

// one pass over the sorted hit keys: the first key of every (query,target) group walks its group,
// run-length-counts the diagonals and keeps the best (count desc, diagonal asc); flag = candidate
__global__ void __launch_bounds__(256) diag_select_kernel(const uint64_t *keys, uint64_t n, int min_hits,
                                                          uint32_t *flag, int32_t *bestdiag) {
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
        const uint64_t k = keys[i], grp = k >> 17;
        uint32_t fl = 0;
        if (i == 0 || (keys[i - 1] >> 17) != grp) {
            int best_cnt = 0, best_d = 0;
            uint64_t b = i;
            while (b < n && (keys[b] >> 17) == grp) {
                const uint64_t kb = keys[b];
                uint64_t e = b + 1;
                while (e < n && keys[e] == kb) e++;
                const int c = (int)(e - b);
                if (c > best_cnt) { best_cnt = c; best_d = (int)(kb & 0x1FFFF) - 65536; }
                b = e;
            }
            if (best_cnt >= min_hits) { fl = 1; bestdiag[i] = best_d; }
        }
        flag[i] = fl;
    }
}

__global__ void __launch_bounds__(256) cand_scatter_kernel(const uint64_t *keys, uint64_t n, const uint32_t *flag,
                                                           const uint64_t *pos, const int32_t *bestdiag, uint32_t qbegin,
                                                           uint32_t *cq, uint32_t *ct, int32_t *cd) {
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
        if (!flag[i]) continue;
        const uint64_t k = keys[i], w = pos[i];
        cq[w] = qbegin + (uint32_t)(k >> 41);
        ct[w] = (uint32_t)(k >> 17) & 0xFFFFFFu;
        cd[w] = bestdiag[i];
    }
}

// E4 sort key: [ query : 32 | 255 - score : 8 | target : 24 ]; score < min -> all-ones (sorted last)
__global__ void __launch_bounds__(256) select_key_kernel(uint64_t n, const uint32_t *cq, const uint32_t *ct, const int32_t *score,
                                                         int min_score, uint64_t *key, unsigned long long *n_kept) {
    unsigned long long kept = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
        const int s = score[i];
        const bool ok = s >= min_score;
        key[i] = ok ? ((uint64_t)cq[i] << 32) | ((uint64_t)(255 - s) << 24) | ct[i] : ~0ull;
        kept += ok;
    }
    for (int o = 32; o > 0; o >>= 1) kept += __shfl_down(kept, o, 64);
    if ((threadIdx.x & 63) == 0 && kept) atomicAdd(n_kept, kept);
}

// E4 truncation: entry i of the (query, score desc, target) sorted list survives iff its rank inside its
// query's run is < max_seqs
__global__ void __launch_bounds__(256) rank_flag_kernel(const uint64_t *skey, uint64_t n, uint32_t max_seqs, uint32_t *flag) {
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
        const uint64_t qk = skey[i] & 0xFFFFFFFF00000000ull;
        uint64_t lo = 0, hi = i;   // first index of this query's run
        while (lo < hi) { const uint64_t m = (lo + hi) >> 1; if (skey[m] < qk) lo = m + 1; else hi = m; }
        flag[i] = (i - lo) < max_seqs ? 1u : 0u;
    }
}

__global__ void __launch_bounds__(256) hit_scatter_kernel(const uint64_t *skey, const int32_t *sdiag, uint64_t n, const uint32_t *flag,
                                                          const uint64_t *pos, uint32_t *hq, uint32_t *ht, int32_t *hs, int32_t *hd) {
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
        if (!flag[i]) continue;
        const uint64_t k = skey[i], w = pos[i];
        hq[w] = (uint32_t)(k >> 32);
        ht[w] = (uint32_t)(k & 0xFFFFFFu);
        hs[w] = 255 - (int32_t)((k >> 24) & 0xFF);
        hd[w] = sdiag[i];
    }
}

// per-query hit counts from the query-sorted hit array
__global__ void __launch_bounds__(256) hit_count_kernel(const uint32_t *hq, uint64_t n_hits, uint32_t n, uint32_t *cnt) {
    for (uint32_t q = blockIdx.x * 256 + threadIdx.x; q < n; q += gridDim.x * 256) {
        uint64_t lo = 0, hi = n_hits;
        while (lo < hi) { const uint64_t m = (lo + hi) >> 1; if (hq[m] < q) lo = m + 1; else hi = m; }
        const uint64_t b = lo;
        hi = n_hits;
        while (lo < hi) { const uint64_t m = (lo + hi) >> 1; if (hq[m] <= q) lo = m + 1; else hi = m; }
        cnt[q] = (uint32_t)(lo - b);
    }
}

static inline dim3 grid_for(uint64_t n, uint32_t cap = 16384) {
    const uint64_t b = (n + 255) / 256;
    return dim3((uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(b, cap)));
}

struct WidenU32 {
    __host__ __device__ uint64_t operator()(uint32_t x) const { return (uint64_t)x; }
};

void Engine::prefilter(uint32_t tbegin, uint32_t tend) {
    if (!have_db) fail(UC_ERR_ARGS, "no database loaded");
    if (tbegin > tend || tend > hdb.n) fail(UC_ERR_ARGS, "prefilter: bad target range");
    UC_HIP(hipSetDevice(device));
    const uint32_t n = hdb.n;
    KmerCfg cfg;
    for (int m = 0; m < K; m++) cfg.koff[m] = p.koff[m];
    cfg.span = p.span;
    cfg.thr = p.kmer_thr;

    hit_cnt.assign(n, 0);
    hit_off.assign((size_t)n + 1, 0);
    n_hits = 0;
    alns_valid = false;
    edges.clear();

    DevBuf<unsigned long long> d_counters;   // [0] sim k-mers, [1] kept candidates, [2] ungapped overlap residues
    d_counters.reserve(3);
    UC_HIP(hipMemsetAsync(d_counters.p, 0, 24, stream));
    DevBuf<char> d_temp;
    auto temp_reserve = [&](size_t bytes) { d_temp.reserve(bytes + 256); };

    // ------------------------------------------------------------ E1: index of targets [tbegin, tend)
    Timer t_index;
    timed_ms_begin();
    const uint32_t tp0 = h_poff[tbegin], tp1 = h_poff[tend];
    const uint32_t nres = tp1 - tp0;
    DevBuf<uint32_t> d_koff, d_eseq;
    DevBuf<uint16_t> d_epos;
    d_koff.reserve((size_t)KSPACE + 1);
    uint32_t n_entries = 0;
    {
        DevBuf<uint32_t> k_in, k_out;
        DevBuf<uint64_t> v_in, v_out;
        const size_t cap = std::max<uint32_t>(nres, 1);
        k_in.reserve(cap); k_out.reserve(cap); v_in.reserve(cap); v_out.reserve(cap);
        if (nres) {
            hipLaunchKernelGGL(kmer_extract_kernel, grid_for(nres), dim3(256), 0, stream, ddb, tbegin, tend, cfg, tp0, tp1, k_in.p, v_in.p);
            size_t tb = 0;
            UC_HIP(rocprim::radix_sort_pairs(nullptr, tb, k_in.p, k_out.p, v_in.p, v_out.p, (size_t)nres, 0u, 32u, stream));
            temp_reserve(tb);
            UC_HIP(rocprim::radix_sort_pairs(d_temp.p, tb, k_in.p, k_out.p, v_in.p, v_out.p, (size_t)nres, 0u, 32u, stream));
        }
        // number of valid entries = first index with key >= KSPACE: reuse the offsets kernel's last slot
        hipLaunchKernelGGL(kmer_offsets_kernel, grid_for((uint64_t)KSPACE + 1), dim3(256), 0, stream, k_out.p, nres, d_koff.p);
        UC_HIP(hipMemcpyAsync(&n_entries, d_koff.p + KSPACE, 4, hipMemcpyDeviceToHost, stream));
        UC_HIP(hipStreamSynchronize(stream));
        d_eseq.reserve(std::max<uint32_t>(n_entries, 1));
        d_epos.reserve(std::max<uint32_t>(n_entries, 1));
        if (n_entries)
            hipLaunchKernelGGL(split_entries_kernel, grid_for(n_entries), dim3(256), 0, stream, v_out.p, n_entries, d_eseq.p, d_epos.p);
        UC_HIP(hipStreamSynchronize(stream));
    }
    UC_HIP(hipGetLastError());
    double gpu_ms = timed_ms_end();
    stats.n_index_entries += n_entries;
    stats.algorithmic_bytes[UC_ST_INDEX] += 6ull * n_entries + 8ull * KSPACE;
    stats.stage_seconds[UC_ST_INDEX] += t_index.seconds();

    // ------------------------------------------------------------ E2-E4 over query batches
    const uint64_t HIT_CAP = 1ull << 30;       // keys per batch (8 GiB + 8 GiB sort double buffer)
    double hits_per_res = 64.0;                // adaptive estimate
    DevBuf<uint32_t> d_cnt, d_flag, d_cq, d_ct;
    DevBuf<uint64_t> d_hoff, d_keys, d_keys2, d_pos, d_skey, d_skey2;
    DevBuf<int32_t> d_bestdiag, d_cd, d_cd2, d_score;
    uint64_t n_hits_total = 0, n_cand_total = 0;
    double t_kmer = 0, t_ung = 0, t_sel = 0;

    for (uint32_t qa = 0; qa < n;) {
        Timer t_b;
        timed_ms_begin();
        // choose batch [qa, qb) by estimated hits
        uint32_t qb = qa;
        {
            const double budget = (double)HIT_CAP * 0.5;
            uint64_t res = 0;
            while (qb < n && qb - qa < (1u << 23) - 1) {
                res += h_len[qb];
                if (qb > qa && (double)res * hits_per_res > budget) break;
                qb++;
            }
        }
        uint64_t total_hits = 0;
        uint32_t qp0 = 0, qp1 = 0, nq_res = 0;
        for (;;) {   // count; shrink the batch if it overflows the cap
            qp0 = h_poff[qa]; qp1 = h_poff[qb]; nq_res = qp1 - qp0;
            d_cnt.reserve(nq_res); d_hoff.reserve((size_t)nq_res + 1);
            hipLaunchKernelGGL(sim_count_kernel, grid_for(nq_res), dim3(256), 0, stream, ddb, cfg, qa, qb, qp0, qp1, d_koff.p, d_cnt.p, d_counters.p);
            size_t tb = 0;
            auto in = rocprim::make_transform_iterator(d_cnt.p, WidenU32());
            UC_HIP(rocprim::exclusive_scan(nullptr, tb, in, d_hoff.p, (uint64_t)0, (size_t)nq_res, rocprim::plus<uint64_t>(), stream));
            temp_reserve(tb);
            UC_HIP(rocprim::exclusive_scan(d_temp.p, tb, in, d_hoff.p, (uint64_t)0, (size_t)nq_res, rocprim::plus<uint64_t>(), stream));
            uint64_t last_off = 0; uint32_t last_cnt = 0;
            UC_HIP(hipMemcpyAsync(&last_off, d_hoff.p + (nq_res - 1), 8, hipMemcpyDeviceToHost, stream));
            UC_HIP(hipMemcpyAsync(&last_cnt, d_cnt.p + (nq_res - 1), 4, hipMemcpyDeviceToHost, stream));
            UC_HIP(hipStreamSynchronize(stream));
            total_hits = last_off + last_cnt;
            if (total_hits <= HIT_CAP || qb - qa == 1) break;
            // too many: the sim counter over-counts on a retry, so remember and subtract
            qb = qa + std::max<uint32_t>(1, (qb - qa) / 2);
            UC_HIP(hipMemsetAsync(d_counters.p, 0, 8, stream));   // n_sim restarts for this batch (accumulated on host below)
        }
        if (total_hits > (1ull << 34)) fail(UC_ERR_GENERIC, "query %u alone produces %llu k-mer hits", qa, (unsigned long long)total_hits);
        hits_per_res = std::max(1.0, (double)total_hits / std::max<uint32_t>(1, nq_res)) * 1.25;
        n_hits_total += total_hits;
        // harvest the similar-k-mer counter of this batch
        {
            unsigned long long ns = 0;
            UC_HIP(hipMemcpy(&ns, d_counters.p, 8, hipMemcpyDeviceToHost));
            stats.n_sim_kmers += ns;
            UC_HIP(hipMemsetAsync(d_counters.p, 0, 8, stream));
        }
        uint64_t n_cand = 0;
        if (total_hits) {
            d_keys.reserve(total_hits); d_keys2.reserve(total_hits);
            hipLaunchKernelGGL(sim_emit_kernel, grid_for(nq_res), dim3(256), 0, stream, ddb, cfg, qa, qb, qp0, qp1,
                               d_koff.p, d_eseq.p, d_epos.p, d_hoff.p, d_keys.p);
            unsigned qbits = 1;
            while ((1u << qbits) < qb - qa) qbits++;
            size_t tb = 0;
            UC_HIP(rocprim::radix_sort_keys(nullptr, tb, d_keys.p, d_keys2.p, (size_t)total_hits, 0u, 41u + qbits, stream));
            temp_reserve(tb);
            UC_HIP(rocprim::radix_sort_keys(d_temp.p, tb, d_keys.p, d_keys2.p, (size_t)total_hits, 0u, 41u + qbits, stream));
            d_flag.reserve(total_hits); d_bestdiag.reserve(total_hits); d_pos.reserve(total_hits);
            hipLaunchKernelGGL(diag_select_kernel, grid_for(total_hits), dim3(256), 0, stream, d_keys2.p, total_hits, p.min_diag_hits, d_flag.p, d_bestdiag.p);
            auto fin = rocprim::make_transform_iterator(d_flag.p, WidenU32());
            UC_HIP(rocprim::exclusive_scan(nullptr, tb, fin, d_pos.p, (uint64_t)0, (size_t)total_hits, rocprim::plus<uint64_t>(), stream));
            temp_reserve(tb);
            UC_HIP(rocprim::exclusive_scan(d_temp.p, tb, fin, d_pos.p, (uint64_t)0, (size_t)total_hits, rocprim::plus<uint64_t>(), stream));
            uint64_t lp = 0; uint32_t lf = 0;
            UC_HIP(hipMemcpyAsync(&lp, d_pos.p + (total_hits - 1), 8, hipMemcpyDeviceToHost, stream));
            UC_HIP(hipMemcpyAsync(&lf, d_flag.p + (total_hits - 1), 4, hipMemcpyDeviceToHost, stream));
            UC_HIP(hipStreamSynchronize(stream));
            n_cand = lp + lf;
        }
        UC_HIP(hipGetLastError());
        gpu_ms += timed_ms_end();
        t_kmer += t_b.seconds();
        n_cand_total += n_cand;

        if (n_cand) {
            // -------------------------------------------------------- E3: ungapped rescoring
            Timer t_u;
            timed_ms_begin();
            d_cq.reserve(n_cand); d_ct.reserve(n_cand); d_cd.reserve(n_cand); d_score.reserve(n_cand);
            hipLaunchKernelGGL(cand_scatter_kernel, grid_for(total_hits), dim3(256), 0, stream, d_keys2.p, total_hits, d_flag.p, d_pos.p,
                               d_bestdiag.p, qa, d_cq.p, d_ct.p, d_cd.p);
            launch_ungapped(ddb, n_cand, d_cq.p, d_ct.p, d_cd.p, d_score.p, d_counters.p + 2, stream);
            UC_HIP(hipGetLastError());
            gpu_ms += timed_ms_end();
            t_ung += t_u.seconds();
            // -------------------------------------------------------- E4: select
            Timer t_s;
            timed_ms_begin();
            d_skey.reserve(n_cand); d_skey2.reserve(n_cand); d_cd2.reserve(n_cand);
            UC_HIP(hipMemsetAsync(d_counters.p + 1, 0, 8, stream));
            hipLaunchKernelGGL(select_key_kernel, grid_for(n_cand), dim3(256), 0, stream, n_cand, d_cq.p, d_ct.p, d_score.p, p.min_ungapped, d_skey.p, d_counters.p + 1);
            size_t tb = 0;
            UC_HIP(rocprim::radix_sort_pairs(nullptr, tb, d_skey.p, d_skey2.p, d_cd.p, d_cd2.p, (size_t)n_cand, 0u, 64u, stream));
            temp_reserve(tb);
            UC_HIP(rocprim::radix_sort_pairs(d_temp.p, tb, d_skey.p, d_skey2.p, d_cd.p, d_cd2.p, (size_t)n_cand, 0u, 64u, stream));
            unsigned long long kept = 0;
            UC_HIP(hipMemcpyAsync(&kept, d_counters.p + 1, 8, hipMemcpyDeviceToHost, stream));
            UC_HIP(hipStreamSynchronize(stream));
            if (kept) {   // rank inside each query's run, keep the first max_seqs, append to the device hit lists
                d_flag.reserve(kept); d_pos.reserve(kept);
                hipLaunchKernelGGL(rank_flag_kernel, grid_for(kept), dim3(256), 0, stream, d_skey2.p, (uint64_t)kept, (uint32_t)p.max_seqs, d_flag.p);
                auto rin = rocprim::make_transform_iterator(d_flag.p, WidenU32());
                UC_HIP(rocprim::exclusive_scan(nullptr, tb, rin, d_pos.p, (uint64_t)0, (size_t)kept, rocprim::plus<uint64_t>(), stream));
                temp_reserve(tb);
                UC_HIP(rocprim::exclusive_scan(d_temp.p, tb, rin, d_pos.p, (uint64_t)0, (size_t)kept, rocprim::plus<uint64_t>(), stream));
                uint64_t lp = 0; uint32_t lf = 0;
                UC_HIP(hipMemcpyAsync(&lp, d_pos.p + (kept - 1), 8, hipMemcpyDeviceToHost, stream));
                UC_HIP(hipMemcpyAsync(&lf, d_flag.p + (kept - 1), 4, hipMemcpyDeviceToHost, stream));
                UC_HIP(hipStreamSynchronize(stream));
                const uint64_t add = lp + lf;
                d_hq.grow_preserve(n_hits + add, n_hits); d_ht.grow_preserve(n_hits + add, n_hits);
                d_hs.grow_preserve(n_hits + add, n_hits); d_hd.grow_preserve(n_hits + add, n_hits);
                hipLaunchKernelGGL(hit_scatter_kernel, grid_for(kept), dim3(256), 0, stream, d_skey2.p, d_cd2.p, (uint64_t)kept, d_flag.p, d_pos.p,
                                   d_hq.p + n_hits, d_ht.p + n_hits, d_hs.p + n_hits, d_hd.p + n_hits);
                n_hits += add;
            }
            UC_HIP(hipGetLastError());
            gpu_ms += timed_ms_end();
            t_sel += t_s.seconds();
        }
        qa = qb;
    }
    // per-query counts / offsets (the host only keeps these two small arrays)
    if (n_hits) {
        d_cnt.reserve(n);
        hipLaunchKernelGGL(hit_count_kernel, grid_for(n), dim3(256), 0, stream, d_hq.p, n_hits, n, d_cnt.p);
        UC_HIP(hipMemcpyAsync(hit_cnt.data(), d_cnt.p, (size_t)n * 4, hipMemcpyDeviceToHost, stream));
        UC_HIP(hipStreamSynchronize(stream));
    }
    for (uint32_t q = 0; q < n; q++) hit_off[q + 1] = hit_off[q] + hit_cnt[q];
    if (hit_off[n] != n_hits) fail(UC_ERR_GENERIC, "prefilter: hit list bookkeeping mismatch");
    unsigned long long ovl = 0;
    UC_HIP(hipMemcpy(&ovl, d_counters.p + 2, 8, hipMemcpyDeviceToHost));
    const uint64_t ungapped_bytes = ovl + 16ull * n_cand_total;   // (overlap + 16) B per candidate, SURVEY.md 8(d)
    stats.n_kmer_hits += n_hits_total;
    stats.n_candidates += n_cand_total;
    stats.n_prefilter_hits += n_hits;
    stats.algorithmic_bytes[UC_ST_KMER] += 8ull * stats.n_sim_kmers + 6ull * n_hits_total + 8ull * n_cand_total;
    stats.algorithmic_bytes[UC_ST_UNGAPPED] += ungapped_bytes;
    stats.algorithmic_bytes[UC_ST_SELECT] += 16ull * n_cand_total;
    stats.stage_seconds[UC_ST_KMER] += t_kmer;
    stats.stage_seconds[UC_ST_UNGAPPED] += t_ung;
    stats.stage_seconds[UC_ST_SELECT] += t_sel;
    stats.prefilter_kernel_ms += gpu_ms;
}

}  // namespace uc
