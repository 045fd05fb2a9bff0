// uc_linclust.hip — stage E8a: candidate pairs of the linear-time pre-clustering step (spec UC-1 E8a; restates Linclust,
// Steinegger & Soeding 2018 - the redundancy filter in front of Foldseek's default clustering workflow).
// Every sequence keeps its m k-mers with the smallest (hash, position); the kept (k-mer, sequence) entries are grouped by
// k-mer, the longest sequence of a group (ties: smallest id) is its centre, and every other member forms a candidate pair
// (centre, member).  On the device since r3c (the host version - hashing 47 M k-mers on the CPU threads and two LSD sorts - was
// 90 ms of the 0.5 s default workflow at C2; here: one selection kernel, two rocPRIM sorts, a group kernel, a compaction).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <vector>

#include <rocprim/rocprim.hpp>

#include "uc_engine.h"

namespace uc {

namespace {

struct LcCfg { int koff[K]; int span; int m; };

__device__ __forceinline__ uint64_t lc_hash(uint32_t v) {   // SplitMix64 finaliser of the k-mer value
    uint64_t z = (uint64_t)v + 0x9E3779B97F4A7C15ull;
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 27; z *= 0x94D049BB133111EBull; z ^= z >> 31;
    return z;
}

// A wave per sequence.  Round k finds the smallest (hash, position) above the winner of round k - 1: nothing is stored or
// removed, the candidates are simply hashed again (m x L x ~30 instructions / 64 lanes: 2 ms for 47 M residues, m = 21).
__global__ void __launch_bounds__(256) lc_select_kernel(const DeviceDb db, LcCfg cfg, uint64_t *ent, unsigned long long *n_valid) {
    const int lane = threadIdx.x & 63;
    const uint32_t nw = gridDim.x * 4, w0 = blockIdx.x * 4 + (threadIdx.x >> 6);
    unsigned long long valid = 0;
    for (uint32_t s = w0; s < db.n; s += nw) {
        const uint32_t base = db.off[s];
        const int64_t L = (int64_t)db.len[s];
        const int64_t ncand = min((int64_t)65536, L - cfg.span + 1);
        uint64_t ph = 0;
        uint32_t pp = 0;
        bool first = true, done = false;
        for (int k = 0; k < cfg.m; k++) {
            uint64_t bh = ~0ull;
            uint32_t bp = 0xFFFFFFFFu, bv = 0;
            if (!done) {
                for (int64_t j = lane; j < ncand; j += 64) {
                    uint32_t v = 0, mul = 1;
                    bool ok = true;
#pragma unroll
                    for (int t = 0; t < K; t++) {
                        const uint32_t c = db.s3[base + (uint32_t)j + (uint32_t)cfg.koff[t]];
                        ok &= c < (uint32_t)KA;
                        v += c * mul; mul *= KA;
                    }
                    if (!ok) continue;
                    const uint64_t h = lc_hash(v);
                    const bool above = first || h > ph || (h == ph && (uint32_t)j > pp);
                    const bool below = h < bh || (h == bh && (uint32_t)j < bp);
                    if (above && below) { bh = h; bp = (uint32_t)j; bv = v; }
                }
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) {
                    const uint64_t oh = (uint64_t)__shfl_xor((unsigned long long)bh, o, 64);
                    const uint32_t op = (uint32_t)__shfl_xor((int)bp, o, 64), ov = (uint32_t)__shfl_xor((int)bv, o, 64);
                    if (oh < bh || (oh == bh && op < bp)) { bh = oh; bp = op; bv = ov; }
                }
                if (bp == 0xFFFFFFFFu) done = true;
            }
            if (lane == 0) {
                ent[(uint64_t)s * cfg.m + k] = done ? ~0ull : (((uint64_t)bv << 32) | s);
                valid += done ? 0 : 1;
            }
            ph = bh; pp = bp; first = false;
        }
    }
    if (lane == 0 && valid) atomicAdd(n_valid, valid);
}

// entries sorted by (k-mer, sequence): the head of every k-mer group picks the centre and writes the group's pairs
__global__ void __launch_bounds__(256) lc_group_kernel(const uint64_t *ent, uint64_t ne, const uint32_t *len, uint64_t *pr) {
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < ne; i += (uint64_t)gridDim.x * 256) {
        const uint32_t v = (uint32_t)(ent[i] >> 32);
        if (i > 0 && (uint32_t)(ent[i - 1] >> 32) == v) continue;
        uint32_t c = (uint32_t)ent[i], lc = len[c];          // centre: longest sequence of the group, ties: smallest id
        uint64_t e = i + 1;
        for (; e < ne && (uint32_t)(ent[e] >> 32) == v; e++) {
            const uint32_t s = (uint32_t)ent[e], ls = len[s];
            if (ls > lc) { c = s; lc = ls; }
        }
        for (uint64_t k = i; k < e; k++) {
            const uint32_t s = (uint32_t)ent[k];
            pr[k] = (s != c && (k == i || ent[k] != ent[k - 1])) ? (((uint64_t)c << 32) | s) : ~0ull;
        }
    }
}

__global__ void __launch_bounds__(256) lc_flag_kernel(const uint64_t *pr, uint64_t n, uint32_t *flag) {
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256)
        flag[i] = (pr[i] != ~0ull && (i == 0 || pr[i] != pr[i - 1])) ? 1u : 0u;
}

__global__ void __launch_bounds__(256) lc_scatter_kernel(const uint64_t *pr, const uint32_t *flag, const uint32_t *pos, uint64_t n, uint32_t *out) {
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256)
        if (flag[i]) { out[2ull * pos[i]] = (uint32_t)(pr[i] >> 32); out[2ull * pos[i] + 1] = (uint32_t)pr[i]; }
}

__global__ void __launch_bounds__(256) lc_hits_kernel(const uint32_t *pairs, uint64_t np, uint32_t *hq, uint32_t *ht, uint32_t *cnt) {
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < np; i += (uint64_t)gridDim.x * 256) {
        const uint32_t c = pairs[2 * i];
        hq[i] = c; ht[i] = pairs[2 * i + 1];
        atomicAdd(cnt + c, 1u);
    }
}

inline dim3 lc_grid(uint64_t n) { return dim3((uint32_t)std::min<uint64_t>(std::max<uint64_t>(1, (n + 255) / 256), 16384)); }

}  // namespace

// (centre, member) candidate pairs of the linear-time pre-step for the database resident on this engine's device, sorted by
// (centre, member), unique
std::vector<uint32_t> Engine::linclust_pairs_impl(uint64_t *install) {
    PressureScope ps(*this, 0);
    if (!have_db) fail(UC_ERR_ARGS, "no database loaded");
    UC_HIP(hipSetDevice(device));
    const uint32_t n = hdb.n;
    const int m = p.kmer_per_seq;
    std::vector<uint32_t> out;
    if (n == 0) return out;
    LcCfg cfg;
    for (int k = 0; k < K; k++) cfg.koff[k] = p.koff[k];
    cfg.span = p.span;
    cfg.m = m;
    const uint64_t cap = (uint64_t)n * m;
    unsigned nbits = 1;
    while (nbits < 32 && (1ull << nbits) < n) nbits++;
    const unsigned pbits = 32 + nbits;          // (centre << 32 | member); the invalid key (all ones) still sorts last
    DevBuf<uint64_t> ent, ent2, pr, pr2;
    DevBuf<uint32_t> flag, pos, dout;
    DevBuf<unsigned long long> cnt;
    DevBuf<char> tmp;
    ent.reserve(cap); ent2.reserve(cap); cnt.reserve(1);
    UC_HIP(hipMemsetAsync(cnt.p, 0, 8, stream));
    hipLaunchKernelGGL(lc_select_kernel, dim3((uint32_t)std::min<uint64_t>((n + 3) / 4, 8192)), dim3(256), 0, stream, ddb, cfg, ent.p, cnt.p);
    size_t tb = 0;
    UC_HIP(rocprim::radix_sort_keys(nullptr, tb, ent.p, ent2.p, (size_t)cap, 0u, 58u, stream));
    tmp.reserve(tb + 256);
    UC_HIP(rocprim::radix_sort_keys(tmp.p, tb, ent.p, ent2.p, (size_t)cap, 0u, 58u, stream));
    unsigned long long ne = 0;
    UC_HIP(hipMemcpyAsync(&ne, cnt.p, 8, hipMemcpyDeviceToHost, stream));
    UC_HIP(hipStreamSynchronize(stream));
    if (ne == 0) return out;
    pr.reserve(ne); pr2.reserve(ne); flag.reserve(ne); pos.reserve(ne);
    hipLaunchKernelGGL(lc_group_kernel, lc_grid(ne), dim3(256), 0, stream, ent2.p, (uint64_t)ne, ddb.len, pr.p);
    UC_HIP(rocprim::radix_sort_keys(nullptr, tb, pr.p, pr2.p, (size_t)ne, 0u, pbits, stream));
    tmp.reserve(tb + 256);
    UC_HIP(rocprim::radix_sort_keys(tmp.p, tb, pr.p, pr2.p, (size_t)ne, 0u, pbits, stream));
    hipLaunchKernelGGL(lc_flag_kernel, lc_grid(ne), dim3(256), 0, stream, pr2.p, (uint64_t)ne, flag.p);
    UC_HIP(rocprim::exclusive_scan(nullptr, tb, flag.p, pos.p, 0u, (size_t)ne, rocprim::plus<uint32_t>(), stream));
    tmp.reserve(tb + 256);
    UC_HIP(rocprim::exclusive_scan(tmp.p, tb, flag.p, pos.p, 0u, (size_t)ne, rocprim::plus<uint32_t>(), stream));
    uint32_t lp = 0, lf = 0;
    UC_HIP(hipMemcpyAsync(&lp, pos.p + (ne - 1), 4, hipMemcpyDeviceToHost, stream));
    UC_HIP(hipMemcpyAsync(&lf, flag.p + (ne - 1), 4, hipMemcpyDeviceToHost, stream));
    UC_HIP(hipStreamSynchronize(stream));
    const uint64_t np = (uint64_t)lp + lf;
    if (np == 0) return out;
    dout.reserve(2 * np);
    hipLaunchKernelGGL(lc_scatter_kernel, lc_grid(ne), dim3(256), 0, stream, pr2.p, flag.p, pos.p, (uint64_t)ne, dout.p);
    if (install) {   // the pairs ARE the hit lists of the pre-step: query = centre, target = member, sorted by (centre, member)
        const size_t cap = (size_t)np;
        d_hq.reserve(cap); d_ht.reserve(cap); d_hs.reserve(cap); d_hd.reserve(cap);
        DevBuf<uint32_t> dcnt;
        dcnt.reserve(n);
        UC_HIP(hipMemsetAsync(dcnt.p, 0, (size_t)n * 4, stream));
        UC_HIP(hipMemsetAsync(d_hs.p, 0, cap * 4, stream));
        UC_HIP(hipMemsetAsync(d_hd.p, 0, cap * 4, stream));
        hipLaunchKernelGGL(lc_hits_kernel, lc_grid(np), dim3(256), 0, stream, dout.p, np, d_hq.p, d_ht.p, dcnt.p);
        hit_cnt.assign(n, 0);
        UC_HIP(hipMemcpyAsync(hit_cnt.data(), dcnt.p, (size_t)n * 4, hipMemcpyDeviceToHost, stream));
        UC_HIP(hipStreamSynchronize(stream));
        UC_HIP(hipGetLastError());
        hit_off.assign((size_t)n + 1, 0);
        for (uint32_t q = 0; q < n; q++) hit_off[q + 1] = hit_off[q] + hit_cnt[q];
        n_hits = hit_off[n];
        if (n_hits != np) fail(UC_ERR_GENERIC, "linclust: hit list bookkeeping mismatch");
        alns_valid = false;
        clear_edges();
        *install = np;
        return out;
    }
    out.resize(2 * np);
    UC_HIP(hipMemcpyAsync(out.data(), dout.p, 2 * np * 4, hipMemcpyDeviceToHost, stream));
    UC_HIP(hipStreamSynchronize(stream));
    UC_HIP(hipGetLastError());
    return out;
}

std::vector<uint32_t> Engine::linclust_pairs() { return linclust_pairs_impl(nullptr); }

// the same pairs installed as this engine's hit lists (one rank: nothing goes through the host); returns their number
uint64_t Engine::linclust_hits() {
    uint64_t np = 0;
    hit_cnt.assign(hdb.n, 0);
    hit_off.assign((size_t)hdb.n + 1, 0);
    n_hits = 0;
    alns_valid = false;
    clear_edges();
    (void)linclust_pairs_impl(&np);
    return np;
}

void preload_linclust_module() { hipFuncAttributes fa; (void)hipFuncGetAttributes(&fa, (const void *)lc_select_kernel); }
}  // namespace uc
