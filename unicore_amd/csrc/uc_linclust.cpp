// uc_linclust.cpp — stage E8a: candidate pairs of the linear-time pre-clustering step (spec UC-1 E8a;
// restates Linclust, Steinegger & Soeding 2018 - the redundancy filter in front of Foldseek's default
// clustering workflow).  Host side: hashing 47 M k-mers and sorting 3 M kept entries is a fraction of a
// second; the alignments of the candidate pairs - the actual work - run through the GPU stages E5/E6 like any hit list.
#include <algorithm>
#include <cstdint>
#include <thread>
#include <vector>

#include "uc_engine.h"

namespace uc {

namespace {
inline uint64_t lc_hash(uint32_t v) {   // SplitMix64 finaliser of the k-mer value
    uint64_t z = (uint64_t)v + 0x9E3779B97F4A7C15ull;
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 27; z *= 0x94D049BB133111EBull; z ^= z >> 31;
    return z;
}
struct Cand { uint64_t h; uint32_t v, pos; };
struct Ent { uint32_t v, seq; };

void radix_sort_u64(std::vector<uint64_t> &a) {   // LSD, 11 bits per pass, passes whose digit is constant are skipped
    if (a.size() < 2) return;
    uint64_t all_or = 0, all_and = ~0ull;
    for (uint64_t x : a) { all_or |= x; all_and &= x; }
    const uint64_t varying = all_or ^ all_and;
    std::vector<uint64_t> b(a.size());
    std::vector<size_t> cnt(2048);
    for (int sh = 0; sh < 64; sh += 11) {
        if (((varying >> sh) & 2047ull) == 0) continue;
        std::fill(cnt.begin(), cnt.end(), 0);
        for (uint64_t x : a) cnt[(x >> sh) & 2047]++;
        size_t run = 0;
        for (size_t &c : cnt) { const size_t k = c; c = run; run += k; }
        for (uint64_t x : a) b[cnt[(x >> sh) & 2047]++] = x;
        a.swap(b);
    }
}
}  // namespace

std::vector<uint32_t> linclust_pairs(const HostDb &db, const Params &p, int threads) {
    const uint32_t n = db.n;
    const int m = p.kmer_per_seq;
    std::vector<std::vector<Ent>> part;
    const unsigned T = std::max(1, std::min(threads, 64));
    part.resize(T);
    auto work = [&](unsigned t) {   // every sequence keeps its m k-mers with the smallest (hash, position)
        std::vector<Cand> cand;
        const uint32_t lo = (uint32_t)((uint64_t)n * t / T), hi = (uint32_t)((uint64_t)n * (t + 1) / T);
        for (uint32_t s = lo; s < hi; s++) {
            const uint8_t *x = db.s3.data() + db.off[s];
            const int64_t l = (int64_t)db.len(s);
            cand.clear();
            for (int64_t j = 0; j + p.span <= l && j <= 65535; j++) {
                uint32_t v = 0, mul = 1;
                bool ok = true;
                for (int k = 0; k < K; k++) {
                    const uint8_t c = x[j + p.koff[k]];
                    ok &= c < KA;
                    v += c * mul; mul *= KA;
                }
                if (ok) cand.push_back({lc_hash(v), v, (uint32_t)j});
            }
            const size_t keep = std::min<size_t>(cand.size(), (size_t)m);
            std::partial_sort(cand.begin(), cand.begin() + keep, cand.end(),
                              [](const Cand &a, const Cand &b) { return a.h != b.h ? a.h < b.h : a.pos < b.pos; });
            for (size_t k = 0; k < keep; k++) part[t].push_back({cand[k].v, s});
        }
    };
    if (T == 1) work(0);
    else {
        std::vector<std::thread> th;
        for (unsigned t = 0; t < T; t++) th.emplace_back(work, t);
        for (auto &x : th) x.join();
    }
    // group by k-mer value: entries as (value << 32 | sequence) keys, LSD radix sort (the comparison sorts of these two
    // lists were most of this function's time)
    std::vector<uint64_t> ent;
    {
        size_t tot = 0;
        for (auto &v : part) tot += v.size();
        ent.reserve(tot);
        for (auto &v : part)
            for (const Ent &x : v) ent.push_back(((uint64_t)x.v << 32) | x.seq);
    }
    radix_sort_u64(ent);
    std::vector<uint64_t> pr;   // (centre << 32 | member)
    for (size_t b = 0; b < ent.size();) {
        size_t e = b;
        while (e < ent.size() && (ent[e] >> 32) == (ent[b] >> 32)) e++;
        uint32_t c = (uint32_t)ent[b];         // centre: longest sequence of the group, ties: smallest id
        uint64_t lc = db.len(c);
        for (size_t k = b + 1; k < e; k++)
            if (db.len((uint32_t)ent[k]) > lc) { c = (uint32_t)ent[k]; lc = db.len(c); }
        for (size_t k = b; k < e; k++)
            if ((uint32_t)ent[k] != c && (k == b || ent[k] != ent[k - 1])) pr.push_back(((uint64_t)c << 32) | (uint32_t)ent[k]);
        b = e;
    }
    radix_sort_u64(pr);
    pr.erase(std::unique(pr.begin(), pr.end()), pr.end());
    std::vector<uint32_t> out(2 * pr.size());
    for (size_t k = 0; k < pr.size(); k++) { out[2 * k] = (uint32_t)(pr[k] >> 32); out[2 * k + 1] = (uint32_t)pr[k]; }
    return out;
}

}  // namespace uc
